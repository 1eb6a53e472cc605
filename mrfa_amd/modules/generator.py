"""OcclusionAwareGenerator encode / decode.  reference: modules/generator.py:8-69."""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from ..engine import Ctx, View, run_program
from .util import ChannelBlock2d, DownBlock2d, ResBlock2d, SameBlock2d, UpBlock2d


class OcclusionAwareGenerator(nn.Module):
    """Same kwargs / state_dict as the reference (generator.py:9-32): first, down_blocks, up_blocks, resblock,
    channel_block, final."""

    def __init__(self, num_channels, block_expansion, max_features, num_up_blocks):
        super().__init__()
        self.num_up_blocks = num_up_blocks
        self.first = SameBlock2d(num_channels, block_expansion, kernel_size=(7, 7), padding=(3, 3))
        down_blocks, up_blocks, resblock, channel_block = [], [], [], []
        for i in range(num_up_blocks):
            in_features = min(max_features, block_expansion * (2 ** i))
            out_features = min(max_features, block_expansion * (2 ** (i + 1)))
            down_blocks.append(DownBlock2d(in_features, out_features, kernel_size=(3, 3), padding=(1, 1)))
            up_blocks.append(UpBlock2d(out_features, in_features, kernel_size=(3, 3), padding=(1, 1)))
            resblock.append(ResBlock2d(out_features, kernel_size=(3, 3), padding=(1, 1)))
            channel_block.append(ChannelBlock2d(out_features * 2, kernel_size=(3, 3), padding=(1, 1)))
        self.down_blocks = nn.ModuleList(down_blocks)
        self.up_blocks = nn.ModuleList(up_blocks[::-1])
        self.resblock = nn.ModuleList(resblock[::-1])
        self.channel_block = nn.ModuleList(channel_block[::-1])
        self.final = nn.Conv2d(block_expansion, num_channels, kernel_size=(7, 7), padding=(3, 3))

    # ---- engine programs -------------------------------------------------------------------------------------
    def level_channels(self) -> List[int]:
        """channels of encode() outputs, coarse first"""
        return [self.down_blocks[-1].conv.out_channels] + [d.conv.in_channels for d in self.down_blocks[::-1]]

    def run_encode(self, e: Ctx, x: View) -> List[View]:
        """generator.py:34-42: 6-level feature pyramid, coarse first"""
        feats = [self.first.run(e, x, need_dx=False)]
        for d in self.down_blocks:
            feats.append(d.run(e, feats[-1]))
        return feats[::-1]

    def run_decode(self, e: Ctx, warp_f: List[View], warp_img: View, occ: List[View], cat_bufs: Optional[List[View]]) -> View:
        """generator.py:44-64.  With coarse warps (`cat_bufs[i]` = [blend | warp_f_c[i]] buffers whose second halves the
        caller already filled) the channel blocks run; in prior_only mode cat_bufs is None."""
        n_up = self.num_up_blocks
        if cat_bufs is not None:
            c0 = warp_f[0].C
            e.blend(warp_f[0], None, occ[0], out=cat_bufs[0].slice(0, c0))
            out = cat_bufs[0]
        else:
            out = e.blend(warp_f[0], None, occ[0])
        for i in range(n_up):
            if cat_bufs is not None:
                st = e.bn_stats_buf(self.resblock[i].norm1)
                out = self.channel_block[i].run(e, out, out_stats=st, out_fin=e.fin(self.resblock[i].norm1))
                out = self.resblock[i].run(e, out, x_stats=st)
            else:
                out = self.resblock[i].run(e, out)
            slot = None
            if cat_bufs is not None and i != n_up - 1:
                slot = cat_bufs[i + 1].slice(0, warp_f[i + 1].C)
            up = self.up_blocks[i].run(e, out, out=slot, blend=(warp_f[i + 1], occ[i + 1]))
            out = cat_bufs[i + 1] if slot is not None else up
        logits = e.conv(out, self.final)
        sig = e.act(logits, 2)
        return e.blend(warp_img, sig, occ[-1])

    # ---- public NCHW API (reference signatures) ----------------------------------------------------------------
    def encode(self, x: torch.Tensor):
        def program(e: Ctx, xin):
            feats = self.run_encode(e, e.from_nchw(xin))
            outs = tuple(e.to_nchw(f) for f in feats)
            return outs, tuple((lambda g, f=f: e.seed_grad_nchw(f, g)) for f in feats), (None,)
        return list(run_program(self, program, [x]))

    def decode(self, warp_f, warp_img, occlusion, warp_f_c=None, occlusion_c=None):
        n = len(warp_f)
        has_c = warp_f_c is not None
        ins = list(warp_f) + [warp_img] + list(occlusion) + (list(warp_f_c) if has_c else [])

        def program(e: Ctx, *t):
            wf = [e.from_nchw(a) for a in t[:n]]
            wi = e.from_nchw(t[n])
            oc = [e.from_nchw(a) for a in t[n + 1:2 * n + 1]]
            views = wf + [wi] + oc
            cats = None
            if has_c:
                wc = t[2 * n + 1:]
                cats = []
                for i in range(self.num_up_blocks):
                    c = wf[i].C
                    buf = e.new(wf[i].N, wf[i].H, wf[i].W, 2 * c)
                    e.from_nchw(wc[i], out=buf.slice(c, 2 * c))
                    cats.append(buf)
                    views.append(buf.slice(c, 2 * c))
                views.append(None)
            out = self.run_decode(e, wf, wi, oc, cats)
            y = e.to_nchw(out)
            gfn = tuple((lambda v=v: e.grad_to_nchw(v) if (v is not None and v.has_grad) else None) for v in views)
            return (y,), (lambda g: e.seed_grad_nchw(out, g),), gfn
        return run_program(self, program, ins)[0]

    def forward(self, x):
        raise NotImplementedError("the reference's OcclusionAwareGenerator.forward is broken (decode(x) without warps)")
