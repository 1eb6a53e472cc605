"""Building blocks of the MRFA hot path with the reference's class names, constructor arguments and parameter
names (modules/util.py in the reference), executing on the HIP engine.

Each block is an nn.Module that only *holds* parameters (nn.Conv2d / BatchNorm2d containers, so state_dict keys, shapes
and default initialisation equal the reference's); the computation is `run(e, x)` on NHWC Views of mrfa_amd.engine,
i.e. explicit launches of the kernels in libmrfa_hip.so.  Calling a block like a function (`block(x)` with an NCHW
tensor) goes through the same engine via run_block().
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import BatchNorm2d

from ..engine import Ctx, View, run_program


class AttributeDict(dict):
    """reference: modules/util.py:9-24"""

    def __getattr__(self, attr):
        try:
            return self[attr]
        except KeyError:
            raise AttributeError(attr)

    def __setattr__(self, attr, value):
        self[attr] = value


def convert_dict_to_attrit_dict(d):
    out = AttributeDict()
    for k, v in d.items():
        out[k] = convert_dict_to_attrit_dict(v) if isinstance(v, dict) else v
    return out


# ------------------------------------------------------------------------------------------------ small torch helpers
# (used by the tiny "glue" islands that run as torch device ops on (B,10,2)-sized tensors)
def make_coordinate_grid(spatial_size, type=None, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(h,w,2) grid over [-1,1]^2, last dim (x,y).  reference: modules/util.py:90-108.  The reference's second argument is a
    tensor-type STRING (`type=frame.type()`, model.py:37,45); a tensor to take dtype / device from is accepted as well."""
    like = type if like is None else like
    if not torch.is_tensor(like):
        like = torch.empty(0).type(like) if like is not None else torch.empty(0)
    h, w = spatial_size
    xs = 2.0 * (torch.arange(w, dtype=like.dtype, device=like.device) / (w - 1)) - 1.0
    ys = 2.0 * (torch.arange(h, dtype=like.dtype, device=like.device) / (h - 1)) - 1.0
    return torch.stack([xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)], dim=-1)


_COORD_GRIDS: dict = {}


def coords_grid_nhwc(h: int, w: int, like: torch.Tensor) -> torch.Tensor:
    """(1,h,w,2) pixel-index grid (x,y): NHWC form of coords_grid, reference: modules/util.py:53-56.  A constant: kept per (size, device, dtype) -- RaftFlow asks
    for seven of them per forward, five small launches each on the decoder's critical path.  Never modified by its users; grids first built while a hipGraph is
    being captured are not kept (they live in that graph's private pool)."""
    key = (h, w, like.device, like.dtype)
    g = _COORD_GRIDS.get(key)
    if g is None:
        ys, xs = torch.meshgrid(torch.arange(h, device=like.device), torch.arange(w, device=like.device), indexing="ij")
        g = torch.stack([xs, ys], dim=-1).to(like.dtype)[None]
        if not (like.is_cuda and torch.cuda.is_current_stream_capturing()):
            _COORD_GRIDS[key] = g
    return g


def kp2gaussian(kp: torch.Tensor, spatial_size, kp_variance: float) -> torch.Tensor:
    """(B,K,2) -> (B,K,h,w).  reference: modules/util.py:59-87"""
    h, w = spatial_size
    g = make_coordinate_grid((h, w), kp).view(1, 1, h, w, 2)
    d = g - kp.view(kp.shape[0], kp.shape[1], 1, 1, 2)
    return torch.exp(-0.5 * (d * d).sum(-1) / kp_variance)


# ------------------------------------------------------------------------------------------------ reference helper surface
class _Stateless(nn.Module):
    """parameter-free owner for functional programs (run_program wants a module for its train flag and parameter list)"""


_FN = _Stateless()


def coords_grid(batch, ht, wd, device="cpu"):
    """(batch,2,ht,wd) fp32 pixel-index grid, channel 0 = x, channel 1 = y.  reference: modules/util.py:53-56"""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def bilinear_sampler(img, coords, mode='bilinear', mask=False):
    """grid_sample in PIXEL coordinates (align_corners=True, zeros outside): img (N,C,H,W), coords (N,h,w,2) as (x,y) ->
    (N,C,h,w) [, in-bounds mask (N,h,w,1)].  reference: modules/util.py:26-38 (same arguments; `mode` is ignored there too).
    Runs mrfa_grid_sample_fwd/bwd mode 1 (K10), differentiable w.r.t. img and coords."""
    assert img.dim() == 4 and coords.dim() == 4 and coords.shape[-1] == 2 and coords.shape[0] == img.shape[0]
    N, C, H, W = img.shape
    h, w = coords.shape[1], coords.shape[2]

    def program(e: Ctx, img_, coords_):
        f = e.from_nchw(img_)
        # the kernel samples at (output pixel index + flow): hand it coords - index grid (exact in fp32 for |index| < 2^24)
        flow = e.wrap_nhwc((coords_.float() - coords_grid_nhwc(h, w, coords_.float())).contiguous())
        o = e.grid_sample(f, flow, 1)
        return (e.to_nchw(o),), (lambda g: e.seed_grad_nchw(o, g),), \
            ((lambda: e.grad_to_nchw(f) if f.has_grad else None), (lambda: flow.st.grad.view(N, h, w, 2) if flow.has_grad else None))
    out = run_program(_FN, program, [img, coords])[0]
    if mask:
        xg, yg = (2 * coords[..., 0:1] / (W - 1) - 1), (2 * coords[..., 1:2] / (H - 1) - 1)
        return out, ((xg > -1) & (yg > -1) & (xg < 1) & (yg < 1)).float()
    return out


def batch_bilinear_sampler(img, coords, mode='bilinear', mask=False, h=256, w=256, mini_batch=4):
    """reference: modules/util.py:40-51 -- the same sampling issued in mini-batches of `mini_batch` x h x w one-channel maps to
    bound torch's temporaries; the kernel has none, so this is bilinear_sampler (identical results)."""
    return bilinear_sampler(img, coords, mode=mode, mask=mask)


# ------------------------------------------------------------------------------------------------ blocks
def run_block(block: nn.Module, x: torch.Tensor, **kw) -> torch.Tensor:
    """NCHW tensor in, NCHW tensor out, through the engine (one autograd node)."""
    def program(e: Ctx, xin):
        xv = e.from_nchw(xin)
        yv = block.run(e, xv, **kw)
        y = e.to_nchw(yv)
        return (y,), (lambda g: e.seed_grad_nchw(yv, g),), (lambda: e.grad_to_nchw(xv),)
    return run_program(block, program, [x])[0]


class _Block(nn.Module):
    def forward(self, x):
        return run_block(self, x)


class DownBlock2d(_Block):
    """conv3x3 -> BN -> ReLU -> avgpool2.  reference: modules/util.py:180-197"""

    def __init__(self, in_features, out_features, kernel_size=3, padding=1, groups=1):
        super().__init__()
        assert groups == 1
        self.conv = nn.Conv2d(in_features, out_features, kernel_size=kernel_size, padding=padding)
        self.norm = BatchNorm2d(out_features, affine=True)

    def run(self, e: Ctx, x: View, out: Optional[View] = None, need_dx=True) -> View:
        st = e.bn_stats_buf(self.norm)
        raw = e.conv(x, self.conv, stats=st, need_dx=need_dx, fin=e.fin(self.norm))
        return e.bn_act(raw, self.norm, st, relu=True, pool=True, out=out, sole_consumer=True)


class UpBlock2d(_Block):
    """nearest x2 -> conv3x3 -> BN -> ReLU (upsample fused into the conv's input indexing).  reference: util.py:160-177"""

    def __init__(self, in_features, out_features, kernel_size=3, padding=1, groups=1):
        super().__init__()
        assert groups == 1
        self.conv = nn.Conv2d(in_features, out_features, kernel_size=kernel_size, padding=padding)
        self.norm = BatchNorm2d(out_features, affine=True)

    def run(self, e: Ctx, x: View, out: Optional[View] = None, blend=None) -> View:
        st = e.bn_stats_buf(self.norm)
        raw = e.conv(x, self.conv, stats=st, ups=True, fin=e.fin(self.norm))
        return e.bn_act(raw, self.norm, st, relu=True, blend=blend, out=out, sole_consumer=True)


class SameBlock2d(_Block):
    """conv -> BN -> ReLU.  reference: modules/util.py:199-214"""

    def __init__(self, in_features, out_features, groups=1, kernel_size=3, padding=1):
        super().__init__()
        assert groups == 1
        self.conv = nn.Conv2d(in_features, out_features, kernel_size=kernel_size, padding=padding)
        self.norm = BatchNorm2d(out_features, affine=True)

    def run(self, e: Ctx, x: View, out: Optional[View] = None, need_dx=True) -> View:
        st = e.bn_stats_buf(self.norm)
        raw = e.conv(x, self.conv, stats=st, need_dx=need_dx, fin=e.fin(self.norm))
        return e.bn_act(raw, self.norm, st, relu=True, out=out, sole_consumer=True)


class ResBlock2d(_Block):
    """BN-ReLU-conv, BN-ReLU-conv, + skip; both BN+ReLU run in the conv prologues.  reference: modules/util.py:135-157"""

    def __init__(self, in_features, kernel_size, padding):
        super().__init__()
        self.conv1 = nn.Conv2d(in_features, in_features, kernel_size=kernel_size, padding=padding)
        self.conv2 = nn.Conv2d(in_features, in_features, kernel_size=kernel_size, padding=padding)
        self.norm1 = BatchNorm2d(in_features, affine=True)
        self.norm2 = BatchNorm2d(in_features, affine=True)

    def run(self, e: Ctx, x: View, out: Optional[View] = None, x_stats=None) -> View:
        pre1 = e.prebn(x, self.norm1, x_stats)
        st2 = e.bn_stats_buf(self.norm2)
        y1 = e.conv(x, self.conv1, pre=pre1, stats=st2, fin=e.fin(self.norm2))
        pre2 = e.prebn(y1, self.norm2, st2)
        return e.conv(y1, self.conv2, out=out, pre=pre2, res=x)


class ChannelBlock2d(_Block):
    """BN(2C) -> ReLU -> conv3x3 2C->C.  reference: modules/util.py:111-133"""

    def __init__(self, in_features, kernel_size, padding):
        super().__init__()
        self.conv1 = nn.Conv2d(in_features, in_features // 2, kernel_size=kernel_size, padding=padding)
        self.norm1 = BatchNorm2d(in_features, affine=True)

    def run(self, e: Ctx, x: View, out: Optional[View] = None, out_stats=None, out_fin=None) -> View:
        pre = e.prebn(x, self.norm1)
        return e.conv(x, self.conv1, out=out, pre=pre, stats=out_stats, fin=out_fin)


class Encoder(nn.Module):
    """reference: modules/util.py:217-236"""

    def __init__(self, block_expansion, in_features, num_blocks=3, max_features=256):
        super().__init__()
        blocks = []
        for i in range(num_blocks):
            blocks.append(DownBlock2d(in_features if i == 0 else min(max_features, block_expansion * (2 ** i)),
                                      min(max_features, block_expansion * (2 ** (i + 1))), kernel_size=3, padding=1))
        self.down_blocks = nn.ModuleList(blocks)


class Decoder(nn.Module):
    """reference: modules/util.py:239-263"""

    def __init__(self, block_expansion, in_features, num_blocks=3, max_features=256):
        super().__init__()
        blocks = []
        for i in range(num_blocks)[::-1]:
            in_filters = (1 if i == num_blocks - 1 else 2) * min(max_features, block_expansion * (2 ** (i + 1)))
            out_filters = min(max_features, block_expansion * (2 ** i))
            blocks.append(UpBlock2d(in_filters, out_filters, kernel_size=3, padding=1))
        self.up_blocks = nn.ModuleList(blocks)
        self.out_filters = block_expansion + in_features


class Hourglass(_Block):
    """U-Net with skip concatenation.  reference: modules/util.py:266-278.

    Zero-copy concatenation: the decoder's [up_k | skip] buffers are allocated first and every encoder level writes its
    pooled output straight into the skip slot it will later be read from (no torch.cat, util.py:262)."""

    def __init__(self, block_expansion, in_features, num_blocks=3, max_features=256):
        super().__init__()
        self.encoder = Encoder(block_expansion, in_features, num_blocks, max_features)
        self.decoder = Decoder(block_expansion, in_features, num_blocks, max_features)
        self.out_filters = self.decoder.out_filters
        self.in_features = in_features

    def run(self, e: Ctx, x: View, need_dx=True) -> View:
        downs, ups = self.encoder.down_blocks, self.decoder.up_blocks
        nb = len(downs)
        # channel counts / sizes of the encoder pyramid feats[0..nb]
        chans = [x.C] + [d.conv.out_channels for d in downs]
        sizes = [(x.H >> j, x.W >> j) for j in range(nb + 1)]
        cats: List[View] = []
        for k in range(nb):
            j = nb - 1 - k
            c_up = ups[k].conv.out_channels
            cats.append(e.new(x.N, sizes[j][0], sizes[j][1], c_up + chans[j], pad32=True))
        # skip slot of feats[0] (the input itself)
        c_up_last = ups[nb - 1].conv.out_channels
        e.copy(x, out=cats[nb - 1].slice(c_up_last, c_up_last + chans[0])) if need_dx else \
            _copy_nograd(e, x, cats[nb - 1].slice(c_up_last, c_up_last + chans[0]))
        cur = x
        for j in range(nb):
            if j + 1 <= nb - 1:
                k = nb - 1 - (j + 1)
                c_up = ups[k].conv.out_channels
                slot = cats[k].slice(c_up, c_up + chans[j + 1])
            else:
                slot = None
            cur = downs[j].run(e, cur, out=slot, need_dx=(need_dx or j > 0))
        out = cur
        for k in range(nb):
            c_up = ups[k].conv.out_channels
            ups[k].run(e, out, out=cats[k].slice(0, c_up))
            out = cats[k]
        return out


def _copy_nograd(e: Ctx, x: View, out: View):
    rec = e.record
    e.record = False
    try:
        e.copy(x, out=out)
    finally:
        e.record = rec


class AntiAliasInterpolation2d(nn.Module):
    """Band-limited down-sampling; only the outputs kept by the nearest decimation are computed.
    reference: modules/util.py:282-326 (buffer `weight` (C,1,k,k), sigma=(1/s-1)/2, k=2*round(4 sigma)+1)."""

    def __init__(self, channels, scale):
        super().__init__()
        sigma = (1 / scale - 1) / 2
        kernel_size = 2 * round(sigma * 4) + 1
        self.ka = kernel_size // 2
        self.kb = self.ka - 1 if kernel_size % 2 == 0 else self.ka
        t = torch.arange(kernel_size, dtype=torch.float32)
        mean = (kernel_size - 1) / 2
        g = torch.exp(-(t - mean) ** 2 / (2 * sigma ** 2)) if sigma > 0 else torch.ones(1)
        kernel = g[:, None] * g[None, :]
        kernel = kernel / torch.sum(kernel)
        self.register_buffer('weight', kernel.view(1, 1, kernel_size, kernel_size).repeat(channels, 1, 1, 1))
        self.groups = channels
        self.scale = scale

    def run(self, e: Ctx, x_nchw: torch.Tensor) -> View:
        stride = int(round(1.0 / self.scale))
        assert abs(stride * self.scale - 1.0) < 1e-9 and self.weight.shape[-1] % 2 == 1
        return e.antialias_down(x_nchw, self.weight, stride)

    def forward(self, x):
        if self.scale == 1.0:
            return x
        e = Ctx(x.device, train=False, record=False)
        return e.to_nchw(self.run(e, x))
