"""RaftFlow: structure correlation volume + 6-level coarse-to-fine motion refinement + feature warps + decode.
reference: modules/raft.py:12-311 (CorrBlock, BasicMotionEncoder, RefineFlow, RaftFlow)."""
from __future__ import annotations

import functools
import math
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

from ..engine import Ctx, Storage, View, run_program
from .generator import OcclusionAwareGenerator
from .util import _FN, Hourglass, batch_bilinear_sampler, bilinear_sampler, coords_grid, coords_grid_nhwc, kp2gaussian  # noqa: F401


class CorrBlock:
    """2-level correlation pyramid + (2r+1)^2-window bilinear lookup.  reference: modules/raft.py:12-48 (same constructor and call):
    corr (B*h1*w1, 1, H, W) = one source map per query pixel; __call__(coords (B,2,h1,w1), pixel (x,y) on the H x W map) ->
    (B, num_levels*(2r+1)^2, h1, w1), channel = lvl*(2r+1)^2 + a*(2r+1) + b sampled at (x/2^lvl + a - r, y/2^lvl + b - r).
    Runs mrfa_corr_lookup_fwd/bwd (K11/K13) on the rows of the volume in place; differentiable w.r.t. corr and coords."""

    def __init__(self, corr, num_levels=2, radius=3):
        if num_levels != 2:
            raise NotImplementedError("CorrBlock: the lookup kernel holds the reference's two pyramid levels (raft.py:238)")
        assert corr.dim() == 4 and corr.shape[1] == 1 and corr.shape[2] % 2 == 0 and corr.shape[3] % 2 == 0
        self.num_levels, self.radius = num_levels, radius
        self.corr = corr

    def __call__(self, coords):
        corr, r = self.corr, self.radius
        BQ, _, H, W = corr.shape
        B, _, h1, w1 = coords.shape
        assert B * h1 * w1 == BQ, "CorrBlock: one source map per query pixel"

        def program(e: Ctx, corr_, coords_):
            vol0 = corr_.contiguous().float().view(BQ, H * W)
            vol1 = torch.empty((BQ, (H // 2) * (W // 2)), dtype=torch.float32, device=vol0.device)
            e._chk(e.L.mrfa_avgpool2_fwd(e.s, vol0.data_ptr(), 1, BQ, H, W, 1, vol1.data_ptr(), 1), "corr pyramid level 1")
            cv = e.wrap_nhwc(coords_.float().permute(0, 2, 3, 1).contiguous())
            d = {}

            def dvols():
                if not d:
                    d[0], d[1] = torch.zeros_like(vol0), torch.zeros_like(vol1)
                return d[0], d[1]
            o = e.corr_lookup(vol0, vol1, dvols, H, W, cv, radius=r)

            def dcorr():
                if not d:
                    return None
                e._chk(e.L.mrfa_unpool2_acc(e.s, d[1].data_ptr(), 1, BQ, H // 2, W // 2, 1, d[0].data_ptr(), 1, 0.25), "corr pyramid bwd")
                return d[0].view(BQ, 1, H, W)
            return (e.to_nchw(o),), (lambda g: e.seed_grad_nchw(o, g),), \
                (dcorr, (lambda: cv.st.grad.view(B, h1, w1, 2).permute(0, 3, 1, 2) if cv.has_grad else None))
        return run_program(_FN, program, [corr, coords])[0]


class BasicMotionEncoder(nn.Module):
    """reference: modules/raft.py:50-68 (98 corr channels + 2 flow channels -> 128)"""

    def __init__(self, num_levels=2, radius=3):
        super().__init__()
        cor_planes = num_levels * (2 * radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 128, 1, padding=0)
        self.convc2 = nn.Conv2d(128, 96, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 96, 128 - 2, 3, padding=1)

    def run(self, e: Ctx, flow: View, corr: View, out: View):
        """writes [conv(126) | flow(2)] into `out` (128 channels of the RefineFlow input buffer)"""
        cor1 = e.conv(corr, self.convc1, relu=True)
        corflo = e.new(flow.N, flow.H, flow.W, 160)
        # relu_in: cor1 / flo1 / corflo are ReLU outputs with one consumer each -- their ReLU backward rides in the consumer's data gradient
        e.conv(cor1, self.convc2, out=corflo.slice(0, 96), relu=True, relu_in=True)
        flo1 = e.conv(flow, self.convf1, relu=True)
        e.conv(flo1, self.convf2, out=corflo.slice(96, 160), relu=True, relu_in=True)
        e.conv(corflo, self.conv, out=out.slice(0, 126), relu=True, relu_in=True)
        e.copy(flow, out=out.slice(126, 128))

    def forward(self, delta_flow, corr):
        """(B,2,r,r), (B,98,r,r) -> (B,128,r,r) = [conv(126) | delta_flow].  reference: modules/raft.py:60-68"""
        def program(e: Ctx, flow_, corr_):
            fv = e.from_nchw(flow_)
            cv = e.from_nchw(corr_, out=e.new(corr_.shape[0], corr_.shape[2], corr_.shape[3], corr_.shape[1], pad32=True))
            out = e.new(fv.N, fv.H, fv.W, 128)
            self.run(e, fv, cv, out)
            return (e.to_nchw(out),), (lambda g: e.seed_grad_nchw(out, g),), \
                ((lambda: e.grad_to_nchw(fv) if fv.has_grad else None), (lambda: e.grad_to_nchw(cv) if cv.has_grad else None))
        return run_program(self, program, [delta_flow, corr])[0]


class RefineFlow(nn.Module):
    """The stateless update operator (NOT a GRU).  reference: modules/raft.py:70-88"""

    def __init__(self):
        super().__init__()
        self.convc1 = nn.Conv2d(192, 128, 3, padding=1)
        self.conv1 = nn.Conv2d(256, 128, 3, padding=1)
        self.conv2 = nn.Conv2d(128, 2, 3, padding=1)
        self.convo1 = nn.Conv2d(256, 128, 3, padding=1)
        self.convo2 = nn.Conv2d(128, 1, 3, padding=1)

    def run(self, e: Ctx, inp: View, ctx: View, ctx_relu=False) -> View:
        """inp: 256-channel buffer whose first 128 channels hold the motion features; returns d_flow (.,.,.,3).
        ctx_relu: ctx is a ReLU output consumed only here (the to_context conv of RaftFlow.run): see Ctx.conv's relu_in"""
        e.conv(ctx, self.convc1, out=inp.slice(128, 256), relu=True, relu_in=ctx_relu)
        h1 = e.conv(inp, self.conv1, relu=True)
        d = e.new(inp.N, inp.H, inp.W, 3)
        e.conv(h1, self.conv2, out=d.slice(0, 2), relu_in=True)
        h2 = e.conv(inp, self.convo1, relu=True)
        e.conv(h2, self.convo2, out=d.slice(2, 3), relu_in=True)
        return d

    def forward(self, m_f, warp_f):
        """(B,128,r,r) motion features, (B,192,r,r) context -> (out (B,3,r,r) = [d_flow(2) | d_occlusion(1)], inp (B,256,r,r)).
        reference: modules/raft.py:80-88"""
        def program(e: Ctx, mf_, wf_):
            inp = e.new(mf_.shape[0], mf_.shape[2], mf_.shape[3], 256)
            mv = e.from_nchw(mf_, out=inp.slice(0, 128))
            wv = e.from_nchw(wf_)
            d = self.run(e, inp, wv)
            return (e.to_nchw(d), e.to_nchw(inp)), ((lambda g: e.seed_grad_nchw(d, g)), (lambda g: e.seed_grad_nchw(inp, g))), \
                ((lambda: e.grad_to_nchw(mv) if mv.has_grad else None), (lambda: e.grad_to_nchw(wv) if wv.has_grad else None))
        out, inp = run_program(self, program, [m_f, warp_f])
        return out, inp


class _CorrVolume:
    """All-pairs structure correlation (raft.py:183-185) as two batched MFMA GEMMs: level 0 against k_s and level 1
    against the 2x2-pooled k_s (avg-pooling the volume over its source dims == correlating with pooled keys).
    Gradients: lookup backward scatters into dense dvol buffers; dq = dvol.k, dk = dvol^T.q are GEMMs again."""

    def __init__(self, e: Ctx, q: View, k: View, kpool: View, scale: float):
        self.e, self.q, self.k, self.kp, self.scale = e, q, k, kpool, scale
        B, Q, S0, S1, D = q.N, q.H * q.W, k.H * k.W, kpool.H * kpool.W, q.C
        self.B, self.Q, self.S0, self.S1, self.D = B, Q, S0, S1, D
        self.vol0 = torch.empty((B * Q, S0), dtype=torch.float32, device=e.dev)
        self.vol1 = torch.empty((B * Q, S1), dtype=torch.float32, device=e.dev)
        e.gemm_nt(q.ptr, q.ld, k.ptr, k.ld, self.vol0.data_ptr(), S0, Q, S0, D, scale, B, Q * q.ld, S0 * k.ld, Q * S0)
        e.gemm_nt(q.ptr, q.ld, kpool.ptr, kpool.ld, self.vol1.data_ptr(), S1, Q, S1, D, scale, B, Q * q.ld, S1 * kpool.ld, Q * S1)
        self.dvol0 = self.dvol1 = None
        if e.record:
            e.tape.append(self._bwd)

    def dvols(self):
        if self.dvol0 is None:
            self.dvol0 = torch.zeros_like(self.vol0)
            self.dvol1 = torch.zeros_like(self.vol1)
        return self.dvol0, self.dvol1

    def _bwd(self):
        if self.dvol0 is None:
            return
        e, q, k, kp = self.e, self.q, self.k, self.kp
        B, Q, S0, S1, D = self.B, self.Q, self.S0, self.S1, self.D
        for dvol, kk, S in ((self.dvol0, k, S0), (self.dvol1, kp, S1)):
            # dq[b,i,:] += scale * sum_j dvol[b,i,j] k[b,j,:]   -> NT GEMM against k^T (D x S per batch)
            kt = torch.empty((B, D, S), dtype=torch.float32, device=e.dev)
            e._chk(e.L.mrfa_nhwc_to_nchw(e.s, kk.ptr, kk.ld, kt.data_ptr(), B, D, 1, S, 0), "transpose(k)")
            e.gemm_nt(dvol.data_ptr(), S, kt.data_ptr(), S, q.gptr, q.ld, Q, D, S, self.scale, B, Q * S, D * S, Q * q.ld, accumulate=True)
            # dk[b,j,:] += scale * sum_i dvol[b,i,j] q[b,i,:]   -> TN GEMM (atomics into a dense temp, then added)
            tmp = torch.zeros((B, S, D), dtype=torch.float32, device=e.dev)
            e.gemm_tn_acc(dvol.data_ptr(), S, q.ptr, q.ld, tmp.data_ptr(), S, D, Q, self.scale, B, Q * S, Q * q.ld, S * D)
            e._chk(e.L.mrfa_copy_view(e.s, tmp.data_ptr(), D, B * S, D, kk.gptr, kk.ld, 1.0, 1), "dk")
        self.dvol0 = self.dvol1 = None


_GRID_CONSTS: dict = {}


def _grid_const(e: Ctx, n: int, h: int, w: int, scale: float, offset: float, like: torch.Tensor) -> View:
    """(n, h, w, 2) constant `scale * (x, y) pixel grid + offset` as an engine View (kept per key; built outside hipGraph captures, like coords_grid_nhwc's grids):
    the coordinate terms of RaftFlow enter the fused copies as constants instead of through ATen multiply / add launches"""
    key = (n, h, w, float(scale), float(offset), like.device)
    t = _GRID_CONSTS.get(key)
    if t is None:
        t = (coords_grid_nhwc(h, w, like) * scale + offset).expand(n, h, w, 2).contiguous().view(n * h * w, 2)
        if not (like.is_cuda and torch.cuda.is_current_stream_capturing()):
            _GRID_CONSTS[key] = t
    return View(Storage(t), n, h, w, 2)


class RaftFlow(nn.Module):
    """Same kwargs / state_dict / forward signature as the reference (raft.py:92-141):
    forward(kp_s, kp_d, dense_motion, img, img_full) -> (out, warp_img, occlusion_strip)."""

    def __init__(self, prior_only=False, num_kp=10, dim=256, size=256, generator=None, driving_encoder=None, source_encoder=None):
        super().__init__()
        self.scale = dim ** -0.5
        self.size = size
        self.h = size // 4
        self.w = size // 4
        self.prior_only = prior_only
        self.generator = OcclusionAwareGenerator(**generator)
        channels = {size // 32: 512, size // 16: 512, size // 8: 512, size // 4: 256, size // 2: 128, size: 64}
        self.total_iter = int(math.log(2 ** 5, 2)) + 1
        self.num_iter = int(math.log(2 ** 5, 2)) + 1
        self.basic_res_index = int(math.log(self.h // (size // 32), 2))
        if not self.prior_only:
            self.kp = Hourglass(**driving_encoder)
            self.kp_img = Hourglass(**source_encoder)
            self.kp_head = nn.Conv2d(self.kp.out_filters, dim, kernel_size=1, padding=0)
            self.kp_img_head = nn.Conv2d(self.kp_img.out_filters, dim, kernel_size=1, padding=0)
            self.pos_embedding = nn.Parameter(torch.zeros(1, num_kp, self.h, self.w))
            nn.init.trunc_normal_(self.pos_embedding, std=.02)
            self.corr_enc = BasicMotionEncoder()
            self.refine = RefineFlow()
            self.to_context = nn.ModuleList()
            for i in range(self.num_iter):
                self.to_context.append(nn.Conv2d(channels[(size // 32) * (2 ** i)], 192, 1, padding=0))

    # ------------------------------------------------------------------------------------------------------------
    def _strip(self, e: Ctx, occs: List[View]) -> torch.Tensor:
        """visualisation strip (B,1,size,len*size) (raft.py:170-172,305-309); detached (no caller differentiates it)"""
        with torch.no_grad():
            maps = [F.interpolate(o.tensor().permute(0, 3, 1, 2), size=self.size, mode='bilinear', align_corners=True) for o in occs]
            return torch.cat(maps, dim=3)

    def encode_source(self, kp_s, img, img_full):
        """Source-only half of the forward (inference): the generator's 6-level feature pyramid of the source image and
        the source structure keys k_s / pooled k_s (raft.py:143,179,181) -- ~38 GF per frame that an animation loop over
        ONE source re-computes for every driving frame (demo.py:47-73).  Returns an opaque cache for forward(source_cache=)."""
        assert not self.training, "source caching is an inference feature (eval-mode BatchNorm)"
        with torch.no_grad():
            e = Ctx(img_full.device, train=False, record=False)
            gen = self.generator
            imgf = e.from_nchw(img_full)
            cache = {"imgf": imgf, "feature": gen.run_encode(e, imgf), "shape": tuple(img_full.shape)}
            if not self.prior_only:
                h, w = img.shape[2], img.shape[3]
                in_s = self._source_input(e, kp_s, img, h, w)
                k_s = e.conv(self.kp_img.run(e, in_s), self.kp_img_head)
                cache["k_s"], cache["k_pool"] = k_s, e.avgpool2(k_s)
            e.flush_forward()
        return cache

    def _source_input(self, e: Ctx, kp_s, img, h, w) -> View:
        """[heat-maps(kp_s) + pos_embedding (K) | 1/4-scale source image (3)] as one NHWC buffer (raft.py:177,179)"""
        K = kp_s.shape[1]
        in_s = e.new(kp_s.shape[0], h, w, K + img.shape[1])
        e.kp_gaussian(kp_s, 0.1, in_s.slice(0, K), pos=self.pos_embedding)
        e.from_nchw(img, out=in_s.slice(K, K + img.shape[1]))
        return in_s

    def _program(self, e: Ctx, kp_s, kp_d, deformation, occlusion, img, img_full, cache=None):
        gen = self.generator
        if cache is not None:
            assert not e.record and cache["shape"] == tuple(img_full.shape), "source cache: inference only, same source batch"
            imgf, feature = cache["imgf"], cache["feature"]
        else:
            imgf = e.from_nchw(img_full)
            feature = gen.run_encode(e, imgf)
        b, h, w = img.shape[0], img.shape[2], img.shape[3]
        size = self.size
        deform = e.wrap_nhwc(deformation.contiguous())                          # (B,h,w,2) normalised sampling grid
        prior_occ = e.wrap_nhwc(occlusion.contiguous().view(b, h, w, 1))        # logits
        in_grads = (None, None, (lambda: deform.st.grad.view(b, h, w, 2) if deform.has_grad else None),
                    (lambda: prior_occ.st.grad.view(b, 1, h, w) if prior_occ.has_grad else None), None, None)

        if self.prior_only:                                                     # raft.py:156-173
            warp_f, occs = [], []
            grid_res = None
            for i in range(self.total_iter):
                f = feature[i]
                if deform.H != f.H:
                    grid_res = e.resize(deform, f.H, f.W)
                    occ_res = e.resize(prior_occ, f.H, f.W)
                else:
                    grid_res, occ_res = deform, prior_occ
                warp_f.append(e.grid_sample(f, grid_res, 0))
                occs.append(e.act(occ_res, 2))
            warp_img = e.grid_sample(imgf, grid_res, 0, need_din=False)
            out = gen.run_decode(e, warp_f, warp_img, occs, None)
            outs = (e.to_nchw(out), e.to_nchw(warp_img), self._strip(e, occs))
            seed = ((lambda g: e.seed_grad_nchw(out, g)), (lambda g: e.seed_grad_nchw(warp_img, g)), None)
            return outs, seed, in_grads

        # ---- structure encoders + correlation volumes (raft.py:177-185)
        pos = self.pos_embedding

        # K15 (csrc/prior.hip): Gaussian heat-maps of the keypoints + pos_embedding, written as NHWC (raft.py:177-178)
        in_d = e.kp_gaussian(kp_d, 0.1, e.new(b, h, w, kp_d.shape[1]), pos=pos)
        if cache is not None:
            k_s, k_pool = cache["k_s"], cache["k_pool"]
            fe_d = self.kp.run(e, in_d)
            q_d = e.conv(fe_d, self.kp_head)
        else:
            in_s = self._source_input(e, kp_s, img, h, w)
            # (round 4 ran the source hourglass as a parallel branch of the captured graph: 0.6 ms, and the ninth capture of one process died inside
            # the HIP runtime with that branch in the graph -- removed in round 5)
            k_s = e.conv(self.kp_img.run(e, in_s), self.kp_img_head)            # (B,h,w,dim)
            k_pool = e.avgpool2(k_s)
            fe_d = self.kp.run(e, in_d)
            q_d = e.conv(fe_d, self.kp_head)
        in_grads = ((lambda: e.ext_grads.get(id(kp_s))), (lambda: e.ext_grads.get(id(kp_d)))) + in_grads[2:]
        base = self.basic_res_index
        q_levels = {base: q_d}
        for i in range(base - 1, -1, -1):                                       # pooled queries == volume pooled over driving dims
            q_levels[i] = e.avgpool2(q_levels[i + 1])
        vols = {i: _CorrVolume(e, q_levels[i], k_s, k_pool, self.scale) for i in range(base + 1)}

        # ---- prior initialisation (raft.py:189-206)
        with e.fused_resizes():                                                 # init_flow = (h-1)/2 (deform + 1) - identity grid, one launch
            init_flow = e.copy(deform, mul=(h - 1) / 2.0)
            e.copy(_grid_const(e, b, h, w, -1.0, (h - 1) / 2.0, img_full), out=init_flow, acc=True, nograd=True)
        r0 = size // 32
        with e.fused_resizes():
            flow = e.resize(init_flow, r0, r0, mul=1.0 / 8.0)
            occ = e.resize(prior_occ, r0, r0)

        # decode concat buffers [blend | coarse warp]
        lv_c = gen.level_channels()
        cats = [e.new(b, r0 * (2 ** i), r0 * (2 ** i), 2 * lv_c[i]) for i in range(gen.num_up_blocks)]

        out_warp_f, out_occ = [], []
        d_f_pre = d_occ_pre = None
        for i in range(self.total_iter):
            r = r0 * (2 ** i)
            f = feature[i]
            if i < base:
                cscale, vol, rq = float(2 ** (base - i)), vols[i], r
                flow_q = flow
            else:
                cscale, vol, rq = 1.0, vols[base], h
                flow_q = e.resize(flow, h, w, mul=0.5 ** (i - base)) if i > base else flow
            with e.fused_resizes():                                             # coords = cscale (flow + identity grid), one launch
                coords = e.copy(flow_q, mul=cscale)
                e.copy(_grid_const(e, b, rq, rq, cscale, 0.0, img_full), out=coords, acc=True, nograd=True)
            cfeat = e.corr_lookup(vol.vol0, vol.vol1, vol.dvols, h, w, coords)
            if i > base:
                cfeat = e.resize(cfeat, r, r)
            inp = e.new(b, r, r, 256)
            self.corr_enc.run(e, flow, cfeat, inp)
            ctx = e.grid_sample(f, flow, 1)
            ctx = e.conv(ctx, self.to_context[i], relu=True)
            d_flow = self.refine.run(e, inp, ctx, ctx_relu=True)
            # the running-flow / occlusion updates (raft.py:258-262), the coarse grid of this level and the re-composition for the next one (raft.py:276-295):
            # ~16 copies / resizes of 1- and 2-channel maps, independent of each other -> ONE launch (engine.Ctx.fused_resizes), their backward another
            grid_c = None
            with e.fused_resizes():
                flow_w = e.copy(flow)
                e.copy(d_flow.slice(0, 2), out=flow_w, acc=True)
                occ_new = e.copy(occ)
                e.copy(d_flow.slice(2, 3), out=occ_new, acc=True)
                if i < gen.num_up_blocks:
                    grid_c = e.resize(deform, r, r) if i != base else deform
                if i < self.num_iter - 1:                                           # raft.py:276-295
                    r2 = r * 2
                    sc = 2 ** (base - i) / 2.0
                    nflow = e.resize(d_flow.slice(0, 2), r2, r2, mul=2.0)
                    e.resize(init_flow, r2, r2, mul=1.0 / sc, out=nflow, acc=True)
                    nocc = e.resize(d_flow.slice(2, 3), r2, r2)
                    e.resize(prior_occ, r2, r2, out=nocc, acc=True)
                    if i == 0:
                        d_f_pre = e.resize(d_flow.slice(0, 2), r2, r2, mul=2.0)
                        d_occ_pre = e.resize(d_flow.slice(2, 3), r2, r2)
                    else:
                        e.resize(d_f_pre, r2, r2, mul=2.0, out=nflow, acc=True)
                        e.resize(d_occ_pre, r2, r2, out=nocc, acc=True)
                        nd = e.resize(d_flow.slice(0, 2), r2, r2, mul=2.0)
                        e.resize(d_f_pre, r2, r2, mul=2.0, out=nd, acc=True)
                        no = e.resize(d_flow.slice(2, 3), r2, r2)
                        e.resize(d_occ_pre, r2, r2, out=no, acc=True)
                        d_f_pre, d_occ_pre = nd, no
            out_warp_f.append(e.grid_sample(f, flow_w, 1))
            out_occ.append(e.act(occ_new, 2))
            # coarse (prior-motion) warp straight into its decode concat slot (raft.py:265-272); level 5's is never read
            if i < gen.num_up_blocks:
                e.grid_sample(f, grid_c, 0, out=cats[i].slice(lv_c[i], 2 * lv_c[i]))
            if i < self.num_iter - 1:
                flow, occ = nflow, nocc
        # NB: the image is warped with the last level's INPUT flow, not flow_w (raft.py:302)
        warp_img = e.grid_sample(imgf, flow, 1, need_din=False)
        out = gen.run_decode(e, out_warp_f, warp_img, out_occ, cats)
        strip = self._strip(e, out_occ + [e.act(prior_occ, 2)])
        outs = (e.to_nchw(out), e.to_nchw(warp_img), strip)
        seed = ((lambda g: e.seed_grad_nchw(out, g)), (lambda g: e.seed_grad_nchw(warp_img, g)), None)
        return outs, seed, in_grads

    def forward(self, kp_s, kp_d, dense_motion, img, img_full, source_cache=None):
        """source_cache (extension, inference only): the result of encode_source(kp_s, img, img_full) for this source"""
        if img is None:
            raise ValueError("RaftFlow.forward needs `img` (the 1/4-resolution source); the reference crashes on None too "
                             "(raft.py:144-145 uses a commented-out self.down)")
        ins = [kp_s, kp_d, dense_motion['deformation'], dense_motion['occlusion'], img, img_full]
        program = self._program if source_cache is None else functools.partial(self._program, cache=source_cache)
        out, warp_img, strip = run_program(self, program, ins)
        return out, warp_img, strip
