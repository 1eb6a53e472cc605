"""Streaming animation of ONE source by many driving frames (the reference's make_animation loop, demo.py:47-73 /
animate_ddp.py:88-105) with everything that depends only on the source computed once: KPDetector(source), the 1/4-scale
source, the generator's feature pyramid and the source structure keys (~38 GF of 375 per frame and
one encoder pass).  Optionally the per-frame program is replayed as a hipGraph."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn


class Animator:
    def __init__(self, model: nn.Module, graph: bool = False):
        """model: mrfa_amd.train.HotPath or mrfa_amd.modules.model.MRFA (attributes encoder / dense_motion / decoder / down)"""
        self.m = model.eval()
        self.use_graph = graph
        self.source = None
        self._g: Optional[torch.cuda.CUDAGraph] = None

    @torch.no_grad()
    def set_source(self, source: torch.Tensor):
        m = self.m
        self.source = source
        self.kp_s = m.encoder(source)
        self.img_down = m.down(source)
        self.cache = m.decoder.encode_source(self.kp_s["kp"], self.img_down, source)
        self._g = None

    @torch.no_grad()
    def _frame(self, driving):
        m = self.m
        kp_d = m.encoder(driving)
        dm = m.dense_motion(self.source, kp_d, self.kp_s)
        out, _, _ = m.decoder(self.kp_s["kp"], kp_d["kp"], dm, img=self.img_down, img_full=self.source, source_cache=self.cache)
        return out

    @torch.no_grad()
    def __call__(self, driving: torch.Tensor) -> torch.Tensor:
        assert self.source is not None, "call set_source(source) first"
        if not self.use_graph:
            return self._frame(driving)
        if self._g is None:                                   # capture the per-frame program once per source
            from . import graph_replay_safe
            graph_replay_safe("Animator(graph=True)")
            self._drv = driving.clone()
            eager = self._frame(self._drv).clone()            # packs / tables outside the graph
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._out = self._frame(self._drv)
            self._g = g
            for k in range(3):                                # first AND later replays against the eager frame (mrfa_amd/graph.py)
                g.replay()
                torch.cuda.synchronize()
                diff = (self._out - eager).abs()
                d, dm = float(diff.max()), float(diff.mean())
                # run-to-run summation-order noise (split-K atomics) reaches ~1e-4 on single border pixels of sharply warped frames;
                # a mis-ordered graph is wrong everywhere (stale or zero inputs): gate the mean tightly, the max loosely
                if not (dm <= 2e-5 and d <= 5e-3):
                    self._g = None
                    raise RuntimeError(f"Animator: hipGraph replay {k} differs from the eager frame (max |diff| {d:.3e}, mean {dm:.3e})")
        self._drv.copy_(driving)
        self._g.replay()
        return self._out


# ----------------------------------------------------------------------------------------------- callers of the path
def _hull_area(pts: torch.Tensor) -> torch.Tensor:
    """Area of the convex hull of (N,2) points, on the device and without a host round trip (the reference calls
    scipy.spatial.ConvexHull(...).volume on the host, animate_ddp.py:20-21): a directed edge (i,j) belongs to the
    counter-clockwise hull iff no point lies strictly to its right; the shoelace sum over those edges is the area.  N = 10:
    900 cross products.  Points in general position (no three hull points collinear), as qhull assumes after joggling."""
    d = pts[None, :, :] - pts[:, None, :]                                   # d[i,j] = p_j - p_i
    rel = pts[None, None, :, :] - pts[:, None, None, :]                     # rel[i,.,k] = p_k - p_i
    cross = d[:, :, None, 0] * rel[:, :, :, 1] - d[:, :, None, 1] * rel[:, :, :, 0]       # (i,j,k)
    scale = pts.abs().max().clamp_min(1e-12) ** 2
    on_hull = (cross >= -1e-7 * scale).all(dim=2) & ~torch.eye(pts.shape[0], dtype=torch.bool, device=pts.device)
    shoelace = pts[:, None, 0] * pts[None, :, 1] - pts[None, :, 0] * pts[:, None, 1]      # p_i x p_j
    return 0.5 * (shoelace * on_hull).sum()


def _inv2x2(m: torch.Tensor) -> torch.Tensor:
    a, b, c, d = m[..., 0, 0], m[..., 0, 1], m[..., 1, 0], m[..., 1, 1]
    det = a * d - b * c
    return torch.stack([torch.stack([d, -b], dim=-1), torch.stack([-c, a], dim=-1)], dim=-2) / det[..., None, None]


def normalize_kp(kp_source, kp_driving, kp_driving_initial, adapt_movement_scale=False, use_relative_movement=False,
                 use_relative_jacobian=False):
    """Relative-motion transfer of an animation loop: the driving keypoints' displacement (and Jacobian change) since the
    first driving frame, applied to the source keypoints.  reference: animate_ddp.py:17-37 (same arguments and result; the
    movement scale sqrt(hull area(source)) / sqrt(hull area(driving_initial)) is taken from batch element 0, as there)."""
    scale = 1
    if adapt_movement_scale:
        scale = torch.sqrt(_hull_area(kp_source['kp'][0])) / torch.sqrt(_hull_area(kp_driving_initial['kp'][0]))
    kp_new = dict(kp_driving)
    if use_relative_movement:
        kp_new['kp'] = (kp_driving['kp'] - kp_driving_initial['kp']) * scale + kp_source['kp']
        if use_relative_jacobian:
            diff = torch.matmul(kp_driving['jacobian'], _inv2x2(kp_driving_initial['jacobian']))
            kp_new['jacobian'] = torch.matmul(diff, kp_source['jacobian'])
    return kp_new


def psnr(img1: torch.Tensor, img2: torch.Tensor):
    """20 log10(1 / sqrt(mse)) for images in [0,1].  reference: reconstruction.py:13-19"""
    mse = torch.mean((img1 - img2) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


@torch.no_grad()
def reconstruction(model: nn.Module, video: torch.Tensor, graph: bool = False):
    """The reference's reconstruction loop (reconstruction.py:52-70) on one clip: source = frame 0, driving = every frame t,
    metrics mean|out - driving| and PSNR per frame.  video: (B,3,T,H,W) in [0,1].  The source is fixed for the whole clip, so
    the source half of the path is computed once (Animator).  Returns {'prediction': (B,3,T,H,W), 'l1': [T], 'psnr': [T]}."""
    anim = Animator(model, graph=graph)
    anim.set_source(video[:, :, 0].contiguous())
    preds, l1, ps = [], [], []
    for t in range(video.shape[2]):
        driving = video[:, :, t].contiguous()
        out = anim(driving).clone()
        preds.append(out)
        l1.append(float(torch.abs(out - driving).mean()))
        ps.append(float(psnr(driving, out)))
    return {"prediction": torch.stack(preds, dim=2), "l1": l1, "psnr": ps}


@torch.no_grad()
def make_animation(model: nn.Module, source: torch.Tensor, driving_video: torch.Tensor, relative: bool = True,
                   adapt_movement_scale: bool = False, graph: bool = False):
    """demo.py:47-73 / animate_ddp.py:88-105: animate ONE source by the motion of a driving clip (B,3,T,H,W); with
    relative=True the driving keypoints go through normalize_kp against the first driving frame.  Returns (B,3,T,H,W)."""
    m = model.eval()
    kp_s = m.encoder(source)
    img_down = m.down(source)
    cache = m.decoder.encode_source(kp_s["kp"], img_down, source)
    kp_init = m.encoder(driving_video[:, :, 0].contiguous())
    outs = []
    for t in range(driving_video.shape[2]):
        kp_d = m.encoder(driving_video[:, :, t].contiguous())
        kp_n = normalize_kp(kp_s, kp_d, kp_init, adapt_movement_scale=adapt_movement_scale, use_relative_movement=relative,
                            use_relative_jacobian=relative)
        dm = m.dense_motion(source, kp_n, kp_s)
        out, _, _ = m.decoder(kp_s["kp"], kp_n["kp"], dm, img=img_down, img_full=source, source_cache=cache)
        outs.append(out.clone())
    return torch.stack(outs, dim=2)
