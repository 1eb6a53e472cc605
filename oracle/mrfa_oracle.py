"""CPU oracle for the MRFA hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement (plain torch ops over a flat {name: tensor} parameter
dict, dtype taken from the inputs so an fp64 "truth" run is possible) of the
reference path  KPDetector -> DenseMotionNetwork -> RaftFlow (correlation volume,
6-level coarse-to-fine refinement, feature warps) -> OcclusionAwareGenerator.
Every function cites the reference file:line (paths relative to /root/reference)
whose behaviour it restates.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file, and only as the checker / baseline.  The shipped path (mrfa_amd/) never
imports it and fails loudly when the HIP library is missing.

Parity pin: tests/golden/*.npz were produced by tools/make_goldens.py, which imports
the unmodified reference in the build container and records its outputs on
deterministic (mrfa_amd.utils.prng) inputs/weights; tests/test_oracle_golden.py
checks this oracle against those files (<=1e-5 abs, fp32).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- grids
def coordinate_grid(h: int, w: int, like: torch.Tensor) -> torch.Tensor:
    """(h,w,2) grid over [-1,1]^2, last dim (x,y).  modules/util.py:90-108."""
    xs = 2.0 * (torch.arange(w, dtype=like.dtype, device=like.device) / (w - 1)) - 1.0
    ys = 2.0 * (torch.arange(h, dtype=like.dtype, device=like.device) / (h - 1)) - 1.0
    return torch.stack([xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)], dim=-1)


def pixel_grid(b: int, h: int, w: int, like: torch.Tensor) -> torch.Tensor:
    """(b,2,h,w): channel 0 = x index, channel 1 = y index.  modules/util.py:53-56."""
    ys, xs = torch.meshgrid(torch.arange(h, device=like.device), torch.arange(w, device=like.device), indexing="ij")
    return torch.stack([xs, ys], dim=0).to(like.dtype)[None].expand(b, 2, h, w).contiguous()


def gaussian_heatmap(kp: torch.Tensor, h: int, w: int, variance: float) -> torch.Tensor:
    """kp (B,K,2) -> (B,K,h,w) exp(-0.5*|grid-kp|^2/var).  modules/util.py:59-87."""
    g = coordinate_grid(h, w, kp).view(1, 1, h, w, 2)
    d = g - kp.view(kp.shape[0], kp.shape[1], 1, 1, 2)
    return torch.exp(-0.5 * (d * d).sum(-1) / variance)


def antialias_kernel(scale: float, like: torch.Tensor) -> torch.Tensor:
    """k x k normalised Gaussian.  modules/util.py:286-312."""
    sigma = (1.0 / scale - 1.0) / 2.0
    k = 2 * round(sigma * 4) + 1
    t = torch.arange(k, dtype=torch.float32)
    g = torch.exp(-((t - (k - 1) / 2.0) ** 2) / (2.0 * sigma ** 2))
    ker = g[:, None] * g[None, :]
    ker = ker / ker.sum()
    return ker.to(dtype=like.dtype, device=like.device)


def antialias_down(x: torch.Tensor, scale: float, weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """zero-pad, depthwise Gaussian, nearest decimation.  modules/util.py:318-326."""
    if scale == 1.0:
        return x
    c = x.shape[1]
    ker = antialias_kernel(scale, x) if weight is None else weight[0, 0].to(x.dtype)
    k = ker.shape[-1]
    ka = k // 2
    kb = ka - 1 if k % 2 == 0 else ka
    y = F.conv2d(F.pad(x, (ka, kb, ka, kb)), ker.view(1, 1, k, k).expand(c, 1, k, k), groups=c)
    return F.interpolate(y, scale_factor=(scale, scale))          # nearest: src = floor(dst/scale)


def sample_px(img: torch.Tensor, coords_px: torch.Tensor) -> torch.Tensor:
    """Bilinear sample at PIXEL coordinates (B,Ho,Wo,2)=(x,y), zeros outside.  modules/util.py:26-38."""
    hh, ww = img.shape[-2:]
    gx = 2.0 * coords_px[..., 0] / (ww - 1) - 1.0
    gy = 2.0 * coords_px[..., 1] / (hh - 1) - 1.0
    return F.grid_sample(img, torch.stack([gx, gy], dim=-1), mode="bilinear", padding_mode="zeros", align_corners=True)


def sample_norm(img: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """F.grid_sample defaults (align_corners=False, zeros).  dense_motion.py:83, raft.py:166,168,271."""
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def resize_ac(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(mode='bilinear', align_corners=True).  raft.py:161,205,228,243,266,279-295,308."""
    if isinstance(size, int):
        size = (size, size)
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)


# --------------------------------------------------------------------------- blocks
def conv(x: torch.Tensor, P: Params, pfx: str, pad: int) -> torch.Tensor:
    return F.conv2d(x, P[pfx + ".weight"].to(x.dtype), P[pfx + ".bias"].to(x.dtype), padding=pad)


def batchnorm(x: torch.Tensor, P: Params, pfx: str, train: bool, update_stats: bool = False) -> torch.Tensor:
    """BatchNorm2d(affine=True), eps 1e-5, momentum 0.1.  modules/util.py:122,146-147,170,189,208."""
    g = P[pfx + ".weight"].to(x.dtype)
    b = P[pfx + ".bias"].to(x.dtype)
    if train:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if update_stats:
            n = x.numel() // x.shape[1]
            P[pfx + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(P[pfx + ".running_mean"].dtype))
            P[pfx + ".running_var"].mul_(1 - BN_MOMENTUM).add_(
                BN_MOMENTUM * (var.detach() * n / max(n - 1, 1)).to(P[pfx + ".running_var"].dtype))
    else:
        mean = P[pfx + ".running_mean"].to(x.dtype)
        var = P[pfx + ".running_var"].to(x.dtype)
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(1, -1, 1, 1)) * (inv * g).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def down_block(x, P, pfx, train):
    """conv3x3 -> BN -> ReLU -> avgpool2.  modules/util.py:180-197."""
    return F.avg_pool2d(F.relu(batchnorm(conv(x, P, pfx + ".conv", 1), P, pfx + ".norm", train)), 2)


def up_block(x, P, pfx, train):
    """nearest x2 -> conv3x3 -> BN -> ReLU.  modules/util.py:160-177."""
    x = F.interpolate(x, scale_factor=2)
    return F.relu(batchnorm(conv(x, P, pfx + ".conv", 1), P, pfx + ".norm", train))


def same_block(x, P, pfx, train, pad):
    """conv -> BN -> ReLU.  modules/util.py:199-214."""
    return F.relu(batchnorm(conv(x, P, pfx + ".conv", pad), P, pfx + ".norm", train))


def res_block(x, P, pfx, train):
    """BN-ReLU-conv, BN-ReLU-conv, + skip.  modules/util.py:135-157."""
    y = conv(F.relu(batchnorm(x, P, pfx + ".norm1", train)), P, pfx + ".conv1", 1)
    y = conv(F.relu(batchnorm(y, P, pfx + ".norm2", train)), P, pfx + ".conv2", 1)
    return y + x


def channel_block(x, P, pfx, train):
    """BN(2C) -> ReLU -> conv3x3 2C->C.  modules/util.py:111-133."""
    return conv(F.relu(batchnorm(x, P, pfx + ".norm1", train)), P, pfx + ".conv1", 1)


def _count(P: Params, pfx: str) -> int:
    n = 0
    while f"{pfx}.{n}.conv.weight" in P or f"{pfx}.{n}.conv1.weight" in P or f"{pfx}.{n}.weight" in P:
        n += 1
    return n


def hourglass(x, P, pfx, train):
    """U-Net encoder/decoder with skip concatenation.  modules/util.py:217-278."""
    nb = _count(P, pfx + ".encoder.down_blocks")
    feats = [x]
    for i in range(nb):
        feats.append(down_block(feats[-1], P, f"{pfx}.encoder.down_blocks.{i}", train))
    out = feats.pop()
    for i in range(nb):
        out = up_block(out, P, f"{pfx}.decoder.up_blocks.{i}", train)
        out = torch.cat([out, feats.pop()], dim=1)
    return out


# --------------------------------------------------------------------------- KPDetector
def kp_detector(x, P, pfx="", train=False, temperature=0.1, scale_factor=0.25):
    """modules/kp_detector.py:102-133 (+ gaussian2kp :90-100). Returns {'kp','jacobian'}."""
    if scale_factor != 1:
        x = antialias_down(x, scale_factor, P.get(pfx + "down.weight"))
    fmap = hourglass(x, P, pfx + "predictor", train)
    logits = conv(fmap, P, pfx + "kp", 0)
    b, k, hh, ww = logits.shape
    heat = F.softmax(logits.view(b, k, -1) / temperature, dim=2).view(b, k, hh, ww)
    grid = coordinate_grid(hh, ww, heat).view(1, 1, hh, ww, 2)
    out = {"kp": (heat.unsqueeze(-1) * grid).sum(dim=(2, 3))}
    if (pfx + "jacobian.weight") in P:
        jm = conv(fmap, P, pfx + "jacobian", 0).view(b, 1, 4, hh, ww)
        jac = (heat.unsqueeze(2) * jm).reshape(b, k, 4, -1).sum(-1)
        out["jacobian"] = jac.view(b, k, 2, 2)
    return out


# --------------------------------------------------------------------------- DenseMotionNetwork
def sparse_motions(kp_d: dict, kp_s: dict, h: int, w: int, bg_param: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,K+1,h,w,2): background (identity, or the affine bg_param applied in homogeneous coordinates, dense_motion.py:67-73)
    + J_s J_d^-1 (z - kp_d) + kp_s.  dense_motion.py:48-76."""
    kd, ks = kp_d["kp"], kp_s["kp"]
    b, k = kd.shape[:2]
    ident = coordinate_grid(h, w, kd).view(1, 1, h, w, 2)
    z = ident - kd.view(b, k, 1, 1, 2)
    if "jacobian" in kp_d:
        jac = torch.matmul(kp_s["jacobian"], torch.inverse(kp_d["jacobian"]))       # (B,K,2,2)
        z = torch.einsum("bkij,bkhwj->bkhwi", jac, z)
    d2s = z + ks.view(b, k, 1, 1, 2)
    bg = ident.expand(b, 1, h, w, 2)
    if bg_param is not None:
        hom = torch.cat([bg, torch.ones_like(bg[..., :1])], dim=-1).unsqueeze(-1)
        hom = (bg_param.to(bg).view(b, 1, 1, 1, 3, 3) @ hom).squeeze(-1)
        bg = hom[..., :2] / hom[..., 2:3]
    return torch.cat([bg, d2s], dim=1)


def dense_motion(source, kp_d: dict, kp_s: dict, P, pfx="", train=False, scale_factor=0.25, kp_variance=0.01, bg_param=None):
    """modules/dense_motion.py:104-146 (dropout off)."""
    if scale_factor != 1:
        source = antialias_down(source, scale_factor, P.get(pfx + "down.weight"))
    b, c, h, w = source.shape
    k1 = kp_d["kp"].shape[1] + 1
    heat = gaussian_heatmap(kp_d["kp"], h, w, kp_variance) - gaussian_heatmap(kp_s["kp"], h, w, kp_variance)
    heat = torch.cat([torch.zeros(b, 1, h, w, dtype=heat.dtype, device=heat.device), heat], dim=1).unsqueeze(2)
    motions = sparse_motions(kp_d, kp_s, h, w, bg_param)                         # (B,K1,h,w,2)
    src_rep = source.unsqueeze(1).expand(b, k1, c, h, w).reshape(b * k1, c, h, w)
    deformed = sample_norm(src_rep, motions.reshape(b * k1, h, w, 2)).view(b, k1, c, h, w)
    inp = torch.cat([heat, deformed], dim=2).view(b, k1 * (c + 1), h, w)
    pred = hourglass(inp, P, pfx + "hourglass", train)
    logit = conv(pred, P, pfx + "mask", 3)
    mask = F.softmax(logit, dim=1)
    deformation = (motions.permute(0, 1, 4, 2, 3) * mask.unsqueeze(2)).sum(dim=1).permute(0, 2, 3, 1)
    out = {"sparse_deformed": deformed, "logit_mask": logit, "mask": mask, "deformation": deformation}
    if (pfx + "occlusion.weight") in P:
        out["occlusion"] = conv(pred, P, pfx + "occlusion", 3)                     # logits: no sigmoid (:142-143)
    return out


# --------------------------------------------------------------------------- generator
def generator_encode(x, P, pfx, train) -> List[torch.Tensor]:
    """modules/generator.py:34-42: coarse-first list of 6 feature maps."""
    feats = [same_block(x, P, pfx + "first", train, 3)]
    for i in range(_count(P, pfx + "down_blocks")):
        feats.append(down_block(feats[-1], P, f"{pfx}down_blocks.{i}", train))
    return feats[::-1]


def generator_decode(warp_f, warp_img, occlusion, P, pfx, train, warp_f_c=None):
    """modules/generator.py:44-64 (occlusion_c is accepted upstream but never read)."""
    n_up = _count(P, pfx + "up_blocks")
    out = warp_f[0] * occlusion[0]
    if warp_f_c is not None:
        out = torch.cat([out, warp_f_c[0]], dim=1)
    for i in range(n_up):
        if warp_f_c is not None:
            out = channel_block(out, P, f"{pfx}channel_block.{i}", train)
        out = res_block(out, P, f"{pfx}resblock.{i}", train)
        out = up_block(out, P, f"{pfx}up_blocks.{i}", train)
        out = warp_f[i + 1] * occlusion[i + 1] + out * (1 - occlusion[i + 1])
        if warp_f_c is not None and i != n_up - 1:
            out = torch.cat([out, warp_f_c[i + 1]], dim=1)
    out = torch.sigmoid(conv(out, P, pfx + "final", 3))
    return out * (1 - occlusion[-1]) + warp_img * occlusion[-1]


# --------------------------------------------------------------------------- RAFT pieces
def corr_lookup(corr_maps: torch.Tensor, coords: torch.Tensor, radius: int = 3, levels: int = 2) -> torch.Tensor:
    """CorrBlock: 2-level avg-pool pyramid + (2r+1)^2 bilinear window per level.  modules/raft.py:12-48.

    corr_maps (B*h1*w1, 1, Hs, Ws): one source-space map per query pixel;  coords (B,2,h1,w1) pixel (x,y).
    Output (B, levels*(2r+1)^2, h1, w1); channel = lvl*49 + a*7 + b samples at x+(a-r), y+(b-r)  (the
    reference adds meshgrid(dy,dx) -- ij indexing -- to (x,y), raft.py:31-37).
    """
    b, _, h1, w1 = coords.shape
    q = coords.permute(0, 2, 3, 1).reshape(b * h1 * w1, 1, 1, 2)
    d = torch.linspace(-radius, radius, 2 * radius + 1, dtype=coords.dtype, device=coords.device)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * radius + 1, 2 * radius + 1, 2)
    outs = []
    level_map = corr_maps
    for lvl in range(levels):
        if lvl > 0:
            level_map = F.avg_pool2d(level_map, 2, stride=2)
        win = sample_px(level_map, q / (2 ** lvl) + delta)                     # (Q,1,7,7)
        outs.append(win.view(b, h1, w1, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous()


def motion_encoder(flow, corr, P, pfx):
    """BasicMotionEncoder.  modules/raft.py:60-68."""
    cor = F.relu(conv(corr, P, pfx + ".convc1", 0))
    cor = F.relu(conv(cor, P, pfx + ".convc2", 1))
    flo = F.relu(conv(flow, P, pfx + ".convf1", 3))
    flo = F.relu(conv(flo, P, pfx + ".convf2", 1))
    out = F.relu(conv(torch.cat([cor, flo], dim=1), P, pfx + ".conv", 1))
    return torch.cat([out, flow], dim=1)


def refine_flow(m_f, warp_f, P, pfx):
    """RefineFlow (stateless update operator, NOT a GRU).  modules/raft.py:80-88. Returns (B,3,h,w)."""
    c = F.relu(conv(warp_f, P, pfx + ".convc1", 1))
    inp = torch.cat([m_f, c], dim=1)
    flow = conv(F.relu(conv(inp, P, pfx + ".conv1", 1)), P, pfx + ".conv2", 1)
    occ = conv(F.relu(conv(inp, P, pfx + ".convo1", 1)), P, pfx + ".convo2", 1)
    return torch.cat([flow, occ], dim=1)


def raft_flow(kp_s, kp_d, dm: dict, img, img_full, P, pfx="", size=256, prior_only=False, train=False,
              dim=256, trace: Optional[dict] = None):
    """RaftFlow.forward.  modules/raft.py:141-311.  Returns (out, warp_img, occlusion_strip)."""
    gpfx = pfx + "generator."
    feature = generator_encode(img_full, P, gpfx, train)
    b, _, h, w = img.shape
    n_iter = 6
    base_idx = int(math.log2(h // (size // 32)))
    deformation = dm["deformation"]
    prior_occ = dm["occlusion"]

    if prior_only:                                                             # raft.py:156-173
        warp_f, occs = [], []
        grid_res = None
        for i in range(n_iter):
            hw = feature[i].shape[2:]
            if deformation.shape[2] != hw[0]:
                grid_res = resize_ac(deformation.permute(0, 3, 1, 2), hw)
                occ_res = resize_ac(prior_occ, hw)
            else:
                grid_res = deformation.permute(0, 3, 1, 2)
                occ_res = prior_occ
            warp_f.append(sample_norm(feature[i], grid_res.permute(0, 2, 3, 1)))
            occs.append(torch.sigmoid(occ_res))
        warp_img = sample_norm(img_full, grid_res.permute(0, 2, 3, 1))
        out = generator_decode(warp_f, warp_img, occs, P, gpfx, train)
        strip = torch.cat([resize_ac(o, size) for o in occs], dim=3)
        return out, warp_img, strip

    # structure encoders + all-pairs correlation (raft.py:177-185)
    pos = P[pfx + "pos_embedding"].to(img.dtype)
    hm_s = gaussian_heatmap(kp_s, h, w, 0.1) + pos
    hm_d = gaussian_heatmap(kp_d, h, w, 0.1) + pos
    fe_s = hourglass(torch.cat([hm_s, img], dim=1), P, pfx + "kp_img", train)
    fe_d = hourglass(hm_d, P, pfx + "kp", train)
    k_s = conv(fe_s, P, pfx + "kp_img_head", 0)
    q_d = conv(fe_d, P, pfx + "kp_head", 0)
    f_s = k_s.flatten(2).transpose(1, 2)                                       # (B, hw, C)
    f_d = q_d.flatten(2).transpose(1, 2)
    corr = torch.einsum("bic,bjc->bij", f_d, f_s) * (dim ** -0.5)              # (B, drv, src)

    # prior initialisation in pixel units at the basic (h x w) resolution (raft.py:189-191)
    init_flow = (h - 1) * (deformation.permute(0, 3, 1, 2) + 1) / 2.0 - pixel_grid(b, h, w, img)
    init_occ = prior_occ
    flow = F.interpolate(init_flow, scale_factor=1.0 / 8.0, mode="bilinear", align_corners=True) / 8.0
    occ = F.interpolate(init_occ, scale_factor=1.0 / 8.0, mode="bilinear", align_corners=True)

    # source-major view '(b n) 1 h w' used for pooling over the DRIVING dims (raft.py:208)
    corr_src_major = corr.permute(0, 2, 1).reshape(b * h * w, 1, h, w)

    out_warp_f, out_occ, out_warp_c = [], [], []
    d_flow = d_f_pre = d_occ_pre = None
    ident = None
    for i in range(n_iter):
        r = size // 32 * (2 ** i)
        ident = pixel_grid(b, r, r, img)
        flow_q, ident_q = flow, ident
        if i < base_idx:                                                       # raft.py:218-220
            f = 2 ** (base_idx - i)
            vol = F.avg_pool2d(corr_src_major, f, stride=f)
            cscale = float(f)
        else:
            vol = corr_src_major
            cscale = 1.0
            if i > base_idx:                                                   # raft.py:224-230
                flow_q = resize_ac(flow, h) * (0.5 ** (i - base_idx))
                ident_q = pixel_grid(b, h, w, img)
        # back to driving-major: one (h x w) source map per (pooled) driving pixel (raft.py:235-236)
        rq = vol.shape[-1]
        maps = vol.view(b, h * w, rq * rq).permute(0, 2, 1).reshape(b * rq * rq, 1, h, w)
        cfeat = corr_lookup(maps, (flow_q + ident_q) * cscale)
        if i > base_idx:
            cfeat = resize_ac(cfeat, flow.shape[2])
        m_f = motion_encoder(flow, cfeat, P, pfx + "corr_enc")
        ctx = sample_px(feature[i], (flow + ident).permute(0, 2, 3, 1))
        ctx = F.relu(conv(ctx, P, f"{pfx}to_context.{i}", 0))
        d_flow = refine_flow(m_f, ctx, P, pfx + "refine")
        flow_w = flow + d_flow[:, 0:2]
        d_occ = d_flow[:, 2:]
        occ = occ + d_occ
        if trace is not None:
            trace[f"flow_in_{i}"] = flow
            trace[f"cfeat_{i}"] = cfeat
            trace[f"d_flow_{i}"] = d_flow
            trace[f"occ_{i}"] = occ

        out_warp_f.append(sample_px(feature[i], (flow_w + ident).permute(0, 2, 3, 1)))
        out_occ.append(torch.sigmoid(occ))

        # coarse (prior-motion) warp of the same feature level (raft.py:265-272)
        hw = feature[i].shape[2:]
        if i != base_idx:
            grid_c = resize_ac(deformation.permute(0, 3, 1, 2), hw)
        else:
            grid_c = deformation.permute(0, 3, 1, 2)
        out_warp_c.append(sample_norm(feature[i], grid_c.permute(0, 2, 3, 1)))

        if i < n_iter - 1:                                                     # raft.py:276-295
            r2 = size // 32 * (2 ** (i + 1))
            sc = 2 ** (base_idx - i) / 2.0
            d_f = F.interpolate(d_flow[:, 0:2], scale_factor=2, mode="bilinear", align_corners=True) * 2
            flow = d_f + resize_ac(init_flow, r2) / sc
            if i == 0:
                d_f_pre = d_f
            else:
                up_pre = F.interpolate(d_f_pre, scale_factor=2, mode="bilinear", align_corners=True) * 2
                flow = flow + up_pre
                d_f_pre = d_f + up_pre
            d_o = F.interpolate(d_occ, scale_factor=2, mode="bilinear", align_corners=True)
            occ = d_o + resize_ac(init_occ, r2)
            if i == 0:
                d_occ_pre = d_o
            else:
                up_o = F.interpolate(d_occ_pre, scale_factor=2, mode="bilinear", align_corners=True)
                occ = occ + up_o
                d_occ_pre = d_o + up_o

    # NB: uses `flow` (the last level's input flow), not flow_w (raft.py:302)
    warp_img = sample_px(img_full, (flow + ident).permute(0, 2, 3, 1))
    out = generator_decode(out_warp_f, warp_img, out_occ, P, gpfx, train, warp_f_c=out_warp_c)
    vis = out_occ + [torch.sigmoid(init_occ)]
    strip = torch.cat([resize_ac(o, size) for o in vis], dim=3)
    return out, warp_img, strip


# --------------------------------------------------------------------------- MRFA wiring
def mrfa_forward(source, driving, P, size=256, prior_only=False, train=False, prior="fomm"):
    """MRFA.forward(is_train=False) wiring with the fomm (KPDetector) or mtia (TokenPose_B) prior.  modules/model.py:185-216."""
    if prior == "mtia":
        from .tokenpose_oracle import tokenpose_b
        kp_s = tokenpose_b(source, P, "encoder", train)
        kp_d = tokenpose_b(driving, P, "encoder", train)
    else:
        kp_s = kp_detector(source, P, "encoder.", train)
        kp_d = kp_detector(driving, P, "encoder.", train)
    img_down = antialias_down(source, 0.25)
    bg_param = None
    if any(k.startswith("bg_predictor.") for k in P):              # celebvhq.yaml bg_start 0: model.py:189-192
        from .losses_oracle import bg_motion_predictor
        bg_param = bg_motion_predictor(source, driving, P, "bg_predictor.", train)
    dm = dense_motion(source, kp_d, kp_s, P, "dense_motion.", train, bg_param=bg_param)
    gen, warp_img, occ = raft_flow(kp_s["kp"], kp_d["kp"], dm, img_down, source, P, "decoder.",
                                   size=size, prior_only=prior_only, train=train)
    warp_vis = torch.cat([warp_img, occ.repeat(1, 3, 1, 1)], dim=3)
    return gen, warp_vis, kp_s, kp_d, dm
