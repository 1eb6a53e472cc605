"""CPU oracle for the MTIA prior (TokenPose_B)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional restatement (plain torch ops over a flat {name: tensor} state-dict, dtype taken from the input so an fp64
"truth" run is possible) of  modules/transformer/pose_tokenpose_b.py:16-50  =  HRNET_base (hr_base.py:294-450: stem,
layer1, transition1, stage2, transition2, stage3 with HighResolutionModule 120-289, BasicBlock 26-54, Bottleneck 57-95)
followed by TokenPose_TB_base (tokenpose_base.py:230-468: patch tokens, keypoint / Jacobian tokens, sine position code,
Transformer 137-158 with Attention 60-94 and FeedForward 46-58, heads).  The network shape is read off the parameter
names, so any STAGE2 / STAGE3 configuration of the reference works.

Only tests/, tools/ and bench.py's cpu_baseline leg may import this file, as the checker.

Parity pin: tests/golden/tokenpose.npz, written by tools/make_goldens.py from the unmodified reference imported in the
build container on the deterministic case of tests/cases.py (tokenpose_*); tests/test_oracle_golden.py checks this
file against it.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List

import torch
import torch.nn.functional as F

from .mrfa_oracle import batchnorm

Params = Dict[str, torch.Tensor]


def _w(P: Params, name: str, like: torch.Tensor) -> torch.Tensor:
    return P[name].to(like.dtype)


def _count(P: Params, pfx: str) -> int:
    """number of integer-indexed children under `pfx` (pfx = 'a.b.' -> a.b.0, a.b.1, ...)"""
    idx = {int(m.group(1)) for k in P for m in [re.match(re.escape(pfx) + r"(\d+)\.", k)] if m}
    return max(idx) + 1 if idx else 0


def conv_bn(x, P, conv, bn, train, relu, stride=1, update_stats=True):
    """bias-free conv (3x3 pad 1 or 1x1 pad 0) + BatchNorm2d (+ReLU)"""
    w = _w(P, conv + ".weight", x)
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = batchnorm(y, P, bn, train, update_stats=update_stats and train)
    if train and update_stats and (bn + ".num_batches_tracked") in P:
        P[bn + ".num_batches_tracked"] += 1
    return F.relu(y) if relu else y


def residual_block(x, P, pfx, train):
    """BasicBlock (conv1/bn1, conv2/bn2; hr_base.py:39-54) or Bottleneck (+conv3/bn3; hr_base.py:76-95), told apart by
    the presence of conv3; optional 1x1 conv + BN projection of the skip path ('downsample')"""
    skip = x
    if (pfx + ".downsample.0.weight") in P:
        skip = conv_bn(x, P, pfx + ".downsample.0", pfx + ".downsample.1", train, relu=False)
    y = conv_bn(x, P, pfx + ".conv1", pfx + ".bn1", train, relu=True)
    if (pfx + ".conv3.weight") in P:
        y = conv_bn(y, P, pfx + ".conv2", pfx + ".bn2", train, relu=True)
        y = conv_bn(y, P, pfx + ".conv3", pfx + ".bn3", train, relu=False)
    else:
        y = conv_bn(y, P, pfx + ".conv2", pfx + ".bn2", train, relu=False)
    return F.relu(y + skip)


def block_chain(x, P, pfx, train):
    for i in range(_count(P, pfx + ".")):
        x = residual_block(x, P, f"{pfx}.{i}", train)
    return x


def hr_module(xs: List[torch.Tensor], P, pfx, train) -> List[torch.Tensor]:
    """HighResolutionModule.forward, hr_base.py:270-289: per-branch block chains, then out_i = relu(sum_j f_ij(x_j)) with
    f_ij = identity (j == i) | 1x1 conv + BN + nearest upsample by 2^(j-i) (j > i) | (i-j) stride-2 3x3 conv + BN, ReLU
    between them (j < i)"""
    nb = _count(P, pfx + ".branches.")
    xs = [block_chain(xs[b], P, f"{pfx}.branches.{b}", train) for b in range(nb)]
    if nb == 1:
        return xs
    outs = []
    for i in range(_count(P, pfx + ".fuse_layers.")):
        total = None
        for j in range(nb):
            f = f"{pfx}.fuse_layers.{i}.{j}"
            if j == i:
                t = xs[j]
            elif j > i:
                t = conv_bn(xs[j], P, f + ".0", f + ".1", train, relu=False)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = xs[j]
                for k in range(i - j):
                    t = conv_bn(t, P, f"{f}.{k}.0", f"{f}.{k}.1", train, relu=(k != i - j - 1), stride=2)
            total = t if total is None else total + t
        outs.append(F.relu(total))
    return outs


def _transition(x, P, pfx, train):
    """transition<k>.<i>: absent (identity), (conv3x3, BN, ReLU), or a chain of stride-2 (conv3x3, BN, ReLU).
    hr_base.py:331-373"""
    if (pfx + ".0.weight") in P:
        return conv_bn(x, P, pfx + ".0", pfx + ".1", train, relu=True)
    for k in range(_count(P, pfx + ".")):
        x = conv_bn(x, P, f"{pfx}.{k}.0", f"{pfx}.{k}.1", train, relu=True, stride=2)
    return x


def _has(P, pfx):
    return any(k.startswith(pfx + ".") for k in P)


def hrnet(x, P, pfx, train):
    """HRNET_base.forward, hr_base.py:426-450 -> (B, C0, H/4, W/4)"""
    y = conv_bn(x, P, pfx + ".conv1", pfx + ".bn1", train, relu=True, stride=2)
    y = conv_bn(y, P, pfx + ".conv2", pfx + ".bn2", train, relu=True, stride=2)
    y = block_chain(y, P, pfx + ".layer1", train)
    nb2 = _count(P, pfx + ".stage2.0.branches.")
    xs = [_transition(y, P, f"{pfx}.transition1.{i}", train) if _has(P, f"{pfx}.transition1.{i}") else y for i in range(nb2)]
    for m in range(_count(P, pfx + ".stage2.")):
        xs = hr_module(xs, P, f"{pfx}.stage2.{m}", train)
    nb3 = _count(P, pfx + ".stage3.0.branches.")
    xs = [_transition(xs[-1], P, f"{pfx}.transition2.{i}", train) if _has(P, f"{pfx}.transition2.{i}") else xs[i] for i in range(nb3)]
    for m in range(_count(P, pfx + ".stage3.")):
        xs = hr_module(xs, P, f"{pfx}.stage3.{m}", train)
    return xs[0]


def sine_position_code(h: int, w: int, d_model: int, temperature: float = 10000.0, scale: float = 2 * math.pi) -> torch.Tensor:
    """(1, h*w, d_model): first half codes the row, second half the column, sin / cos interleaved over frequencies
    temperature^(2*floor(k/2)/half); positions are 1-based and normalised by (size + 1e-6).  tokenpose_base.py:340-362"""
    half = d_model // 2
    k = torch.arange(half, dtype=torch.float32)
    freq = temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / half)

    def code(n):
        a = (torch.arange(1, n + 1, dtype=torch.float32) / (n + 1e-6) * scale)[:, None] / freq
        return torch.stack((a[:, 0::2].sin(), a[:, 1::2].cos()), dim=2).flatten(1)
    py = code(h)[:, None, :].expand(h, w, half)
    px = code(w)[None, :, :].expand(h, w, half)
    return torch.cat((py, px), dim=2).reshape(1, h * w, d_model)


def layernorm(x, P, pfx):
    return F.layer_norm(x, (x.shape[-1],), _w(P, pfx + ".weight", x), _w(P, pfx + ".bias", x), 1e-5)


def linear(x, P, pfx):
    b = P.get(pfx + ".bias")
    return F.linear(x, _w(P, pfx + ".weight", x), None if b is None else b.to(x.dtype))


def attention(x, P, pfx, heads):
    """tokenpose_base.py:72-94 (scale_with_head=True: (dim/heads)^-0.5; no mask)"""
    b, n, dim = x.shape
    d = dim // heads
    q, k, v = [t.reshape(b, n, heads, d).transpose(1, 2) for t in linear(x, P, pfx + ".to_qkv").chunk(3, dim=-1)]
    att = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    return linear((att @ v).transpose(1, 2).reshape(b, n, dim), P, pfx + ".to_out.0")


def token_transformer(feature, P, pfx, heads=8, patch=(4, 4), pos_type="sine-full", jacobian_token=True):
    """TokenPose_TB_base.forward, tokenpose_base.py:406-468 (spatial_kp_head / hidden_dim / affine_jacobian off)"""
    b, c, H, W = feature.shape
    p1, p2 = patch
    h, w = H // p1, W // p2
    x = feature.reshape(b, c, h, p1, w, p2).permute(0, 2, 4, 3, 5, 1).reshape(b, h * w, p1 * p2 * c)      # b (h w) (p1 p2 c)
    x = linear(x, P, pfx + ".patch_to_embedding")
    n = x.shape[1]
    tok = _w(P, pfx + ".keypoint_token", x).expand(b, -1, -1)
    nk = tok.shape[1]
    pos = P.get(pfx + ".pos_embedding")
    pos = None if pos is None else pos.to(x.dtype)
    if pos_type in ("sine", "sine-full"):
        x = torch.cat((tok, x + pos[:, :n]), dim=1)
    elif pos_type == "none":
        x = torch.cat((tok, x), dim=1)
    else:
        x = torch.cat((tok, x), dim=1) + pos[:, :n + nk]
    for i in range(_count(P, pfx + ".transformer.layers.")):
        L = f"{pfx}.transformer.layers.{i}"
        if i > 0 and pos_type == "sine-full":                      # position code re-added to the image tokens (:154-155)
            x = torch.cat((x[:, :nk], x[:, nk:] + pos), dim=1)
        x = x + attention(layernorm(x, P, L + ".0.fn.norm"), P, L + ".0.fn.fn", heads)
        y = layernorm(x, P, L + ".1.fn.norm")
        x = x + linear(F.gelu(linear(y, P, L + ".1.fn.fn.net.0")), P, L + ".1.fn.fn.net.3")
    k_tok = x[:, :nk // 2] if jacobian_token else x[:, :nk]
    out = {"kp": 2 * torch.sigmoid(linear(layernorm(k_tok, P, pfx + ".mlp_head.0"), P, pfx + ".mlp_head.1")) - 1}
    if (pfx + ".mlp_head_jacobian.1.weight") in P:
        j_tok = x[:, nk // 2:nk] if jacobian_token else k_tok
        jac = linear(layernorm(j_tok, P, pfx + ".mlp_head_jacobian.0"), P, pfx + ".mlp_head_jacobian.1")
        out["jacobian"] = jac.reshape(b, -1, 2, 2)
    return out


def tokenpose_b(x, P, pfx="", train=False, heads=8, patch=(4, 4), pos_type="sine-full"):
    """TokenPose_B.forward, pose_tokenpose_b.py:39-50 (DATA_PREPROCESS False)"""
    pre = pfx + "." if pfx else ""
    return token_transformer(hrnet(x, P, pre + "pre_feature", train), P, pre + "transformer", heads, patch, pos_type)
