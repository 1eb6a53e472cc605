"""Deterministic, torch-RNG-independent tensor generator.

Goldens, parity tests and bench.py must produce bit-identical inputs/weights in
the build container (where the reference is imported to make the fixtures) and
on the GPU box (where only this repo exists), independent of torch / numpy RNG
versions.  Every tensor is therefore a pure function of (name, shape): a
splitmix64 hash of the element index, seeded by a 64-bit FNV-1a hash of the
name.  numpy uint64 arithmetic only.
"""
from __future__ import annotations

import math

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(name: str, n: int) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (name, index)."""
    seed = np.uint64(_fnv1a(name))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + seed)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def det_uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n) * (hi - lo) + lo
    return torch.from_numpy(u.reshape(tuple(shape))).to(dtype)


def det_normal(name: str, shape, std: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    """Box-Muller on two independent deterministic uniform streams."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = np.maximum(uniform01(name + "/bm1", n), 1e-12)
    u2 = uniform01(name + "/bm2", n)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2) * std
    return torch.from_numpy(z.reshape(tuple(shape))).to(dtype)


def fill_state_dict(sd: dict, tag: str = "w", bn_perturb: bool = True, gain: float = 1.0) -> dict:
    """Deterministic values for every entry of a state_dict-like {name: tensor}.

    conv / linear weights: U(+-gain*sqrt(6/fan_in)) (keeps ReLU activations O(1)
    through the ~15-conv-deep path so parity tests exercise real magnitudes);
    biases U(+-1/sqrt(fan_in-ish)); BN affine and running statistics perturbed
    away from (1,0,0,1) so that folding / eval-mode bugs are visible.
    Returns a new dict of fp32 tensors with the same shapes.
    """
    out = {}
    for name, t in sd.items():
        shape = tuple(t.shape)
        key = f"{tag}:{name}"
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=torch.long)
        elif name.endswith("running_mean"):
            out[name] = det_uniform(key, shape, -0.1, 0.1) if bn_perturb else torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = det_uniform(key, shape, 0.8, 1.25) if bn_perturb else torch.ones(shape)
        elif "norm" in name.split(".")[-2] if "." in name else False:
            if name.endswith("weight"):
                out[name] = det_uniform(key, shape, 0.9, 1.1) if bn_perturb else torch.ones(shape)
            else:
                out[name] = det_uniform(key, shape, -0.1, 0.1) if bn_perturb else torch.zeros(shape)
        elif name.endswith("pos_embedding"):
            out[name] = det_normal(key, shape, 0.02)
        elif name.endswith("down.weight") and len(shape) == 4 and shape[1] == 1:
            out[name] = t.detach().clone().float()          # anti-alias Gaussian buffer: keep analytic value
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            b = gain * math.sqrt(6.0 / fan_in)
            out[name] = det_uniform(key, shape, -b, b)
        else:
            out[name] = det_uniform(key, shape, -0.1, 0.1)
    return out


def fill_tokenpose_state_dict(sd_like: dict, tag: str) -> dict:
    """Deterministic TokenPose_B (MTIA prior) weights: He-uniform convs / Linears (activations stay O(1) through the ~60-conv
    HRNet), BatchNorm / LayerNorm scales in [0.9, 1.1] and shifts in [-0.1, 0.1], perturbed running statistics; the analytic
    sine position code is kept; the Jacobian head (zero-initialised in the reference, tokenpose_base.py:311-312) gets a
    small random weight around the identity so that its gradient path is exercised."""
    sd = fill_state_dict(sd_like, tag=tag)
    for name, t in sd_like.items():
        shape = tuple(t.shape)
        if name.endswith("pos_embedding"):
            sd[name] = t.detach().clone().float()
        elif len(shape) == 1 and name.endswith(".weight"):
            sd[name] = det_uniform(f"{tag}:{name}", shape, 0.9, 1.1)
        elif name.endswith("mlp_head_jacobian.1.weight"):
            sd[name] = sd[name] * 0.3
        elif name.endswith("mlp_head_jacobian.1.bias"):
            sd[name] = torch.tensor([1.0, 0.0, 0.0, 1.0]) + sd[name]
    return sd
