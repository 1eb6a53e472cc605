"""The CPU oracle (oracle/mrfa_oracle.py) against the golden vectors recorded from the reference itself
(tools/make_goldens.py).  This is what pins the oracle: every later HIP-vs-oracle parity claim rests on it."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mrfa_amd.utils.prng import det_uniform
from oracle import mrfa_oracle as O
from tests import cases

TOL = 1e-5          # abs, fp32 oracle vs fp32 reference (measured deltas are <=3e-5 at 256^2)


def _g(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _close(a, ref, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == ref.shape, (a.shape, ref.shape)
    d = np.abs(a - ref).max()
    assert d <= tol, f"max abs diff {d:.3e} > {tol}"


def test_unit_functions(golden_dir):
    g = _g(golden_dir, "unit.npz")
    like = torch.zeros(1)
    _close(O.coordinate_grid(5, 7, like), g["grid_5x7"], 0)
    _close(O.pixel_grid(2, 3, 4, like), g["coords_2x3x4"], 0)
    kp = det_uniform("u/kp", (2, 3, 2), -0.9, 0.9)
    _close(O.gaussian_heatmap(kp, 8, 8, 0.01), g["gauss_001"], 1e-7)
    _close(O.gaussian_heatmap(kp, 6, 9, 0.1), g["gauss_01"], 1e-7)
    img = cases.images("u/aa", 1, 32)
    for s in (0.25, 0.5, 0.125):
        _close(O.antialias_down(img, s), g[f"aa_{s}"], 1e-7)
    small = det_uniform("u/bs_img", (2, 4, 6, 7), -1, 1)
    _close(O.sample_px(small, det_uniform("u/bs_xy", (2, 5, 6, 2), -1.5, 7.5)), g["bilinear_sampler"], 1e-7)
    _close(O.sample_norm(small, det_uniform("u/gs_xy", (2, 5, 6, 2), -1.3, 1.3)), g["grid_sample_default"], 1e-7)
    xr = det_uniform("u/rs", (2, 3, 8, 8), -1, 1)
    _close(O.resize_ac(xr, 3), g["resize_ac_8to3"], 1e-6)
    _close(O.resize_ac(xr, 16), g["resize_ac_8to16"], 1e-6)
    _close(O.resize_ac(xr, 13), g["resize_ac_8to13"], 1e-6)
    maps = det_uniform("u/corr_maps", (18, 1, 8, 8), -1, 1)
    _close(O.corr_lookup(maps, det_uniform("u/corr_xy", (2, 2, 3, 3), -2.0, 9.0)), g["corrblock"], 1e-6)
    kd, ks = cases.keypoints("u/kd", 2), cases.keypoints("u/ks", 2)
    _close(O.sparse_motions(kd, ks, 8, 8), g["sparse_motions"], 1e-6)


BLOCKS = {
    "down": ((16, 8, 3), (2, 8, 8, 8), lambda x, P, t: O.down_block(x, P, "b", t)),
    "up": ((8, 16, 3), (2, 16, 4, 4), lambda x, P, t: O.up_block(x, P, "b", t)),
    "same7": ((8, 3, 7), (2, 3, 8, 8), lambda x, P, t: O.same_block(x, P, "b", t, 3)),
}


def _block_sd(name):
    """state_dict layouts of the reference blocks (modules/util.py:111-214), shapes only."""
    def cv(co, ci, k):
        return {"weight": torch.zeros(co, ci, k, k), "bias": torch.zeros(co)}

    def bn(c):
        return {"weight": torch.zeros(c), "bias": torch.zeros(c), "running_mean": torch.zeros(c),
                "running_var": torch.zeros(c), "num_batches_tracked": torch.zeros((), dtype=torch.long)}
    if name in BLOCKS:
        co, ci, k = BLOCKS[name][0]
        parts = {"conv": cv(co, ci, k), "norm": bn(co)}
    elif name == "res":
        parts = {"conv1": cv(8, 8, 3), "conv2": cv(8, 8, 3), "norm1": bn(8), "norm2": bn(8)}
    elif name == "chan":
        parts = {"conv1": cv(8, 16, 3), "norm1": bn(16)}
    return {f"{p}.{k}": v for p, d in parts.items() for k, v in d.items()}


@pytest.mark.parametrize("name", ["down", "up", "same7", "res", "chan"])
@pytest.mark.parametrize("train", [False, True])
def test_blocks(golden_dir, name, train):
    g = _g(golden_dir, "unit.npz")
    sd = cases.weights_for(_block_sd(name), f"u/{name}")
    P = {"b." + k: v for k, v in sd.items()}
    shapes = {"down": (2, 8, 8, 8), "up": (2, 16, 4, 4), "same7": (2, 3, 8, 8), "res": (2, 8, 8, 8), "chan": (2, 16, 8, 8)}
    fns = {"down": lambda x: O.down_block(x, P, "b", train), "up": lambda x: O.up_block(x, P, "b", train),
           "same7": lambda x: O.same_block(x, P, "b", train, 3), "res": lambda x: O.res_block(x, P, "b", train),
           "chan": lambda x: O.channel_block(x, P, "b", train)}
    x = det_uniform(f"u/{name}/x", shapes[name], -1, 1)
    _close(fns[name](x), g[f"block_{name}_{'train' if train else 'eval'}"], 2e-6)


def _module_sd(golden_dir, which):
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))[which]
    return {n: torch.zeros(s, dtype=getattr(torch, d)) for n, s, d in man}


@pytest.mark.parametrize("train", [False, True])
def test_prior_modules(golden_dir, train):
    g = _g(golden_dir, "prior.npz")
    sfx = "train" if train else "eval"
    x = cases.images("g3/src", 2, 256)
    sd = _module_sd(golden_dir, "KPDetector")
    sd["down.weight"] = O.antialias_kernel(0.25, torch.zeros(1)).view(1, 1, 13, 13).repeat(3, 1, 1, 1)
    P = cases.weights_for(sd, "kp")
    with torch.no_grad():
        o = O.kp_detector(x, P, "", train)
    _close(o["kp"], g[f"kp_{sfx}"], 2e-5)
    _close(o["jacobian"], g[f"jac_{sfx}"], 2e-5)
    sd = _module_sd(golden_dir, "DenseMotionNetwork")
    sd["down.weight"] = O.antialias_kernel(0.25, torch.zeros(1)).view(1, 1, 13, 13).repeat(3, 1, 1, 1)
    P = cases.weights_for(sd, "dm")
    kd, ks = cases.keypoints("g3/kd", 2), cases.keypoints("g3/ks", 2)
    with torch.no_grad():
        o = O.dense_motion(x, kd, ks, P, "", train)
    _close(o["deformation"], g[f"dm_deformation_{sfx}"], 2e-5)
    _close(o["occlusion"], g[f"dm_occlusion_{sfx}"], 2e-5)
    _close(o["mask"][:, :, ::4, ::4], g[f"dm_mask_{sfx}_s4"], 2e-5)
    _close(o["sparse_deformed"][:, :, :, ::4, ::4], g[f"dm_sparse_deformed_{sfx}_s4"], 2e-5)


def raft_inputs(size, b, tagp):
    img_full = cases.images(f"{tagp}/src", b, size)
    img = O.antialias_down(img_full, 0.25)
    kp_s = cases.keypoints(f"{tagp}/ks", b)["kp"]
    kp_d = cases.keypoints(f"{tagp}/kd", b)["kp"]
    dmo = cases.synthetic_dense_motion(f"{tagp}/dm", b, size // 4)
    return kp_s, kp_d, dmo, img, img_full


def raft_params(golden_dir, size, prior_only=False):
    """RaftFlow state_dict shapes for a given size: only pos_embedding and the hourglass depth change."""
    from mrfa_amd.modules.manifest import raft_flow_manifest
    sd = {n: torch.zeros(s) if d != "int64" else torch.zeros(s, dtype=torch.long)
          for n, s, d in raft_flow_manifest(cases.raft_cfg(size, prior_only))}
    return cases.weights_for(sd, "rf")


@pytest.mark.parametrize("size,b,stride", [(64, 2, 1), (128, 2, 2), (256, 1, 4)])
@pytest.mark.parametrize("prior_only", [False, True])
def test_raft_flow(golden_dir, size, b, stride, prior_only):
    g = _g(golden_dir, f"raft_{size}.npz")
    P0 = raft_params(golden_dir, size, prior_only)
    for train in ((False, True) if size <= 128 else (False,)):
        sfx = ("prior_" if prior_only else "") + ("train" if train else "eval")
        P = {k: v.clone() for k, v in P0.items()}
        with torch.no_grad():
            o, w, s = O.raft_flow(*raft_inputs(size, b, f"g3/raft{size}"), P, "", size=size,
                                  prior_only=prior_only, train=train)
        _close(o[:, :, ::stride, ::stride], g[f"out_{sfx}"], 1e-4)
        _close(w[:, :, ::stride, ::stride], g[f"warp_{sfx}"], 1e-4)
        _close(s[:, :, ::stride * 2, ::stride * 2], g[f"strip_{sfx}"], 1e-4)
        _close(o.mean(dim=(2, 3)), g[f"out_mean_{sfx}"], 1e-5)


def test_raft_gradients(golden_dir):
    g = _g(golden_dir, "grads_64.npz")
    names = json.load(open(os.path.join(golden_dir, "grads_64_param_names.json")))
    size, b = 64, 2
    P = raft_params(golden_dir, size)
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in P.items()}
    kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, "g4/raft")
    leaves = [t.clone().requires_grad_(True) for t in (kp_s, kp_d, dmo["deformation"], dmo["occlusion"])]
    driving = cases.images("g4/drv", b, size)
    o, _, _ = O.raft_flow(leaves[0], leaves[1], {"deformation": leaves[2], "occlusion": leaves[3]}, img, img_full,
                          P, "", size=size, train=True)
    loss = (o - driving).abs().mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"][0])) < 1e-6
    for n, t in zip(("kp_s", "kp_d", "deformation", "occlusion"), leaves):
        _close(t.grad, g[f"grad_{n}"], 1e-4)
    norms = np.array([0.0 if P[n].grad is None else P[n].grad.norm().item() for n in names], np.float32)
    ref = g["param_grad_norms"]
    assert np.abs(norms - ref).max() <= 1e-4 + 1e-3 * np.abs(ref).max()
    for key in g:
        if key.startswith("pgrad_"):
            _close(P[key[6:]].grad, g[key], 1e-5 + 1e-3 * np.abs(g[key]).max())
