"""MTIA prior (TokenPose_B): the oracle restatement, the product's host logic (through the
C-ABI emulator, CPU) and the product on the MI355X, all against tests/golden/tokenpose.npz -- outputs and gradients
recorded from the unmodified reference by tools/make_goldens.py:g5_tokenpose.

Gradient tolerances: with batch statistics the conv gradients are differences of nearly equal sums; the fp32 reference
itself is only ~1e-2 (worst parameter) from an fp64 run of the same module, so train-mode per-parameter gradient norms are
compared at 5e-2 and eval-mode ones (no cancellation) at 1e-2, sampled gradients element-wise relative to their scale."""
import json
import os

import numpy as np
import pytest
import torch

from mrfa_amd.modules.manifest import manifest_of
from mrfa_amd.modules.transformer import get_pose_net
from mrfa_amd.modules.util import convert_dict_to_attrit_dict
from mrfa_amd.utils.prng import det_uniform
from oracle import tokenpose_oracle as TO
from tests import cases
from tests.emu import emulated_hip

B = 2


def _g(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "tokenpose.npz"))), json.load(open(os.path.join(golden_dir, "tokenpose_param_names.json")))


def _net():
    net = get_pose_net(convert_dict_to_attrit_dict(cases.tokenpose_cfg()), is_train=True)
    sd = cases.tokenpose_weights(net.state_dict(), "tp")
    net.load_state_dict(sd)
    return net, sd


def _loss(o, dev="cpu"):
    gk, gj = det_uniform("g5/gk", (B, 10, 2)).to(dev), det_uniform("g5/gj", (B, 10, 2, 2)).to(dev)
    return (o["kp"] * gk).sum() + (o["jacobian"] * gj).sum()


def _check_outputs(o, g, sfx, tol):
    for key, name in (("kp", "kp"), ("jacobian", "jac")):
        d = np.abs(o[key].detach().float().cpu().numpy() - g[f"{name}_{sfx}"])
        assert np.isfinite(d).all() and d.max() <= tol, f"{key} {sfx}: max {d.max():.3e}"


def _check_grads(grads: dict, g, names, sfx, norm_tol, elem_tol):
    ref_norms = g[f"param_grad_norms_{sfx}"]
    big = ref_norms.max()
    worst = 0.0
    for n, rn in zip(names, ref_norms):
        gn = 0.0 if grads.get(n) is None else float(grads[n].float().norm())
        # parameters whose gradient is a rounding residue (a conv before a train-mode BatchNorm has d/d(scale) = 0) are
        # compared against the scale of the real gradients instead of their own
        worst = max(worst, abs(gn - rn) / max(rn, 1e-3 * big))
    assert worst <= norm_tol, f"{sfx}: worst per-parameter gradient-norm error {worst:.3e}"
    for key in [k for k in g if k.startswith(f"pgrad_{sfx}_")]:
        n = key[len(f"pgrad_{sfx}_"):]
        ref = g[key]
        d = np.abs(grads[n].detach().float().cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert d <= elem_tol, f"{sfx} d/d {n}: {d:.3e}"


def test_state_dict_layout_matches_reference(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    net, _ = _net()
    assert manifest_of(net) == man["TokenPose_B"]
    assert len(man["TokenPose_B"]) == 1074


@pytest.mark.parametrize("train", [False, True])
def test_oracle_vs_reference_goldens(golden_dir, train):
    g, names = _g(golden_dir)
    sfx = "train" if train else "eval"
    net, sd = _net()
    P = {k: v.clone().requires_grad_(v.is_floating_point() and k in names and k != "transformer.pos_embedding") for k, v in sd.items()}
    o = TO.tokenpose_b(cases.images("g5/img", B, 256), P, "", train)
    _check_outputs(o, g, sfx, 2e-5)
    _loss(o).backward()
    _check_grads({n: P[n].grad for n in names}, g, names, sfx, 5e-2 if train else 1e-2, 5e-2 if train else 1e-2)
    if train:
        for key in [k for k in g if k.startswith("buf_")]:
            assert np.abs(P[key[4:]].numpy() - g[key]).max() <= 1e-4, key


def test_sine_position_code_is_the_reference_buffer():
    net, _ = _net()
    assert torch.equal(net.transformer.pos_embedding, TO.sine_position_code(16, 16, 192))


@pytest.mark.parametrize("train", [False, True])
def test_host_logic_through_abi_emulator(golden_dir, train, fresh_mode):
    """the product nn.Module (engine program, tape, islands) with every kernel replaced by its CPU specification"""
    g, names = _g(golden_dir)
    sfx = "train" if train else "eval"
    net, _ = _net()
    net.train(train)
    with emulated_hip():
        o = net(cases.images("g5/img", B, 256))
        _check_outputs(o, g, sfx, 2e-5)
        _loss(o).backward()
    _check_grads({n: p.grad for n, p in net.named_parameters()}, g, names, sfx, 5e-2 if train else 1e-2, 5e-2 if train else 1e-2)
    if train:
        bufs = dict(net.named_buffers())
        for key in [k for k in g if k.startswith("buf_")]:
            assert np.abs(bufs[key[4:]].numpy() - g[key]).max() <= 1e-4, key
        assert int(bufs["pre_feature.stage3.3.branches.0.3.bn2.num_batches_tracked"]) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
@pytest.mark.parametrize("train", [False, True])
def test_gpu_vs_reference_goldens(golden_dir, train, mode, fresh_mode):
    from mrfa_amd import hip
    g, names = _g(golden_dir)
    sfx = "train" if train else "eval"
    net, _ = _net()
    net.to("cuda:0").train(train)
    prev = hip.mfma_mode()
    hip.set_mfma_mode(mode)
    try:
        o = net(cases.images("g5/img", B, 256).to("cuda:0"))
        _check_outputs(o, g, sfx, 1e-4)
        _loss(o, "cuda:0").backward()
        torch.cuda.synchronize()
    finally:
        hip.set_mfma_mode(prev)
    _check_grads({n: p.grad for n, p in net.named_parameters()}, g, names, sfx, 5e-2 if train else 1e-2, 5e-2 if train else 1e-2)
    if train:
        bufs = dict(net.named_buffers())
        for key in [k for k in g if k.startswith("buf_")]:
            assert np.abs(bufs[key[4:]].cpu().numpy() - g[key]).max() <= 1e-4, key


@pytest.mark.gpu
def test_gpu_vs_oracle_at_bench_batch():
    """B = 8 fresh inputs, train mode: product on the GPU against the CPU oracle (outputs 1e-4, well inside north_star's 1e-3)"""
    net, sd = _net()
    x = cases.images("tp/fresh", 8, 256)
    P = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref = TO.tokenpose_b(x, P, "", True)
        net.to("cuda:0").train(True)
        o = net(x.to("cuda:0"))
    for k in ("kp", "jacobian"):
        d = (o[k].cpu() - ref[k]).abs()
        assert d.max() <= 1e-4, f"{k}: {d.max():.3e}"
    bufs = dict(net.named_buffers())
    for n in ("pre_feature.bn2.running_var", "pre_feature.stage3.3.fuse_layers.0.1.1.running_mean"):
        assert (bufs[n].cpu() - P[n]).abs().max() <= 1e-4, n
