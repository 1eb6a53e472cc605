"""BGMotionPredictor (resnet18), DenseMotionNetwork(bg_param) and the background loss (reference
modules/bg_motion_predictor.py:5-24, dense_motion.py:67-73, model.py:248-253) against tests/golden/background.npz, recorded from the
reference's own classes (tools/make_goldens.py:g8_background)."""
import json
import os

import numpy as np
import pytest
import torch

from mrfa_amd.modules import BGMotionPredictor, DenseMotionNetwork
from mrfa_amd.modules.manifest import manifest_of
from oracle import losses_oracle as LO
from oracle import mrfa_oracle as O
from tests import cases
from tests.emu import emulated_hip

B = 2


def _g(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "background.npz")))


def _bg(dev="cpu"):
    m = BGMotionPredictor()
    sd = cases.bg_weights(m.state_dict())
    m.load_state_dict(sd)
    return m.to(dev), sd


def test_state_dict_layout_matches_reference(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    assert manifest_of(BGMotionPredictor()) == man["BGMotionPredictor"] and len(man["BGMotionPredictor"]) == 122
    m = BGMotionPredictor()
    assert float(m.bg_encoder.fc.weight.abs().max()) == 0 and m.bg_encoder.fc.bias.tolist() == [1, 0, 0, 0, 1, 0]


@pytest.mark.parametrize("train", [False, True])
def test_oracle_vs_reference_goldens(golden_dir, train):
    g = _g(golden_dir)
    sfx = "train" if train else "eval"
    _, sd = _bg()
    src, drv = cases.images("g8/src", B, 256), cases.images("g8/drv", B, 256)
    P = {k: v.clone() for k, v in sd.items()}
    fwd = LO.bg_motion_predictor(src, drv, P, "", train)
    rev = LO.bg_motion_predictor(drv, src, P, "", train)
    assert np.abs(fwd.numpy() - g[f"bg_{sfx}"]).max() <= 1e-5 and np.abs(rev.numpy() - g[f"bg_rev_{sfx}"]).max() <= 1e-5
    assert abs(LO.bg_loss(fwd, rev).item() - g[f"bg_loss_{sfx}"][0]) <= 1e-4 * g[f"bg_loss_{sfx}"][0]
    kd, ks = cases.keypoints("g8/kd", B), cases.keypoints("g8/ks", B)
    assert np.abs(O.sparse_motions(kd, ks, 8, 8, torch.from_numpy(g["bg_eval"])).numpy() - g["sparse_motions_bg"]).max() <= 1e-5


def _check(g, golden_dir, dev, train):
    sfx = "train" if train else "eval"
    m, _ = _bg(dev)
    m.train(train)
    src, drv = cases.images("g8/src", B, 256).to(dev), cases.images("g8/drv", B, 256).to(dev)
    fwd, rev = m(src, drv), m(drv, src)
    assert np.abs(fwd.detach().cpu().numpy() - g[f"bg_{sfx}"]).max() <= 2e-4, np.abs(fwd.detach().cpu().numpy() - g[f"bg_{sfx}"]).max()
    assert np.abs(rev.detach().cpu().numpy() - g[f"bg_rev_{sfx}"]).max() <= 2e-4
    value = torch.matmul(fwd, rev)
    loss = 10 * torch.abs(torch.eye(3, device=dev).view(1, 3, 3) - value).mean()
    assert abs(loss.item() - g[f"bg_loss_{sfx}"][0]) <= 1e-3 * g[f"bg_loss_{sfx}"][0]
    loss.backward()
    names = json.load(open(os.path.join(golden_dir, "bg_param_names.json")))
    params = dict(m.named_parameters())
    ref = g[f"param_grad_norms_{sfx}"]
    errs = [abs(float(params[n].grad.norm()) - rn) / max(rn, 1e-3 * ref.max()) for n, rn in zip(names, ref)]
    assert np.median(errs) <= 1e-2 and max(errs) <= (0.2 if train else 5e-2), (np.median(errs), max(errs))
    if train:
        rv = dict(m.named_buffers())["bg_encoder.bn1.running_var"].cpu().numpy()
        assert np.abs(rv - g["buf_bn1_running_var"]).max() <= 1e-4 * np.abs(g["buf_bn1_running_var"]).max()
        assert int(dict(m.named_buffers())["bg_encoder.layer4.1.bn2.num_batches_tracked"]) == 2          # forward + reverse call


def _check_dense_motion(g, dev):
    dm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    dm.load_state_dict(cases.weights_for(dm.state_dict(), "dm"))
    dm.to(dev).eval()
    kd = {k: v.to(dev) for k, v in cases.keypoints("g8/kd", B).items()}
    ks = {k: v.to(dev) for k, v in cases.keypoints("g8/ks", B).items()}
    bgp = torch.from_numpy(g["bg_eval"]).to(dev).requires_grad_(True)
    r = dm(cases.images("g8/src", B, 256).to(dev), kd, ks, bg_param=bgp)
    assert np.abs(r["deformation"].detach().cpu().numpy() - g["dm_bg_deformation"]).max() <= 1e-4
    assert np.abs(r["occlusion"].detach().cpu().numpy() - g["dm_bg_occlusion"]).max() <= 1e-3
    r["deformation"].sum().backward()
    assert bgp.grad is not None and torch.isfinite(bgp.grad).all() and float(bgp.grad.abs().max()) > 0


@pytest.mark.parametrize("train", [False, True])
def test_product_through_abi_emulator(golden_dir, train, fresh_mode):
    with emulated_hip():
        _check(_g(golden_dir), golden_dir, "cpu", train)


def test_dense_motion_with_background_through_abi_emulator(golden_dir):
    with emulated_hip():
        _check_dense_motion(_g(golden_dir), "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True])
def test_gpu(golden_dir, train, fresh_mode):
    _check(_g(golden_dir), golden_dir, "cuda:0", train)
    _check_dense_motion(_g(golden_dir), "cuda:0")
