"""Callers of the path (last rows): normalize_kp / psnr against values recorded from the reference's own
functions (tests/golden/callers.npz, tools/make_goldens.py:g6_callers), the reconstruction and animation loops on synthetic
video."""
import os

import numpy as np
import pytest
import torch

from mrfa_amd.infer import _hull_area, make_animation, normalize_kp, psnr, reconstruction
from tests import cases


def test_normalize_kp_and_psnr_match_reference(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "callers.npz")))
    ks, kd, k0 = cases.keypoints("g6/ks", 2), cases.keypoints("g6/kd", 2), cases.keypoints("g6/k0", 2)
    for adapt in (False, True):
        for rel, relj in ((False, False), (True, False), (True, True)):
            r = normalize_kp(ks, kd, k0, adapt_movement_scale=adapt, use_relative_movement=rel, use_relative_jacobian=relj)
            tag = f"a{int(adapt)}_r{int(rel)}_j{int(relj)}"
            assert np.abs(r["kp"].numpy() - g[f"norm_kp_{tag}"]).max() <= 1e-5, tag
            assert np.abs(r["jacobian"].numpy() - g[f"norm_jac_{tag}"]).max() <= 1e-5, tag
    assert all(torch.equal(kd[k], cases.keypoints("g6/kd", 2)[k]) for k in kd), "inputs must not be modified"
    a, b = cases.images("g6/a", 2, 32), cases.images("g6/b", 2, 32)
    assert abs(float(psnr(a, b)) - float(g["psnr"][0])) <= 1e-4
    assert psnr(a, a) == float("inf")


def test_hull_area_equals_qhull():
    from scipy.spatial import ConvexHull
    for seed in range(5):
        pts = cases.keypoints(f"hull/{seed}", 1)["kp"][0]
        assert abs(float(_hull_area(pts)) - ConvexHull(pts.numpy()).volume) <= 1e-5
    sq = torch.tensor([[0.0, 0.0], [2.0, 0.0], [2.0, 1.0], [0.0, 1.0], [1.0, 0.5], [0.3, 0.2]])
    assert abs(float(_hull_area(sq)) - 2.0) <= 1e-6


@pytest.mark.gpu
def test_reconstruction_and_animation_loops_on_synthetic_video():
    """reconstruction.py:52-70 / demo.py:47-73 on a 3-frame synthetic clip: frame 0 reconstructs itself through the cached
    source half exactly like the plain forward, metrics are finite, and relative animation of the source by its own clip's
    first frame is the self-reconstruction (normalize_kp with driving == driving_initial returns the source keypoints)."""
    import bench
    from mrfa_amd.train import VOX1, HotPath
    dev = "cuda:0"
    model = HotPath(VOX1, prior="mtia")
    bench.init_weights(model)
    model.to(dev).eval()
    clip = torch.stack([cases.images(f"clip/{t}", 2, 256) for t in range(3)], dim=2).to(dev)        # (B,3,T,H,W)
    r = reconstruction(model, clip)
    assert r["prediction"].shape == clip.shape and len(r["l1"]) == 3 and all(np.isfinite(r["l1"])) and all(np.isfinite(r["psnr"]))
    with torch.no_grad():
        full = model(clip[:, :, 0].contiguous(), clip[:, :, 1].contiguous())
    assert (r["prediction"][:, :, 1] - full).abs().max().item() <= 2e-4
    anim = make_animation(model, clip[:, :, 0].contiguous(), clip, relative=True)
    with torch.no_grad():
        self_rec = model(clip[:, :, 0].contiguous(), clip[:, :, 0].contiguous())
    assert anim.shape == clip.shape and (anim[:, :, 0] - self_rec).abs().max().item() <= 2e-4
    assert (anim[:, :, 1] - anim[:, :, 0]).abs().max().item() > 1e-3


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's own callers as the oracle (tests/golden/dropin_<prior>.npz, written by tools/check_dropin.py from the reference's
# unmodified MRFA.forward(is_train=False), model.py:183-216, and demo.make_animation, demo.py:47-73 -> animate_ddp.normalize_kp with
# scipy's ConvexHull).  Fresh inputs ("dropin/*"), weights with a x50 sharpened mask softmax (pixel-scale motion, out-of-image samples).
def _dropin_model(prior, dev):
    import copy
    from mrfa_amd.modules import MRFA
    from mrfa_amd.modules.util import convert_dict_to_attrit_dict
    from mrfa_amd.train import VOX1
    cfg = copy.deepcopy(VOX1)
    cfg["train_params"] = dict(prior_model=prior, num_epochs=100, bg_start=1000, scales=[1, 0.5, 0.25, 0.125],
                               loss_weights=dict(perceptual=[10, 10, 10, 10, 10], equivariance=10, equivariance_jacobian=10),
                               transform_params=dict(sigma_affine=0.05, sigma_tps=0.005, points_tps=5))
    m = MRFA(convert_dict_to_attrit_dict(cfg))
    if prior == "mtia":
        m.encoder.load_state_dict(cases.tokenpose_weights(m.encoder.state_dict(), "dropin/enc"))
    else:
        m.encoder.load_state_dict(cases.weights_for(m.encoder.state_dict(), "kp"))
    m.dense_motion.load_state_dict(cases.weights_for(m.dense_motion.state_dict(), "dm"))
    m.decoder.load_state_dict(cases.weights_for(m.decoder.state_dict(), "rf"))
    with torch.no_grad():
        m.dense_motion.mask.weight.mul_(50.0)
    return m.to(dev).eval()


def _check_against_reference_callers(golden_dir, prior, dev, graph_modes):
    from mrfa_amd.infer import Animator
    g = np.load(os.path.join(golden_dir, f"dropin_{prior}.npz"))
    m = _dropin_model(prior, dev)
    src = cases.images("dropin/src", 1, 256).to(dev)
    drv = [cases.images(f"dropin/drv{t}", 1, 256).to(dev) for t in range(3)]
    with torch.no_grad():
        gen, warp_img, losses, kp_s, kp_d = m({"source": src, "driving": drv[1]}, is_train=False)
        jac = m.encoder(src)["jacobian"]

    def err(t, ref):
        d = np.abs(t.detach().cpu().numpy() - ref)
        return d.max(), d.mean()
    assert losses == {}
    assert err(kp_s, g["kp_s"])[0] <= 1e-4 and err(kp_d, g["kp_d"])[0] <= 1e-4 and err(jac, g["jac_s"])[0] <= 1e-4
    mx, mean = err(gen[:, :, ::2, ::2], g["gen"])
    assert mean <= 1e-4 and mx <= 5e-3, (mx, mean)              # north_star: L1 <= 1e-3; max: tools/check_dropin.py's note on border pixels
    mx, mean = err(warp_img[:, :, ::4, ::4], g["warp_img_s4"])
    assert mean <= 1e-4 and mx <= 5e-3, (mx, mean)
    clip = torch.stack(drv, dim=2)                               # (1,3,T,H,W)
    ref_anim = g["animation"]                                    # (T,H/2,W/2,3) of the reference's make_animation(relative, adapt_movement_scale)
    for graph in graph_modes:
        anim = make_animation(m, src, clip, relative=True, adapt_movement_scale=True, graph=graph)
        assert anim.shape == clip.shape
        mine = anim[0].permute(1, 2, 3, 0)[:, ::2, ::2, :]
        mx, mean = err(mine, ref_anim)
        assert mean <= 1e-4 and mx <= 5e-3, (graph, mx, mean)
    # the streaming form (source cached once, one call per driving frame; relative motion = kp_driving itself here)
    a = Animator(m, graph=(True in graph_modes))
    a.set_source(src)
    with torch.no_grad():
        full = m({"source": src, "driving": drv[2]}, is_train=False)[0]
        d = (a(drv[2]) - full).abs()
        assert d.mean().item() <= 2e-5 and d.max().item() <= 5e-3, (d.max().item(), d.mean().item())


def test_reference_callers_through_abi_emulator(golden_dir):
    from tests.emu import emulated_hip
    with emulated_hip():
        _check_against_reference_callers(golden_dir, "fomm", "cpu", (False,))


@pytest.mark.gpu
@pytest.mark.parametrize("prior", ["fomm", "mtia"])
def test_reference_callers_gpu(golden_dir, prior):
    """f3: MRFA.forward(is_train=False), make_animation (eager and hipGraph) and the on-device normalize_kp against the reference's own
    callers -- not against this repo's full forward"""
    _check_against_reference_callers(golden_dir, prior, "cuda:0", (False, True))
