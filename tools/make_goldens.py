"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference) on CPU.

Runs only in the build container (the reference does not exist on the GPU box).  Inputs/weights are the
deterministic cases of tests/cases.py; only the reference outputs are stored.  Re-run:
    python tools/make_goldens.py            # writes tests/golden/, prints oracle-vs-reference deltas
Reference pins torch 1.10.1; this container runs torch 2.10 (CPU).  The ops used have unchanged semantics.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_import  # noqa: E402
from tests import cases  # noqa: E402
from mrfa_amd.utils.prng import det_uniform, det_normal  # noqa: E402
from oracle import mrfa_oracle as O  # noqa: E402

ref_import.import_reference()
from modules import util as RU  # noqa: E402
from modules.kp_detector import KPDetector  # noqa: E402
from modules.dense_motion import DenseMotionNetwork  # noqa: E402
from modules.raft import RaftFlow, CorrBlock  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.manual_seed(0)


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


def delta(name, a, b):
    d = (a.detach() - b.detach()).abs()
    print(f"   oracle-vs-ref {name:28s} max {d.max().item():.3e}  mean {d.mean().item():.3e}  |ref| {b.abs().mean().item():.3e}")
    return d.max().item()


def load(mod, tag, **kw):
    sd = cases.weights_for(mod.state_dict(), tag, **kw)
    mod.load_state_dict(sd, strict=True)
    return sd


def manifest(mod):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in mod.state_dict().items()]


# ----------------------------------------------------------------------------------------------- G1 manifests
def g1():
    kp = KPDetector(**cases.KP_DETECTOR_CFG)
    dm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    rf = RaftFlow(**cases.raft_cfg(256))
    rfp = RaftFlow(**cases.raft_cfg(256, prior_only=True))
    man = {"KPDetector": manifest(kp), "DenseMotionNetwork": manifest(dm), "RaftFlow": manifest(rf),
           "RaftFlow_prior_only": manifest(rfp)}
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0)
    print("G1 manifests:", {k: len(v) for k, v in man.items()})


# ----------------------------------------------------------------------------------------------- G2 units
def g2():
    out = {}
    like = torch.zeros(1)
    out["grid_5x7"] = npy(RU.make_coordinate_grid((5, 7), like.type()))
    out["coords_2x3x4"] = npy(RU.coords_grid(2, 3, 4, "cpu"))
    kp = det_uniform("u/kp", (2, 3, 2), -0.9, 0.9)
    out["gauss_001"] = npy(RU.kp2gaussian(kp, (8, 8), 0.01))
    out["gauss_01"] = npy(RU.kp2gaussian(kp, (6, 9), 0.1))
    img = cases.images("u/aa", 1, 32)
    for s in (0.25, 0.5, 0.125):
        out[f"aa_{s}"] = npy(RU.AntiAliasInterpolation2d(3, s)(img))
        delta(f"antialias {s}", O.antialias_down(img, s), torch.from_numpy(out[f"aa_{s}"]))
    small = det_uniform("u/bs_img", (2, 4, 6, 7), -1, 1)
    coords = det_uniform("u/bs_xy", (2, 5, 6, 2), -1.5, 7.5)
    out["bilinear_sampler"] = npy(RU.bilinear_sampler(small, coords))
    delta("bilinear_sampler", O.sample_px(small, coords), torch.from_numpy(out["bilinear_sampler"]))
    gridn = det_uniform("u/gs_xy", (2, 5, 6, 2), -1.3, 1.3)
    out["grid_sample_default"] = npy(F.grid_sample(small, gridn))
    delta("grid_sample default", O.sample_norm(small, gridn), torch.from_numpy(out["grid_sample_default"]))
    xr = det_uniform("u/rs", (2, 3, 8, 8), -1, 1)
    out["resize_ac_8to3"] = npy(F.interpolate(xr, scale_factor=1.0 / 8.0 * 3, mode="bilinear", align_corners=True))
    out["resize_ac_8to16"] = npy(F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True))
    out["resize_ac_8to13"] = npy(F.interpolate(xr, size=13, mode="bilinear", align_corners=True))
    # CorrBlock: 2*3*3 query pixels, 8x8 source maps
    maps = det_uniform("u/corr_maps", (18, 1, 8, 8), -1, 1)
    cxy = det_uniform("u/corr_xy", (2, 2, 3, 3), -2.0, 9.0)
    out["corrblock"] = npy(CorrBlock(maps)(cxy))
    delta("corrblock", O.corr_lookup(maps, cxy), torch.from_numpy(out["corrblock"]))
    # blocks
    blocks = {
        "down": (RU.DownBlock2d(8, 16), (2, 8, 8, 8), lambda x, P, t: O.down_block(x, P, "b", t)),
        "up": (RU.UpBlock2d(16, 8), (2, 16, 4, 4), lambda x, P, t: O.up_block(x, P, "b", t)),
        "same7": (RU.SameBlock2d(3, 8, kernel_size=(7, 7), padding=(3, 3)), (2, 3, 8, 8), lambda x, P, t: O.same_block(x, P, "b", t, 3)),
        "res": (RU.ResBlock2d(8, kernel_size=(3, 3), padding=(1, 1)), (2, 8, 8, 8), lambda x, P, t: O.res_block(x, P, "b", t)),
        "chan": (RU.ChannelBlock2d(16, kernel_size=(3, 3), padding=(1, 1)), (2, 16, 8, 8), lambda x, P, t: O.channel_block(x, P, "b", t)),
    }
    for name, (mod, shp, ofn) in blocks.items():
        sd = load(mod, f"u/{name}")
        x = det_uniform(f"u/{name}/x", shp, -1, 1)
        for train in (False, True):
            mod.train(train)
            y = mod(x.clone())
            key = f"block_{name}_{'train' if train else 'eval'}"
            out[key] = npy(y)
            P = {("b." + k): v.clone() for k, v in sd.items()}
            delta(key, ofn(x.clone(), P, train), y)
        mod.load_state_dict(sd)
    hg = RU.Hourglass(block_expansion=8, in_features=5, num_blocks=3, max_features=32)
    sd = load(hg, "u/hg")
    x = det_uniform("u/hg/x", (2, 5, 16, 16), -1, 1)
    for train in (False, True):
        hg.load_state_dict(sd)
        hg.train(train)
        y = hg(x)
        out[f"hourglass_{'train' if train else 'eval'}"] = npy(y)
        P = {("hg." + k): v.clone() for k, v in sd.items()}
        delta(f"hourglass train={train}", O.hourglass(x, P, "hg", train), y)
    # sparse motions (dense_motion.py:48-76) through a tiny DenseMotionNetwork instance
    dmn = DenseMotionNetwork(block_expansion=8, num_blocks=2, max_features=16, num_kp=10, num_channels=3)
    kd, ks = cases.keypoints("u/kd", 2), cases.keypoints("u/ks", 2)
    sm = dmn.create_sparse_motions(torch.zeros(2, 3, 8, 8), kd, ks)
    out["sparse_motions"] = npy(sm)
    delta("sparse_motions", O.sparse_motions(kd, ks, 8, 8), sm)
    np.savez_compressed(os.path.join(GOLD, "unit.npz"), **out)
    print("G2 unit goldens:", len(out))


# ----------------------------------------------------------------------------------------------- G3 modules
def g3_prior():
    out = {}
    kpm = KPDetector(**cases.KP_DETECTOR_CFG)
    sd = load(kpm, "kp")
    x = cases.images("g3/src", 2, 256)
    for train in (False, True):
        kpm.load_state_dict(sd)
        kpm.train(train)
        with torch.no_grad():
            r = kpm(x)
        sfx = "train" if train else "eval"
        out[f"kp_{sfx}"] = npy(r["kp"])
        out[f"jac_{sfx}"] = npy(r["jacobian"])
        P = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            o = O.kp_detector(x, P, "", train)
        delta(f"KPDetector kp {sfx}", o["kp"], r["kp"])
        delta(f"KPDetector jac {sfx}", o["jacobian"], r["jacobian"])
    dmm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    sd = load(dmm, "dm")
    kd, ks = cases.keypoints("g3/kd", 2), cases.keypoints("g3/ks", 2)
    for train in (False, True):
        dmm.load_state_dict(sd)
        dmm.train(train)
        with torch.no_grad():
            r = dmm(x, kd, ks)
        sfx = "train" if train else "eval"
        out[f"dm_deformation_{sfx}"] = npy(r["deformation"])
        out[f"dm_occlusion_{sfx}"] = npy(r["occlusion"])
        out[f"dm_mask_{sfx}_s4"] = npy(r["mask"][:, :, ::4, ::4])
        out[f"dm_logit_mask_{sfx}_s4"] = npy(r["logit_mask"][:, :, ::4, ::4])
        out[f"dm_sparse_deformed_{sfx}_s4"] = npy(r["sparse_deformed"][:, :, :, ::4, ::4])
        P = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            o = O.dense_motion(x, kd, ks, P, "", train)
        for key in ("deformation", "occlusion", "mask", "sparse_deformed"):
            delta(f"DenseMotion {key} {sfx}", o[key], r[key])
        disp = (r["deformation"] - O.coordinate_grid(64, 64, x)[None]) * 31.5
        print(f"   prior displacement px: mean {disp.abs().mean():.2f} max {disp.abs().max():.2f}; mask max-prob {r['mask'].max(1).values.mean():.2f}")
    np.savez_compressed(os.path.join(GOLD, "prior.npz"), **out)
    print("G3 prior goldens:", len(out))


def run_raft(rf, sd, size, b, train, tagp, prior_only=False):
    rf.load_state_dict(sd)
    rf.train(train)
    h = size // 4
    img_full = cases.images(f"{tagp}/src", b, size)
    img = O.antialias_down(img_full, 0.25)
    kp_s = cases.keypoints(f"{tagp}/ks", b)["kp"]
    kp_d = cases.keypoints(f"{tagp}/kd", b)["kp"]
    dmo = cases.synthetic_dense_motion(f"{tagp}/dm", b, h)
    return (kp_s, kp_d, dmo, img, img_full)


def g3_raft():
    for size, b, stride in ((64, 2, 1), (128, 2, 2), (256, 1, 4)):
        out = {}
        for prior_only in (False, True):
            rf = RaftFlow(**cases.raft_cfg(size, prior_only))
            sd = load(rf, "rf")
            for train in ((False, True) if size <= 128 else (False,)):
                args = run_raft(rf, sd, size, b, train, f"g3/raft{size}", prior_only)
                with torch.no_grad():
                    o_ref, w_ref, s_ref = rf(*args)
                sfx = ("prior_" if prior_only else "") + ("train" if train else "eval")
                out[f"out_{sfx}"] = npy(o_ref[:, :, ::stride, ::stride])
                out[f"warp_{sfx}"] = npy(w_ref[:, :, ::stride, ::stride])
                out[f"strip_{sfx}"] = npy(s_ref[:, :, ::stride * 2, ::stride * 2])
                out[f"out_mean_{sfx}"] = npy(o_ref.mean(dim=(2, 3)))
                out[f"out_std_{sfx}"] = npy(o_ref.std(dim=(2, 3)))
                P = {k: v.clone() for k, v in sd.items()}
                trace = {}
                with torch.no_grad():
                    o, w, s = O.raft_flow(*args, P, "", size=size, prior_only=prior_only, train=train, trace=trace)
                delta(f"RaftFlow{size} out {sfx}", o, o_ref)
                delta(f"RaftFlow{size} warp {sfx}", w, w_ref)
                delta(f"RaftFlow{size} strip {sfx}", s, s_ref)
                if not prior_only and not train:
                    for i in range(6):
                        df = trace[f"d_flow_{i}"]
                        print(f"      level {i}: |flow_in| mean {trace[f'flow_in_{i}'].abs().mean():.3f} max {trace[f'flow_in_{i}'].abs().max():.2f}"
                              f"  |d_flow| mean {df[:, :2].abs().mean():.3f} max {df[:, :2].abs().max():.2f}")
                        if size <= 128:
                            out[f"trace_flow_in_{i}"] = npy(trace[f"flow_in_{i}"][:, :, ::stride, ::stride])
                            out[f"trace_d_flow_{i}"] = npy(df[:, :, ::stride, ::stride])
        np.savez_compressed(os.path.join(GOLD, f"raft_{size}.npz"), **out)
        print(f"G3 raft size={size}: {len(out)} arrays")


# ----------------------------------------------------------------------------------------------- G4 gradients
def g4():
    size, b = 64, 2
    out = {}
    rf = RaftFlow(**cases.raft_cfg(size))
    sd = load(rf, "rf")
    rf.train(True)
    kp_s, kp_d, dmo, img, img_full = run_raft(rf, sd, size, b, True, "g4/raft")
    driving = cases.images("g4/drv", b, size)
    leaves = [kp_s, kp_d, dmo["deformation"], dmo["occlusion"]]
    for t in leaves:
        t.requires_grad_(True)
    o_ref, _, _ = rf(kp_s, kp_d, dmo, img, img_full)
    loss = (o_ref - driving).abs().mean()
    loss.backward()
    out["loss"] = np.array([loss.item()], np.float32)
    for n, t in zip(("kp_s", "kp_d", "deformation", "occlusion"), leaves):
        out[f"grad_{n}"] = npy(t.grad)
    names, norms = [], []
    for n, p in rf.named_parameters():
        names.append(n)
        norms.append(0.0 if p.grad is None else p.grad.norm().item())
    out["param_grad_norms"] = np.array(norms, np.float32)
    for n in ("refine.conv2.weight", "refine.convo2.bias", "to_context.5.weight", "generator.final.weight",
              "corr_enc.convf1.weight", "kp_head.weight", "generator.first.conv.weight", "pos_embedding"):
        out["pgrad_" + n] = npy(dict(rf.named_parameters())[n].grad)
    with open(os.path.join(GOLD, "grads_64_param_names.json"), "w") as f:
        json.dump(names, f)
    # oracle gradients
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    l2 = [t.detach().clone().requires_grad_(True) for t in leaves]
    o, _, _ = O.raft_flow(l2[0], l2[1], {"deformation": l2[2], "occlusion": l2[3]}, img, img_full, P, "", size=size, train=True)
    lo = (o - driving).abs().mean()
    lo.backward()
    print(f"   loss ref {loss.item():.6f} oracle {lo.item():.6f}")
    for n, t, t2 in zip(("kp_s", "kp_d", "deformation", "occlusion"), leaves, l2):
        delta(f"grad {n}", t2.grad, t.grad)
    worst = 0.0
    for n, p in rf.named_parameters():
        g = P[n].grad
        if p.grad is None:
            continue
        rel = (g - p.grad).norm().item() / (p.grad.norm().item() + 1e-12)
        worst = max(worst, rel)
    print(f"   worst relative param-grad error oracle-vs-ref: {worst:.3e}")
    np.savez_compressed(os.path.join(GOLD, "grads_64.npz"), **out)
    print("G4 gradient goldens:", len(out))


# ----------------------------------------------------------------------------------------------- G5 MTIA prior
def g5_tokenpose():
    """TokenPose_B (modules/transformer/pose_tokenpose_b.py) at vox1.yaml's mtia_kp_detector: state_dict manifest, eval and
    train-mode outputs + BatchNorm buffers after the train forward, eval-mode gradients (strict: well conditioned) and
    train-mode gradients (loose: batch-statistics cancellation makes fp32 itself ~1e-2 relative, measured vs fp64)."""
    import contextlib
    import io
    from modules.transformer.pose_tokenpose_b import get_pose_net
    from modules.util import convert_dict_to_attrit_dict
    from oracle import tokenpose_oracle as TO
    with contextlib.redirect_stdout(io.StringIO()):          # the reference prints its config
        net = get_pose_net(convert_dict_to_attrit_dict(cases.tokenpose_cfg()), is_train=True)
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    man["TokenPose_B"] = manifest(net)
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0)
    sd = cases.tokenpose_weights(net.state_dict(), "tp")
    b = 2
    x = cases.images("g5/img", b, 256)
    gk, gj = det_uniform("g5/gk", (b, 10, 2)), det_uniform("g5/gj", (b, 10, 2, 2))
    out, names = {}, [n for n, _ in net.named_parameters()]
    for train in (False, True):
        sfx = "train" if train else "eval"
        net.load_state_dict(sd)
        net.train(train)
        net.zero_grad()
        r = net(x)
        ((r["kp"] * gk).sum() + (r["jacobian"] * gj).sum()).backward()
        out[f"kp_{sfx}"], out[f"jac_{sfx}"] = npy(r["kp"]), npy(r["jacobian"])
        out[f"param_grad_norms_{sfx}"] = np.array([0.0 if p.grad is None else p.grad.norm().item() for _, p in net.named_parameters()],
                                                  np.float32)
        for n in ("pre_feature.conv1.weight", "pre_feature.layer1.0.downsample.0.weight", "pre_feature.stage2.0.fuse_layers.1.0.0.0.weight",
                  "pre_feature.stage3.2.fuse_layers.2.0.1.0.weight", "pre_feature.stage3.3.branches.2.3.bn2.weight",
                  "transformer.keypoint_token", "transformer.patch_to_embedding.weight", "transformer.transformer.layers.0.0.fn.fn.to_qkv.weight",
                  "transformer.transformer.layers.11.1.fn.fn.net.3.bias", "transformer.transformer.layers.5.0.fn.norm.weight",
                  "transformer.mlp_head.1.weight", "transformer.mlp_head_jacobian.0.bias"):
            out[f"pgrad_{sfx}_" + n] = npy(dict(net.named_parameters())[n].grad)
        if train:
            bufs = dict(net.named_buffers())
            for n in ("pre_feature.bn1.running_mean", "pre_feature.bn2.running_var", "pre_feature.stage3.3.fuse_layers.0.2.1.running_var",
                      "pre_feature.stage2.0.fuse_layers.1.0.0.1.running_mean"):
                out["buf_" + n] = npy(bufs[n])
            assert int(bufs["pre_feature.bn1.num_batches_tracked"]) == 1
        # oracle
        P = {k: v.clone().requires_grad_(v.is_floating_point() and k in names and k != "transformer.pos_embedding") for k, v in sd.items()}
        o = TO.tokenpose_b(x, P, "", train)
        ((o["kp"] * gk).sum() + (o["jacobian"] * gj).sum()).backward()
        delta(f"TokenPose_B kp {sfx}", o["kp"], r["kp"])
        delta(f"TokenPose_B jacobian {sfx}", o["jacobian"], r["jacobian"])
        worst = max((P[n].grad - p.grad).norm().item() / (p.grad.norm().item() + 1e-12) for n, p in net.named_parameters() if p.grad is not None)
        print(f"   worst relative param-grad error oracle-vs-ref ({sfx}): {worst:.3e}")
        if train:
            for n in ("pre_feature.bn1.running_mean", "pre_feature.stage3.3.fuse_layers.0.2.1.running_var"):
                delta("buffer " + n, P[n], bufs[n])
    print(f"   |kp| mean {np.abs(out['kp_eval']).mean():.3f}  jac spread {out['jac_eval'].std():.3f}")
    with open(os.path.join(GOLD, "tokenpose_param_names.json"), "w") as f:
        json.dump(names, f)
    np.savez_compressed(os.path.join(GOLD, "tokenpose.npz"), **out)
    print("G5 TokenPose_B goldens:", len(out))


# ----------------------------------------------------------------------------------------------- G6 callers of the path
def _ref_function(path, name, extra_globals):
    """compile ONE function of a reference script that cannot be imported as a module here (its other imports -- imageio,
    skimage, lpips -- are absent) straight from the reference file; nothing of it is written anywhere"""
    import ast
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = dict(extra_globals)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def g6_callers():
    """normalize_kp (animate_ddp.py:17-37) and psnr (reconstruction.py:13-19) of the reference on fixed keypoint dicts"""
    from scipy.spatial import ConvexHull
    ref_norm = _ref_function(os.path.join(ref_import.REF_ROOT, "animate_ddp.py"), "normalize_kp", {"torch": torch, "np": np, "ConvexHull": ConvexHull})
    ref_psnr = _ref_function(os.path.join(ref_import.REF_ROOT, "reconstruction.py"), "psnr", {"torch": torch})
    out = {}
    ks, kd, k0 = cases.keypoints("g6/ks", 2), cases.keypoints("g6/kd", 2), cases.keypoints("g6/k0", 2)
    for adapt in (False, True):
        for rel, relj in ((False, False), (True, False), (True, True)):
            r = ref_norm({k: v.clone() for k, v in ks.items()}, {k: v.clone() for k, v in kd.items()}, {k: v.clone() for k, v in k0.items()},
                         adapt_movement_scale=adapt, use_relative_movement=rel, use_relative_jacobian=relj)
            tag = f"a{int(adapt)}_r{int(rel)}_j{int(relj)}"
            out[f"norm_kp_{tag}"], out[f"norm_jac_{tag}"] = npy(r["kp"]), npy(r["jacobian"])
    a, b = cases.images("g6/a", 2, 32), cases.images("g6/b", 2, 32)
    out["psnr"] = np.array([float(ref_psnr(a, b))], np.float32)
    np.savez_compressed(os.path.join(GOLD, "callers.npz"), **out)
    print("G6 caller goldens:", len(out))


# ----------------------------------------------------------------------------------------------- G7 training losses
def g7_losses():
    """The reference's OWN MRFA.forward(is_train=True) (model.py:183-257: Vgg19 79-121, ImagePyramide, Transform, loss wiring)
    on CPU, fomm prior, B=1, deterministic weights.  torchvision is absent: `models.vgg19` is bound to the published VGG19
    architecture (oracle/losses_oracle.py) with deterministic weights, `.cuda()` is neutralised (model.py:155,157)."""
    import modules.model as RM
    from oracle import losses_oracle as LO
    from mrfa_amd.train import VOX1

    class _Features(torch.nn.Module):            # stands in for torchvision.models.vgg19(pretrained=True): only .features is used
        def __init__(self):
            super().__init__()
            layers, cin = [], 3
            for v in LO.VGG19_CFG:
                if v == 'M':
                    layers.append(torch.nn.MaxPool2d(2, 2))
                else:
                    layers += [torch.nn.Conv2d(cin, v, 3, padding=1), torch.nn.ReLU(inplace=True)]
                    cin = v
            self.features = torch.nn.Sequential(*layers)
    RM.models.vgg19 = lambda pretrained=True: _Features()
    old_cuda = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **k: self
    made = []
    old_init = RM.Transform.__init__

    def rec_init(self, bs, **kw):
        old_init(self, bs, **kw)
        made.append(self)
    RM.Transform.__init__ = rec_init
    try:
        cfg = RU.convert_dict_to_attrit_dict({
            "fomm_kp_detector": cases.KP_DETECTOR_CFG, "dense_motion": cases.DENSE_MOTION_CFG, "raft_flow": cases.raft_cfg(256),
            "train_params": dict(prior_model="fomm", num_epochs=100, bg_start=1000, scales=[1, 0.5, 0.25, 0.125], clip=10, lr=2.0e-4,
                                 transform_params=dict(sigma_affine=0.05, sigma_tps=0.005, points_tps=5),
                                 loss_weights=dict(perceptual=[10, 10, 10, 10, 10], equivariance=10, equivariance_jacobian=10))})
        m = RM.MRFA(cfg)
    finally:
        torch.nn.Module.cuda = old_cuda
    load(m.encoder, "encoder.")
    load(m.dense_motion, "dense_motion.")
    load(m.decoder, "decoder.")
    vsd = cases.vgg_weights(m.vgg.state_dict())
    m.vgg.load_state_dict(vsd)
    m.train(True)
    b = 1
    x = {"source": cases.images("g7/src", b, 256), "driving": cases.images("g7/drv", b, 256)}
    torch.manual_seed(1234)
    gen, warp_img, lv, kp_s, kp_d = m(x, epoch=0, is_train=True)
    total = sum(v.mean() for v in lv.values())
    total.backward()
    tr = made[-1]
    out = {"perceptual": np.array([lv["perceptual"].item()], np.float32), "equivariance": np.array([lv["equivariance"].item()], np.float32),
           "equivariance_jacobian": npy(lv["equivariance_jacobian"]), "gen_s4": npy(gen[:, :, ::4, ::4]),
           "theta": npy(tr.theta), "control_points": npy(tr.control_points), "control_params": npy(tr.control_params),
           "kp_d": npy(kp_d)}
    names = [n for n, p in m.named_parameters() if p.grad is not None]
    out["param_grad_norms"] = np.array([dict(m.named_parameters())[n].grad.norm().item() for n in names], np.float32)
    with open(os.path.join(GOLD, "losses_param_names.json"), "w") as f:
        json.dump(names, f)
    # stand-alone pieces on fixed inputs (so that the loss module can be checked without the generator in front of it)
    g_in, r_in = cases.images("g7/gen", b, 256).requires_grad_(True), cases.images("g7/real", b, 256)
    pg, pr = m.pyramid(g_in), m.pyramid(r_in)
    val = 0
    for sc in m.scales:
        xv, yv = m.vgg(pg["prediction_" + str(sc)]), m.vgg(pr["prediction_" + str(sc)])
        for i, w in enumerate(m.loss_weights["perceptual"]):
            val = val + w * torch.abs(xv[i] - yv[i].detach()).mean()
    val.backward()
    out["alone_perceptual"] = np.array([val.item()], np.float32)
    out["alone_dgen_s4"] = npy(g_in.grad[:, :, ::4, ::4])
    out["alone_dgen_norm"] = np.array([g_in.grad.norm().item()], np.float32)
    for sc in m.scales:
        out[f"pyr_{sc}_s2"] = npy(pr["prediction_" + str(sc)][:, :, ::2, ::2])
    feats = m.vgg(r_in)
    for i, f in enumerate(feats):
        out[f"vgg_{i}_mean"] = np.array([f.mean().item(), f.abs().max().item()], np.float32)
    kq = cases.keypoints("g7/kq", b)
    out["warp_kp"] = npy(tr.warp_coordinates(kq["kp"]))
    out["warp_jac"] = npy(tr.jacobian(kq["kp"].clone().requires_grad_(True)))
    out["warp_frame_s4"] = npy(tr.transform_frame(r_in)[:, :, ::4, ::4])
    # oracle deltas
    P = {k: v.clone() for k, v in vsd.items()}
    o_val = LO.perceptual(g_in.detach(), r_in, P, m.scales, m.loss_weights["perceptual"])
    print(f"   perceptual ref {val.item():.6f} oracle {o_val.item():.6f}")
    delta("warp_coordinates", LO.warp_coordinates(kq["kp"], tr.theta, tr.control_points, tr.control_params), tr.warp_coordinates(kq["kp"]))
    delta("transform_frame", LO.transform_frame(r_in, tr.theta, tr.control_points, tr.control_params), tr.transform_frame(r_in))
    np.savez_compressed(os.path.join(GOLD, "losses.npz"), **out)
    print("G7 loss goldens:", len(out), {k: float(v.mean()) for k, v in lv.items()})


# ----------------------------------------------------------------------------------------------- G8 background motion
def g8_background():
    """The reference's BGMotionPredictor (bg_motion_predictor.py:5-24), DenseMotionNetwork with bg_param (dense_motion.py:67-73) and
    the background loss (model.py:248-253).  torchvision is absent: `models.resnet18` is bound to the published resnet18
    architecture restated below (the reference's class swaps its stem and fc as usual)."""
    import modules.bg_motion_predictor as RB

    class _Block(torch.nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(cout)
            self.relu = torch.nn.ReLU(inplace=True)
            self.conv2 = torch.nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = torch.nn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = torch.nn.Sequential(torch.nn.Conv2d(cin, cout, 1, stride, bias=False), torch.nn.BatchNorm2d(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + idt)

    class _ResNet18(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(64)
            self.relu = torch.nn.ReLU(inplace=True)
            self.maxpool = torch.nn.MaxPool2d(3, 2, 1)
            cin = 64
            for k, (c, st) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
                setattr(self, f"layer{k}", torch.nn.Sequential(_Block(cin, c, st), _Block(c, c, 1)))
                cin = c
            self.avgpool = torch.nn.AdaptiveAvgPool2d((1, 1))
            self.fc = torch.nn.Linear(512, 1000)

        def forward(self, x):
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
            return self.fc(torch.flatten(self.avgpool(x), 1))
    RB.models.resnet18 = lambda pretrained=False: _ResNet18()
    bg = RB.BGMotionPredictor()
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    man["BGMotionPredictor"] = manifest(bg)
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0)
    sd = cases.bg_weights(bg.state_dict())
    out = {}
    b = 2
    src, drv = cases.images("g8/src", b, 256), cases.images("g8/drv", b, 256)
    from oracle import losses_oracle as LO
    for train in (False, True):
        sfx = "train" if train else "eval"
        bg.load_state_dict(sd)
        bg.train(train)
        bg.zero_grad()
        fwd, rev = bg(src, drv), bg(drv, src)
        value = torch.matmul(fwd, rev)
        loss = 10 * torch.abs(torch.eye(3).view(1, 3, 3) - value).mean()
        loss.backward()
        out[f"bg_{sfx}"], out[f"bg_rev_{sfx}"], out[f"bg_loss_{sfx}"] = npy(fwd), npy(rev), np.array([loss.item()], np.float32)
        out[f"param_grad_norms_{sfx}"] = np.array([p.grad.norm().item() for _, p in bg.named_parameters()], np.float32)
        P = {k: v.clone() for k, v in sd.items()}
        o = LO.bg_motion_predictor(src, drv, P, "", train)
        delta(f"BGMotionPredictor {sfx}", o, fwd)
        if train:
            out["buf_bn1_running_var"] = npy(dict(bg.named_buffers())["bg_encoder.bn1.running_var"])
    with open(os.path.join(GOLD, "bg_param_names.json"), "w") as f:
        json.dump([n for n, _ in bg.named_parameters()], f)
    # dense motion with a background transform
    dmm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    dsd = load(dmm, "dm")
    kd, ks = cases.keypoints("g8/kd", b), cases.keypoints("g8/ks", b)
    bgp = torch.from_numpy(out["bg_eval"])
    dmm.eval()
    with torch.no_grad():
        r = dmm(src, kd, ks, bg_param=bgp)
        sm = dmm.create_sparse_motions(torch.zeros(b, 3, 8, 8), kd, ks, bg_param=bgp)
    out["dm_bg_deformation"], out["dm_bg_occlusion"], out["sparse_motions_bg"] = npy(r["deformation"]), npy(r["occlusion"]), npy(sm)
    delta("sparse_motions(bg)", O.sparse_motions(kd, ks, 8, 8, bgp), sm)
    with torch.no_grad():
        od = O.dense_motion(src, kd, ks, {k: v.clone() for k, v in dsd.items()}, "", False, bg_param=bgp)
    delta("DenseMotion(bg) deformation", od["deformation"], r["deformation"])
    np.savez_compressed(os.path.join(GOLD, "background.npz"), **out)
    print("G8 background goldens:", len(out), "bg_eval[0] =", out["bg_eval"][0].round(3).tolist())


# ----------------------------------------------------------------------------------------------- G9 whole-MRFA key layout
def g9_mrfa_manifest():
    """state_dict manifests of the reference's OWN MRFA (model.py:145-181) built from its two YAML files: vox1.yaml (MTIA prior, no
    background predictor) and celebvhq.yaml (bg_start 0 -> BGMotionPredictor), plus the FOMM-prior variant of vox1.  These pin the
    checkpoint key set demo.py / Logger.load_cpk load with strict=True (`pyramid.*`, `vgg.*` at the top level, then the networks).
    torchvision is absent: vgg19 / resnet18 are bound to the published architectures as in g7 / g8 (shapes and names only matter)."""
    import copy
    import yaml
    import modules.model as RM
    import modules.bg_motion_predictor as RB
    from oracle import losses_oracle as LO

    class _Features(torch.nn.Module):
        def __init__(self):
            super().__init__()
            layers, cin = [], 3
            for v in LO.VGG19_CFG:
                if v == 'M':
                    layers.append(torch.nn.MaxPool2d(2, 2))
                else:
                    layers += [torch.nn.Conv2d(cin, v, 3, padding=1), torch.nn.ReLU(inplace=True)]
                    cin = v
            self.features = torch.nn.Sequential(*layers)

    class _Block(torch.nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(cout)
            self.conv2 = torch.nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = torch.nn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = torch.nn.Sequential(torch.nn.Conv2d(cin, cout, 1, stride, bias=False), torch.nn.BatchNorm2d(cout))

    class _ResNet18(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(64)
            cin = 64
            for k, (c, st) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
                setattr(self, f"layer{k}", torch.nn.Sequential(_Block(cin, c, st), _Block(c, c, 1)))
                cin = c
            self.fc = torch.nn.Linear(512, 1000)
    RM.models.vgg19 = lambda pretrained=True: _Features()
    RB.models.resnet18 = lambda pretrained=False: _ResNet18()
    RM.BGMotionPredictor = RB.BGMotionPredictor
    old_cuda = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **k: self
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    try:
        for name, yml, prior in (("MRFA_vox1", "vox1.yaml", None), ("MRFA_celebvhq", "celebvhq.yaml", None), ("MRFA_vox1_fomm", "vox1.yaml", "fomm")):
            cfg = yaml.safe_load(open(os.path.join(ref_import.REF_ROOT, "config", yml)))
            if prior:
                cfg = copy.deepcopy(cfg)
                cfg["train_params"]["prior_model"] = prior
            m = RM.MRFA(RU.convert_dict_to_attrit_dict(cfg))
            man[name] = manifest(m)
            print(f"   {name}: {len(man[name])} entries, first {man[name][0][0]}, has bg_predictor {hasattr(m, 'bg_predictor')}")
    finally:
        torch.nn.Module.cuda = old_cuda
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0)



# ----------------------------------------------------------------------------------------------- G10 reference helper surface
def g10_helpers():
    """Forward values AND autograd gradients of the reference's own small callables: bilinear_sampler / batch_bilinear_sampler /
    coords_grid (util.py:26-56), CorrBlock (raft.py:12-48), BasicMotionEncoder.forward (raft.py:60-68), RefineFlow.forward
    (raft.py:80-88) -- the names VERDICT r1 lists as missing from mrfa_amd.modules."""
    from modules.raft import BasicMotionEncoder, RefineFlow
    out = {}
    img = det_uniform("h/bs_img", (2, 4, 6, 7), -1, 1).requires_grad_(True)
    xy = det_uniform("h/bs_xy", (2, 5, 6, 2), -1.5, 7.5).requires_grad_(True)
    y, m = RU.bilinear_sampler(img, xy, mask=True)
    (y * det_uniform("h/bs_w", tuple(y.shape), -1, 1)).sum().backward()
    out["bs_out"], out["bs_mask"], out["bs_dimg"], out["bs_dxy"] = npy(y), npy(m), npy(img.grad), npy(xy.grad)
    maps_b = det_uniform("h/bbs_img", (2 * 4 * 4, 1, 6, 6), -1, 1)
    xy_b = det_uniform("h/bbs_xy", (2 * 4 * 4, 7, 7, 2), -1.0, 6.0)
    out["bbs_out"] = npy(RU.batch_bilinear_sampler(maps_b, xy_b, h=4, w=4, mini_batch=1))
    out["coords_grid"] = npy(RU.coords_grid(2, 3, 5, "cpu"))
    maps = det_uniform("h/corr_maps", (2 * 4 * 5, 1, 8, 8), -1, 1).requires_grad_(True)
    cxy = det_uniform("h/corr_xy", (2, 2, 4, 5), -2.0, 9.0).requires_grad_(True)
    c = CorrBlock(maps)(cxy)
    (c * det_uniform("h/corr_w", tuple(c.shape), -1, 1)).sum().backward()
    out["cb_out"], out["cb_dmaps"], out["cb_dxy"] = npy(c), npy(maps.grad), npy(cxy.grad)
    enc = BasicMotionEncoder()
    load(enc, "h/enc")
    flow = det_uniform("h/enc_flow", (2, 2, 8, 8), -3, 3).requires_grad_(True)
    corr = det_uniform("h/enc_corr", (2, 98, 8, 8), -1, 1).requires_grad_(True)
    o = enc(flow, corr)
    (o * det_uniform("h/enc_w", tuple(o.shape), -1, 1)).sum().backward()
    out["enc_out"], out["enc_dflow"], out["enc_dcorr"] = npy(o), npy(flow.grad), npy(corr.grad)
    out["enc_pgrad_norms"] = np.array([p.grad.norm().item() for _, p in enc.named_parameters()], np.float32)
    ref = RefineFlow()
    load(ref, "h/ref")
    mf = det_uniform("h/ref_mf", (2, 128, 8, 8), -1, 1).requires_grad_(True)
    wf = det_uniform("h/ref_wf", (2, 192, 8, 8), -1, 1).requires_grad_(True)
    d, inp = ref(mf, wf)
    ((d * det_uniform("h/ref_w", tuple(d.shape), -1, 1)).sum() + (inp * det_uniform("h/ref_wi", tuple(inp.shape), -0.1, 0.1)).sum()).backward()
    out["ref_out"], out["ref_inp"], out["ref_dmf"], out["ref_dwf"] = npy(d), npy(inp), npy(mf.grad), npy(wf.grad)
    out["ref_pgrad_norms"] = np.array([p.grad.norm().item() for _, p in ref.named_parameters()], np.float32)
    np.savez_compressed(os.path.join(GOLD, "helpers.npz"), **out)
    print("G10 helper goldens:", {k: v.shape for k, v in out.items()})


# ----------------------------------------------------------------------------------------------- G11 prior-stage + chained gradients
def _grad_record(out, tag, mods, samples):
    """per-parameter gradient norms of `mods` = [(prefix, module)] and a few whole gradient tensors"""
    names, norms = [], []
    for pfx, m in mods:
        for n, p in m.named_parameters():
            names.append(pfx + n)
            norms.append(0.0 if p.grad is None else p.grad.norm().item())
    out[f"{tag}_pgrad_norms"] = np.array(norms, np.float32)
    allp = {pfx + n: p for pfx, m in mods for n, p in m.named_parameters()}
    for n in samples:
        out[f"{tag}_pgrad_{n}"] = npy(allp[n].grad)
    return names


def g11_prior_grads():
    """VERDICT r1 'weak' item 1: backward goldens from the reference's autograd for KPDetector (kp_detector.py:102-133) and
    DenseMotionNetwork (dense_motion.py:104-146) alone, and for the chained KPDetector -> DenseMotionNetwork -> RaftFlow pipeline at
    256 x 256 (model.py:185-210 with is_train wiring, loss = mean|out - driving|), eval-mode BatchNorm at B=2 and train-mode at B=4.
    The fp64 oracle's gradient norms are stored next to the reference's so that a test can see how far fp32 itself is from the truth."""
    out, names = {}, {}
    for train, b in ((False, 2), (True, 4)):
        sfx = "train" if train else "eval"
        # ---- KPDetector
        kpm = KPDetector(**cases.KP_DETECTOR_CFG)
        sd = load(kpm, "kp")
        kpm.train(train)
        x = cases.images(f"g11/x_{sfx}", b, 256)
        r = kpm(x)
        loss = (r["kp"] * det_uniform("g11/wkp", (b, 10, 2), -1, 1)).sum() + (r["jacobian"] * det_uniform("g11/wjac", (b, 10, 2, 2), -1, 1)).sum()
        loss.backward()
        out[f"kp_{sfx}_loss"] = np.array([loss.item()], np.float32)
        names[f"kp_{sfx}"] = _grad_record(out, f"kp_{sfx}", [("", kpm)], ["kp.weight", "jacobian.bias", "predictor.decoder.up_blocks.4.conv.weight",
                                                                         "predictor.encoder.down_blocks.0.norm.weight"])
        P = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        r64 = O.kp_detector(x.double(), P, "", train)
        ((r64["kp"] * det_uniform("g11/wkp", (b, 10, 2), -1, 1).double()).sum() + (r64["jacobian"] * det_uniform("g11/wjac", (b, 10, 2, 2), -1, 1).double()).sum()).backward()
        out[f"kp_{sfx}_pgrad_norms_fp64"] = np.array([0.0 if P[n].grad is None else P[n].grad.norm().item() for n in names[f"kp_{sfx}"]], np.float32)
        # ---- DenseMotionNetwork
        dmm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
        sdd = load(dmm, "dm")
        dmm.train(train)
        kd, ks = cases.keypoints(f"g11/kd_{sfx}", b), cases.keypoints(f"g11/ks_{sfx}", b)
        for d in (kd, ks):
            for t in d.values():
                t.requires_grad_(True)
        r = dmm(x, kd, ks)
        loss = ((r["deformation"] * det_uniform("g11/wdef", (b, 64, 64, 2), -1, 1)).sum() + (r["occlusion"] * det_uniform("g11/wocc", (b, 1, 64, 64), -1, 1)).sum()
                + (r["mask"] * det_uniform("g11/wmask", (b, 11, 64, 64), -1, 1)).sum()) / 64.0
        loss.backward()
        out[f"dm_{sfx}_loss"] = np.array([loss.item()], np.float32)
        for nm, d in (("kd", kd), ("ks", ks)):
            out[f"dm_{sfx}_grad_{nm}_kp"], out[f"dm_{sfx}_grad_{nm}_jac"] = npy(d["kp"].grad), npy(d["jacobian"].grad)
        names[f"dm_{sfx}"] = _grad_record(out, f"dm_{sfx}", [("", dmm)], ["mask.weight", "occlusion.bias", "hourglass.encoder.down_blocks.0.conv.weight",
                                                                         "hourglass.decoder.up_blocks.4.norm.bias"])
        P = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sdd.items()}
        kd64 = {k: v.detach().double().requires_grad_(True) for k, v in kd.items()}
        ks64 = {k: v.detach().double().requires_grad_(True) for k, v in ks.items()}
        r64 = O.dense_motion(x.double(), kd64, ks64, P, "", train)
        (((r64["deformation"] * det_uniform("g11/wdef", (b, 64, 64, 2), -1, 1).double()).sum() + (r64["occlusion"] * det_uniform("g11/wocc", (b, 1, 64, 64), -1, 1).double()).sum()
          + (r64["mask"] * det_uniform("g11/wmask", (b, 11, 64, 64), -1, 1).double()).sum()) / 64.0).backward()
        for nm, d in (("kd", kd64), ("ks", ks64)):
            out[f"dm_{sfx}_grad_{nm}_kp_fp64"], out[f"dm_{sfx}_grad_{nm}_jac_fp64"] = npy(d["kp"].grad), npy(d["jacobian"].grad)
        out[f"dm_{sfx}_pgrad_norms_fp64"] = np.array([0.0 if P[n].grad is None else P[n].grad.norm().item() for n in names[f"dm_{sfx}"]], np.float32)
        for nm in ("kd", "ks"):
            for short in ("kp", "jac"):
                a_, t_ = out[f"dm_{sfx}_grad_{nm}_{short}"], out[f"dm_{sfx}_grad_{nm}_{short}_fp64"]
                print(f"   DenseMotion {sfx} d{nm}.{short}: reference-vs-fp64 max {np.abs(a_ - t_).max():.2e} of {np.abs(t_).max():.2e}")
        # ---- chained pipeline, 256 x 256
        kpm = KPDetector(**cases.KP_DETECTOR_CFG)
        dmm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
        rf = RaftFlow(**cases.raft_cfg(256))
        sds = {"encoder.": load(kpm, "kp"), "dense_motion.": load(dmm, "dm"), "decoder.": load(rf, "rf")}
        for m in (kpm, dmm, rf):
            m.train(train)
        src, drv = cases.images(f"g11/src_{sfx}", b, 256), cases.images(f"g11/drv_{sfx}", b, 256)
        down = RU.AntiAliasInterpolation2d(3, 0.25)
        k_s, k_d = kpm(src), kpm(drv)
        dmo = dmm(src, k_d, k_s)
        gen, _, _ = rf(k_s["kp"], k_d["kp"], dmo, img=down(src), img_full=src)
        loss = (gen - drv).abs().mean()
        loss.backward()
        out[f"chain_{sfx}_loss"] = np.array([loss.item()], np.float32)
        out[f"chain_{sfx}_gen_s4"] = npy(gen[:, :, ::4, ::4])
        names[f"chain_{sfx}"] = _grad_record(out, f"chain_{sfx}", [("encoder.", kpm), ("dense_motion.", dmm), ("decoder.", rf)],
                                              ["encoder.kp.weight", "encoder.jacobian.weight", "dense_motion.mask.weight", "decoder.refine.conv2.weight",
                                               "decoder.to_context.0.weight", "decoder.generator.final.weight", "decoder.pos_embedding"])
        # the same pipeline through the oracle in fp64: the truth the fp32 reference itself is measured against
        P = {}
        for pfx, sd_ in sds.items():
            P.update({pfx + k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd_.items()})
        g64, _, _, _, _ = O.mrfa_forward(src.double(), drv.double(), P, size=256, train=train, prior="fomm")
        l64 = (g64 - drv.double()).abs().mean()
        l64.backward()
        out[f"chain_{sfx}_gen_s4_fp64"] = npy(g64.detach().float()[:, :, ::4, ::4])      # the truth of the output the reference itself is measured against
        print(f"   chain {sfx} output: reference-vs-fp64 mean {np.abs(out[f'chain_{sfx}_gen_s4'] - out[f'chain_{sfx}_gen_s4_fp64']).mean():.2e} "
              f"max {np.abs(out[f'chain_{sfx}_gen_s4'] - out[f'chain_{sfx}_gen_s4_fp64']).max():.2e}")
        truth = np.array([0.0 if (n not in P or P[n].grad is None) else P[n].grad.norm().item() for n in names[f"chain_{sfx}"]], np.float64)
        out[f"chain_{sfx}_pgrad_norms_fp64"] = truth.astype(np.float32)
        # how far the fp32 reference's WHOLE sampled tensors are from the fp64 truth (relative L2): the norm of a tensor can agree to
        # 0.1 % while the tensor itself is 3 % away, so the whole-tensor checks need their own noise figure
        for key in [k for k in list(out) if k.startswith(f"chain_{sfx}_pgrad_") and not k.endswith("_fp64") and k != f"chain_{sfx}_pgrad_norms"]:
            n = key[len(f"chain_{sfx}_pgrad_"):]
            t_ = P[n].grad.detach().numpy().astype(np.float64)
            r_ = out[key].astype(np.float64)
            out[key + "_noise_fp64"] = np.array([np.linalg.norm(r_ - t_) / max(np.linalg.norm(t_), 1e-30)], np.float32)
            print(f"   chain {sfx} d/d {n}: reference-vs-fp64 relative L2 {float(out[key + '_noise_fp64'][0]):.2e}")
        refn = out[f"chain_{sfx}_pgrad_norms"].astype(np.float64)
        rel = np.abs(refn - truth) / np.maximum(truth, 1e-3 * truth.max())
        print(f"   chain {sfx}: loss ref {loss.item():.6f} fp64 oracle {l64.item():.6f}; reference-vs-fp64 gradient norms: median {np.median(rel):.2e} max {rel.max():.2e}")
    with open(os.path.join(GOLD, "prior_grads_param_names.json"), "w") as f:
        json.dump(names, f)
    np.savez_compressed(os.path.join(GOLD, "prior_grads.npz"), **out)
    print("G11 prior / chained gradient goldens:", len(out), {k: float(v[0]) for k, v in out.items() if k.endswith("_loss")})


# ----------------------------------------------------------------------------------------------- G12 the headline program: MTIA chain
def g13_kp_occlusion():
    """KPDetector(estimate_occlusion=True) of the reference (kp_detector.py:41-48,124-128; off in both YAMLs): outputs and autograd gradients, eval- and
    train-mode BatchNorm, B = 2 -- loss = weighted sums of kp, jacobian and kp_occlusion."""
    out, names = {}, {}
    cfg = dict(cases.KP_DETECTOR_CFG, estimate_occlusion=True)
    for train in (False, True):
        sfx = "train" if train else "eval"
        kpm = KPDetector(**cfg)
        load(kpm, "kpocc")
        kpm.train(train)
        x = cases.images(f"g13/x_{sfx}", 2, 256)
        r = kpm(x)
        loss = ((r["kp"] * det_uniform("g13/wkp", (2, 10, 2), -1, 1)).sum() + (r["jacobian"] * det_uniform("g13/wjac", (2, 10, 2, 2), -1, 1)).sum()
                + (r["kp_occlusion"] * det_uniform("g13/wocc", (2, 10, 1, 1), -1, 1)).sum())
        loss.backward()
        out[f"{sfx}_kp"], out[f"{sfx}_jac"], out[f"{sfx}_occ"] = npy(r["kp"]), npy(r["jacobian"]), npy(r["kp_occlusion"])
        out[f"{sfx}_loss"] = np.array([loss.item()], np.float32)
        names[sfx] = _grad_record(out, sfx, [("", kpm)], ["kp_occlusion.4.weight", "kp_occlusion.4.bias", "kp_occlusion.0.conv.weight", "kp_occlusion.3.norm.weight"])
    np.savez_compressed(os.path.join(GOLD, "kp_occlusion.npz"), **out)
    with open(os.path.join(GOLD, "kp_occlusion_param_names.json"), "w") as f:
        json.dump(names, f)
    print("G13 kp_occlusion goldens:", len(out), "occ eval", out["eval_occ"].reshape(2, 10)[0, :4])


def g12_chain_mtia():
    """the reference's OWN `MRFA` (model.py:145-216) built with `prior_model: mtia` -- TokenPose_B encoder on source and
    driving, DenseMotionNetwork, RaftFlow -- run forward (`MRFA.forward(x, epoch=0, is_train=False)`: the wiring of model.py:185-210
    without the loss modules) + the surrogate objective mean|gen - driving| of SURVEY 8(d) + `.backward()` (train.py:58-64), B=2,
    eval-mode BatchNorm (well conditioned) and train-mode (the benchmark's mode).  Stored: loss, sub-sampled output, keypoints and
    d loss / d keypoints (retained on the encoder's outputs), every parameter's gradient norm and <= 64 evenly spread entries of every
    parameter's gradient (cases.sample_index) -- and the same quantities from an fp64 run of the oracle, the truth that says how far the
    fp32 reference itself is from exact arithmetic (the band the GPU is allowed)."""
    import contextlib
    import io
    import modules.model as RM
    from mrfa_amd.train import VOX1
    old_cuda = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **k: self          # model.py:155: `.cuda()` on the pyramid
    try:
        cfg = RU.convert_dict_to_attrit_dict({
            "mtia_kp_detector": cases.tokenpose_cfg(), "dense_motion": cases.DENSE_MOTION_CFG, "raft_flow": cases.raft_cfg(256),
            "train_params": dict(prior_model="mtia", num_epochs=100, bg_start=1000, scales=[1, 0.5, 0.25, 0.125], clip=10, lr=2.0e-4,
                                 transform_params=dict(sigma_affine=0.05, sigma_tps=0.005, points_tps=5),
                                 loss_weights=dict(perceptual=[0, 0, 0, 0, 0], equivariance=0, equivariance_jacobian=0))})
        with contextlib.redirect_stdout(io.StringIO()):
            m = RM.MRFA(cfg)
    finally:
        torch.nn.Module.cuda = old_cuda
    sds = cases.mtia_chain_weights(m.encoder.state_dict(), m.dense_motion.state_dict(), m.decoder.state_dict())
    mods = [("encoder.", m.encoder), ("dense_motion.", m.dense_motion), ("decoder.", m.decoder)]
    names = [pfx + n for pfx, mod in mods for n, _ in mod.named_parameters()]
    out = {}
    b = 2
    kept = []
    hook = m.encoder.register_forward_hook(lambda mod, inp, o: (kept.append(o), [v.retain_grad() for v in o.values()])[0] and None)
    for train in (False, True):
        sfx = "train" if train else "eval"
        for pfx, mod in mods:
            mod.load_state_dict(sds[pfx])
        m.train(train)
        m.zero_grad()
        kept.clear()
        src, drv = cases.images(f"g12/src_{sfx}", b, 256), cases.images(f"g12/drv_{sfx}", b, 256)
        gen, warp_img, lv, kps, kpd = m({"source": src, "driving": drv}, epoch=0, is_train=False)
        loss = (gen - drv).abs().mean()
        loss.backward()
        k_s, k_d = kept
        out[f"{sfx}_loss"] = np.array([loss.item()], np.float64)
        out[f"{sfx}_gen_s4"] = npy(gen[:, :, ::4, ::4])
        for nm, d in (("s", k_s), ("d", k_d)):
            out[f"{sfx}_kp_{nm}"], out[f"{sfx}_jac_{nm}"] = npy(d["kp"]), npy(d["jacobian"])
            out[f"{sfx}_dkp_{nm}"], out[f"{sfx}_djac_{nm}"] = npy(d["kp"].grad), npy(d["jacobian"].grad)
        P32 = {pfx + n: p for pfx, mod in mods for n, p in mod.named_parameters()}
        out[f"{sfx}_pgrad_norms"] = np.array([0.0 if P32[n].grad is None else P32[n].grad.double().norm().item() for n in names], np.float64)
        out[f"{sfx}_pgrad_samples"] = np.concatenate([
            (np.zeros(len(cases.sample_index(P32[n].numel())), np.float32) if P32[n].grad is None
             else npy(P32[n].grad.reshape(-1)[cases.sample_index(P32[n].numel())])) for n in names])
        if train:
            bufs = {pfx + n: v for pfx, mod in mods for n, v in mod.named_buffers()}
            for n in ("encoder.pre_feature.bn1.running_mean", "encoder.pre_feature.stage3.3.fuse_layers.0.2.1.running_var",
                      "decoder.generator.first.norm.running_var", "dense_motion.hourglass.encoder.down_blocks.0.norm.running_mean"):
                out["train_buf_" + n] = npy(bufs[n])
        # ---- a SECOND fp32 realisation of the unmodified reference: the same modules with ATen's native convolution kernels instead of
        # oneDNN's (another summation order).  One fp32 run is one sample of the rounding noise (its distance from fp64 differed by up to
        # 10x between the two backends, sub-network by sub-network): the parity gates take the larger of the two as the noise level.
        for pfx, mod in mods:
            mod.load_state_dict(sds[pfx])
        m.train(train)
        m.zero_grad()
        kept.clear()
        with torch.backends.mkldnn.flags(enabled=False):
            gen2, _, _, _, _ = m({"source": src, "driving": drv}, epoch=0, is_train=False)
            (gen2 - drv).abs().mean().backward()
        k_s2, k_d2 = kept
        out[f"{sfx}_gen_s4_alt"] = npy(gen2[:, :, ::4, ::4])
        for nm, d in (("s", k_s2), ("d", k_d2)):
            out[f"{sfx}_dkp_{nm}_alt"], out[f"{sfx}_djac_{nm}_alt"] = npy(d["kp"].grad), npy(d["jacobian"].grad)
        out[f"{sfx}_pgrad_norms_alt"] = np.array([0.0 if P32[n].grad is None else P32[n].grad.double().norm().item() for n in names], np.float64)
        out[f"{sfx}_pgrad_samples_alt"] = np.concatenate([
            (np.zeros(len(cases.sample_index(P32[n].numel())), np.float32) if P32[n].grad is None
             else npy(P32[n].grad.reshape(-1)[cases.sample_index(P32[n].numel())])) for n in names])
        # ---- the same program through the oracle: fp32 (oracle-vs-reference delta printed) and fp64 (the truth)
        for dt_, key in ((torch.float32, None), (torch.float64, "fp64")):
            P = {}
            for pfx, sd_ in sds.items():
                P.update({pfx + k: ((v.to(dt_).clone().requires_grad_(pfx + k in names and not k.endswith("pos_embedding") or k == "pos_embedding"))
                                    if v.is_floating_point() else v.clone()) for k, v in sd_.items()})
            g_o, _, ks_o, kd_o, _ = O.mrfa_forward(src.to(dt_), drv.to(dt_), P, size=256, train=train, prior="mtia")
            for d in (ks_o, kd_o):
                for v in d.values():
                    v.retain_grad()
            l_o = (g_o - drv.to(dt_)).abs().mean()
            l_o.backward()
            gn = np.array([0.0 if P[n].grad is None else P[n].grad.double().norm().item() for n in names], np.float64)
            if key is None:
                delta(f"mtia chain {sfx} gen", g_o, gen)
                delta(f"mtia chain {sfx} kp_d", kd_o["kp"], k_d["kp"])
                rel = np.abs(gn - out[f"{sfx}_pgrad_norms"]) / np.maximum(out[f"{sfx}_pgrad_norms"], 1e-3 * out[f"{sfx}_pgrad_norms"].max())
                print(f"   oracle(fp32)-vs-ref {sfx}: loss {l_o.item():.7f} vs {loss.item():.7f}; gradient norms median {np.median(rel):.2e} max {rel.max():.2e}")
                continue
            out[f"{sfx}_loss_fp64"] = np.array([l_o.item()], np.float64)
            out[f"{sfx}_gen_s4_fp64"] = npy(g_o[:, :, ::4, ::4].float())
            out[f"{sfx}_pgrad_norms_fp64"] = gn
            out[f"{sfx}_pgrad_samples_fp64"] = np.concatenate([
                (np.zeros(len(cases.sample_index(P[n].numel())), np.float64) if P[n].grad is None
                 else P[n].grad.reshape(-1)[cases.sample_index(P[n].numel())].numpy().astype(np.float64)) for n in names])
            for nm, d in (("s", ks_o), ("d", kd_o)):
                out[f"{sfx}_dkp_{nm}_fp64"], out[f"{sfx}_djac_{nm}_fp64"] = d["kp"].grad.numpy(), d["jacobian"].grad.numpy()
            rel = np.abs(out[f"{sfx}_pgrad_norms"] - gn) / np.maximum(gn, 1e-3 * gn.max())
            print(f"   reference-vs-fp64 {sfx}: loss {loss.item():.7f} vs {l_o.item():.7f}; gradient norms median {np.median(rel):.2e} max {rel.max():.2e}; "
                  f"gen mean {np.abs(out[f'{sfx}_gen_s4'] - out[f'{sfx}_gen_s4_fp64']).mean():.2e}")
    hook.remove()
    with open(os.path.join(GOLD, "chain_mtia_param_names.json"), "w") as f:
        json.dump(names, f)
    np.savez_compressed(os.path.join(GOLD, "chain_mtia.npz"), **out)
    print("G12 MTIA chain goldens:", len(out), {k: float(v[0]) for k, v in out.items() if k.endswith("_loss")},
          "bytes", os.path.getsize(os.path.join(GOLD, "chain_mtia.npz")))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3_prior", "g3_raft", "g4", "g5_tokenpose", "g6_callers", "g7_losses", "g8_background", "g9_mrfa_manifest", "g10_helpers", "g11_prior_grads", "g12_chain_mtia", "g13_kp_occlusion"]
    for w in which:
        print("==", w)
        globals()[w]()
