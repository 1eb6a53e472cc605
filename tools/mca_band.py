"""Conditioning band of the headline program (tests/test_headline.py: MTIA chain, B = 2, train-mode BatchNorm) by Monte-Carlo arithmetic:
the fp64 oracle with every tensor-valued operation's result (convolutions, linears, BatchNorm / LayerNorm, samplers, resizes, pools, softmax,
matrix products) multiplied by (1 + e), e UNIFORM in [-u, u], u = 2^-24 = fp32's unit roundoff, forward AND backward: the error model of one
round-to-nearest per operation result (bounded by u, rms u / sqrt(3)) -- a LOWER bound of what a real fp32 implementation does, which rounds every
partial sum.  (Round 5 drew e from N(0, u): 1.7x the rms and unbounded tails; the bands here are correspondingly narrower.)  K runs -> per
sub-network the largest 1 - cos and relative norm deviation of the gradient from the unperturbed fp64 run: how far two correct fp32 implementations of THIS program may be
expected to disagree -- a property of the reference's arithmetic at these weights, measured without the implementation under test.
Stored in tests/golden/chain_mtia_mca.npz (+ _groups.json); tests/headline_checks.reference_band() merges it with the reference's own
fp32-vs-fp64 distances.        python tools/mca_band.py [runs=8]   |   python tools/mca_band.py fomm [runs=8]"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mrfa_oracle as O  # noqa: E402
from tests import cases, headline_checks as H  # noqa: E402

U = 2.0 ** -24


class Jitter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * (1.0 + U * (2.0 * torch.rand_like(x) - 1.0))

    @staticmethod
    def backward(ctx, g):
        return g * (1.0 + U * (2.0 * torch.rand_like(g) - 1.0))


def jittered(fn):
    def w(*a, **k):
        y = fn(*a, **k)
        return Jitter.apply(y) if torch.is_tensor(y) and y.is_floating_point() and y.requires_grad else y
    return w


PATCH_F = ["conv2d", "linear", "batch_norm", "layer_norm", "grid_sample", "interpolate", "avg_pool2d", "softmax", "gelu", "pad"]
PATCH_T = ["matmul", "einsum", "bmm"]


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_num_threads(8)
    g, names = H.load_golden(os.path.join(ROOT, "tests", "golden"))
    from mrfa_amd.train import VOX1, HotPath
    m = HotPath(VOX1, prior="mtia")
    sds = cases.mtia_chain_weights(m.encoder.state_dict(), m.dense_motion.state_dict(), m.decoder.state_dict())
    numels = {pfx + k: v.numel() for pfx, sd in sds.items() for k, v in sd.items()}
    del m
    src, drv = cases.images("g12/src_train", 2, 256).double(), cases.images("g12/drv_train", 2, 256).double()
    segs, _ = H.sample_segments(names, numels)

    def run():
        P = {}
        for pfx, sd_ in sds.items():
            # (same trainable set as the golden's fp64 run, tools/make_goldens.py:g12: the transformer's sine position code is a constant)
            P.update({pfx + k: ((v.double().clone().requires_grad_(pfx + k in names and not k.endswith("pos_embedding") or k == "pos_embedding"))
                                if v.is_floating_point() else v.clone()) for k, v in sd_.items()})
        gen, *_ = O.mrfa_forward(src, drv, P, size=256, train=True, prior="mtia")
        (gen - drv).abs().mean().backward()
        return H.sampled({n: P[n].grad for n in names}, names, numels), H.norms_of({n: P[n].grad for n in names}, names)

    base_s, base_n = run()
    # the unperturbed run must BE the golden's fp64 run (same program, same weights)
    assert np.allclose(base_n, g["train_pgrad_norms_fp64"], rtol=1e-9, atol=1e-12), np.abs(base_n - g["train_pgrad_norms_fp64"]).max()
    saved = {k: getattr(F, k) for k in PATCH_F}
    saved_t = {k: getattr(torch, k) for k in PATCH_T}
    groups = sorted({H.subnet(n) for n in names})
    worst = {grp: [[], []] for grp in groups}          # per run: (1 - cos, relative norm deviation)
    try:
        for k in PATCH_F:
            setattr(F, k, jittered(saved[k]))
        for k in PATCH_T:
            setattr(torch, k, jittered(saved_t[k]))
        for r in range(runs):
            torch.manual_seed(1000 + r)
            s, n = run()
            tab = H.group_table(names, segs, s, n, base_s, base_n)
            for grp, (cd, nr, _) in tab.items():
                worst[grp][0].append(cd)
                worst[grp][1].append(nr)
            print(f"run {r}: " + "  ".join(f"{grp.split('.')[-1]} {tab[grp][1]:.1e}" for grp in groups), flush=True)
    finally:
        for k, v in saved.items():
            setattr(F, k, v)
        for k, v in saved_t.items():
            setattr(torch, k, v)
    # the band = the 90th percentile over the runs (a quantile, not the largest of K draws); the maxima are kept beside it
    q = lambda v: float(np.quantile(np.array(v), 0.9))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chain_mtia_mca.npz"), train_cos=np.array([q(worst[grp][0]) for grp in groups]),
                        train_norm=np.array([q(worst[grp][1]) for grp in groups]), train_cos_max=np.array([max(worst[grp][0]) for grp in groups]),
                        train_norm_max=np.array([max(worst[grp][1]) for grp in groups]), runs=np.array([runs]), u=np.array([U]),
                        model=np.array(["uniform[-u,u] per operation result, q90 over runs"]))
    with open(os.path.join(ROOT, "tests", "golden", "chain_mtia_mca_groups.json"), "w") as f:
        json.dump(groups, f)
    for grp in groups:
        print(f"{grp:28s} 1-cos q90 {q(worst[grp][0]):.2e} max {max(worst[grp][0]):.2e}  |g| rel q90 {q(worst[grp][1]):.2e} max {max(worst[grp][1]):.2e}")


def main_fomm_chain():
    """the same band for tests/grad_checks.check_chained_pipeline_gradients (KPDetector -> DenseMotion -> RaftFlow, B = 4, train-mode BatchNorm;
    golden tools/make_goldens.py:g11_prior_grads): PER PARAMETER, the largest relative deviation of its gradient norm from the unperturbed fp64 run
    (relative to max(norm, 1e-3 x the largest norm), the scale the test uses) -> tests/golden/prior_grads_mca.npz"""
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.set_num_threads(8)
    from mrfa_amd.modules import DenseMotionNetwork, KPDetector, RaftFlow
    g = np.load(os.path.join(ROOT, "tests", "golden", "prior_grads.npz"))
    names = json.load(open(os.path.join(ROOT, "tests", "golden", "prior_grads_param_names.json")))["chain_train"]
    mods = [("encoder.", KPDetector(**cases.KP_DETECTOR_CFG), "kp"), ("dense_motion.", DenseMotionNetwork(**cases.DENSE_MOTION_CFG), "dm"),
            ("decoder.", RaftFlow(**cases.raft_cfg(256)), "rf")]
    sds = {pfx: cases.weights_for(m.state_dict(), tag) for pfx, m, tag in mods}
    src, drv = cases.images("g11/src_train", 4, 256).double(), cases.images("g11/drv_train", 4, 256).double()

    def run():
        P = {}
        for pfx, sd_ in sds.items():
            P.update({pfx + k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd_.items()})
        gen, *_ = O.mrfa_forward(src, drv, P, size=256, train=True, prior="fomm")
        (gen - drv).abs().mean().backward()
        return np.array([0.0 if (n not in P or P[n].grad is None) else P[n].grad.norm().item() for n in names], np.float64)
    base = run()
    truth = g["chain_train_pgrad_norms_fp64"].astype(np.float64)
    assert np.allclose(base, truth, rtol=1e-5, atol=1e-9), np.abs(base - truth).max()          # (the golden stores the fp64 norms as float32)
    scale = np.maximum(base, 1e-3 * base.max())
    saved = {k: getattr(F, k) for k in PATCH_F}
    saved_t = {k: getattr(torch, k) for k in PATCH_T}
    devs = []
    try:
        for k in PATCH_F:
            setattr(F, k, jittered(saved[k]))
        for k in PATCH_T:
            setattr(torch, k, jittered(saved_t[k]))
        for r in range(runs):
            torch.manual_seed(2000 + r)
            d = np.abs(run() - base) / scale
            devs.append(d)
            print(f"run {r}: median {np.median(d):.2e} max {d.max():.2e} ({names[int(np.argmax(d))]})", flush=True)
    finally:
        for k, v in saved.items():
            setattr(F, k, v)
        for k, v in saved_t.items():
            setattr(torch, k, v)
    worst = np.quantile(np.stack(devs), 0.9, axis=0)          # (q90 over the runs; the maxima beside it)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "prior_grads_mca.npz"), chain_train_norm=worst, chain_train_norm_max=np.stack(devs).max(0),
                        runs=np.array([runs]), u=np.array([U]), model=np.array(["uniform[-u,u] per operation result, q90 over runs"]))
    print(f"chain_train: per-parameter band median {np.median(worst):.2e} max {worst.max():.2e}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fomm":
        main_fomm_chain()
    else:
        main()
