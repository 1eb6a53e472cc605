"""The headline program -- the MTIA-prior training step of BASELINE config 2: TokenPose_B on source and driving, DenseMotionNetwork,
RaftFlow, loss = mean|gen - driving|, backward -- against the reference's OWN MRFA (tests/golden/chain_mtia.npz,
tools/make_goldens.py:g12_chain_mtia; model.py:185-210, train.py:58-64) and against the oracle's autograd.

What is compared, per sub-network (first two components of the parameter names: `encoder.pre_feature`, `encoder.transformer`,
`dense_motion.hourglass`, `decoder.generator`, `decoder.refine`, ...):
  * the direction of the gradient vector: 1 - cosine,
  * its length: relative difference of the L2 norm,
against (a) the reference's gradients -- the golden keeps every parameter's gradient norm and <= 64 evenly spread entries of every
parameter's gradient -- and (b) the oracle's autograd gradients, whole vectors.  The allowance of a sub-network is a constant plus
a multiple of the distance of the fp32 REFERENCE itself from an fp64 run of the same program (stored next to the golden): a result
cannot be held closer to the reference than the reference is to exact arithmetic.  That distance is taken from TWO fp32 realisations
of the unmodified reference (oneDNN and ATen-native convolution kernels: `*_alt`), the larger one per sub-network: one run is one
sample of the rounding noise, and the two differ by up to 10x in single sub-networks (train-mode BatchNorm over 8 values at the
bottom of the hourglasses).  No term of the allowance is measured from the implementation under test (no run-to-run widening).  Sub-networks whose gradient is tiny next to their network's (`dense_motion.mask`:
0.3 % of dense motion's gradient norm, a sum of cancelling terms) are measured relative to NET_FLOOR = 1 % of the norm of their top-level
network (encoder / dense_motion / decoder) instead of their own: an absolute error that is invisible in the network's update is not
held against a near-zero denominator."""
import json
import os

import numpy as np
import torch

from tests import cases

# constants of the gates (see check_against_reference): cosine distance, relative norm, d loss / d keypoints.
# Round 5: the norm multiple is back at 3 (round 4 had raised it to 6 after single-replay failures; the cosine multiple stays 6 = 2.4 x the error, 1 - cos
# being quadratic in it) because the BAND they multiply is now the right one.  Until round 4 it was the distance of the reference's two CPU realisations (oneDNN / ATen-native convolutions) from its fp64 run -- but those two
# share most of their rounding: seven CPU realisations (threads 1 / 3 / 8, both backends, samples flipped) put `decoder.to_context`'s gradient norm within
# -1.0e-3 ... +0.7e-3 of the fp64 run, while the Monte-Carlo-arithmetic band of the same program (tools/mca_band.py: the fp64 run with every operation
# result perturbed by ONE fp32 unit roundoff, forward and backward, 8 runs) scatters it by 0.5e-3 ... 4.7e-3, `decoder.kp_head` by up to 2.1e-2: at
# B = 2 in train mode the program amplifies rounding 10^3-10^4-fold, and ANY independent fp32 implementation lands somewhere in that scatter (the HIP
# path: to_context +1.3e-3 ... +8.3e-3 over 20 replays, kp_head -0.6e-2 ... -3.2e-2, stable offsets per kernel selection plus ~1e-3 of atomic-order
# noise).  reference_band() takes the larger of the two per sub-network; neither contains a term measured from the implementation under test.
COS_BASE, COS_MULT = 2e-5, 6.0
NORM_BASE, NORM_MULT = 1e-3, 3.0
DKP_BASE, DKP_MULT = 1e-3, 3.0
NET_FLOOR = 1e-2


def _floor_factor(names, norms_ref):
    """{sub-network: f >= 1}: f = NET_FLOOR * |gradient of its top-level network| / |its own gradient| where that exceeds 1"""
    net = {}
    for i, n in enumerate(names):
        net[n.split(".")[0]] = net.get(n.split(".")[0], 0.0) + float(norms_ref[i]) ** 2
    out = {}
    for grp in {subnet(n) for n in names}:
        own = np.sqrt(sum(float(norms_ref[i]) ** 2 for i, n in enumerate(names) if subnet(n) == grp))
        out[grp] = max(1.0, NET_FLOOR * np.sqrt(net[grp.split(".")[0]]) / max(own, 1e-300))
    return out


def load_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "chain_mtia.npz"))
    names = json.load(open(os.path.join(golden_dir, "chain_mtia_param_names.json")))
    return g, names


def subnet(name: str) -> str:
    return ".".join(name.split(".")[:2])


def sample_segments(names, numels):
    """name -> [lo, hi) of its entries inside the golden's concatenated `*_pgrad_samples` vector"""
    off, segs = 0, {}
    for n in names:
        k = len(cases.sample_index(numels[n]))
        segs[n] = (off, off + k)
        off += k
    return segs, off


def sampled(grads: dict, names, numels) -> np.ndarray:
    """the golden's sample vector taken from {name: gradient tensor or None}"""
    parts = []
    for n in names:
        idx = cases.sample_index(numels[n])
        g = grads.get(n)
        parts.append(np.zeros(len(idx), np.float64) if g is None else g.detach().reshape(-1)[idx.to(g.device)].double().cpu().numpy())
    return np.concatenate(parts)


def norms_of(grads: dict, names) -> np.ndarray:
    return np.array([0.0 if grads.get(n) is None else float(grads[n].detach().double().norm()) for n in names])


def _cosdist(x, y):
    return 1.0 - float(np.dot(x, y) / max(np.linalg.norm(x) * np.linalg.norm(y), 1e-300))


def group_table(names, segs, samples_a, norms_a, samples_b, norms_b):
    """per sub-network: (1 - cos of the sampled entries, relative difference of the whole-vector norms, norm of b)"""
    out = {}
    for grp in sorted({subnet(n) for n in names}):
        idx = [i for i, n in enumerate(names) if subnet(n) == grp]
        sel = np.concatenate([np.arange(*segs[names[i]]) for i in idx])
        na, nb = np.sqrt((norms_a[idx] ** 2).sum()), np.sqrt((norms_b[idx] ** 2).sum())
        out[grp] = (_cosdist(samples_a[sel], samples_b[sel]), abs(na - nb) / max(nb, 1e-300), nb)
    return out


def check_against_reference(g, names, sfx, loss, gen, kps, dkps, grads, numels, what, loss_tol=2e-5):
    """`grads`: {parameter name: gradient}; `kps` / `dkps`: {"kp_s", "jac_s", "kp_d", "jac_d"} -> tensors (values / d loss / d value).
    Returns the worst (measured / allowed) ratio over all gates, for the record."""
    segs, total = sample_segments(names, numels)
    assert total == len(g[f"{sfx}_pgrad_samples"])
    ref_loss, truth_loss = float(g[f"{sfx}_loss"][0]), float(g[f"{sfx}_loss_fp64"][0])
    assert abs(loss - ref_loss) <= loss_tol + 3.0 * abs(ref_loss - truth_loss), (what, loss, ref_loss)
    gerr = np.abs(gen.detach().float().cpu().numpy()[:, :, ::4, ::4] - g[f"{sfx}_gen_s4"])
    nz = _noise(g, f"{sfx}_gen_s4", f"{sfx}_gen_s4_fp64")
    assert gerr.mean() <= 1e-4 + 3.0 * nz.mean() and gerr.max() <= 5e-3 + 3.0 * nz.max(), (what, gerr.mean(), gerr.max(), nz.mean(), nz.max())
    worst = 0.0
    for key in ("kp_s", "jac_s", "kp_d", "jac_d"):
        if kps is not None:
            e = np.abs(kps[key].detach().cpu().numpy() - g[f"{sfx}_{key}"]).max()
            assert e <= 1e-4, (what, key, e)                          # keypoints <= 1e-4
        ref, truth = g[f"{sfx}_d{key}"], g[f"{sfx}_d{key}_fp64"]
        e = np.abs(dkps[key].detach().cpu().numpy() - ref).max()
        allow = DKP_BASE * np.abs(ref).max() + DKP_MULT * _noise(g, f"{sfx}_d{key}", f"{sfx}_d{key}_fp64").max()
        worst = max(worst, e / allow)
        assert e <= allow, (what, "d" + key, e, allow)
    got_s, got_n = sampled(grads, names, numels), norms_of(grads, names)
    ref_s, ref_n = g[f"{sfx}_pgrad_samples"].astype(np.float64), g[f"{sfx}_pgrad_norms"]
    tru_s, tru_n = g[f"{sfx}_pgrad_samples_fp64"], g[f"{sfx}_pgrad_norms_fp64"]
    band = reference_band(g, names, numels, sfx)                           # the reference's own distance from exact arithmetic
    mine = group_table(names, segs, got_s, got_n, ref_s, ref_n)
    ff = _floor_factor(names, ref_n)
    lines, fails = [], []
    for grp, (cd, nr, nb) in mine.items():
        a_c, a_n = (COS_BASE + COS_MULT * band[grp][0]) * ff[grp] ** 2, (NORM_BASE + NORM_MULT * band[grp][1]) * ff[grp]
        lines.append(f"  {grp:30s} 1-cos {cd:.2e} (allow {a_c:.2e})  |g| rel {nr:.2e} (allow {a_n:.2e})  |g_ref| {nb:.3e}")
        worst = max(worst, cd / a_c, nr / a_n)
        if not (cd <= a_c and nr <= a_n):
            fails.append((grp, cd, a_c, nr, a_n))
    print(f"{what} vs the reference's MRFA ({sfx}): loss {loss:.7f} (reference {ref_loss:.7f})\n" + "\n".join(lines))
    assert not fails, (what, sfx, fails)
    return worst


def oracle_run(sds: dict, src, drv, train: bool, threads: int = 0, native_convs: bool = False):
    """the oracle's forward + autograd backward of the same program on the host -> (loss, gen, kps, dkps, {name: grad}, buffers).
    native_convs: ATen's own convolution kernels instead of oneDNN's -- a second fp32 realisation (another summation order)"""
    if native_convs:
        with torch.backends.mkldnn.flags(enabled=False):
            return oracle_run(sds, src, drv, train, threads)
    from oracle import mrfa_oracle as O
    if threads:
        torch.set_num_threads(threads)
    P = {}
    for pfx, sd in sds.items():
        for k, v in sd.items():
            frozen = k.endswith(("running_mean", "running_var", "num_batches_tracked", "down.weight")) or (pfx == "encoder." and k.endswith("pos_embedding"))
            P[pfx + k] = v.clone().requires_grad_(True) if (v.is_floating_point() and not frozen) else v.clone()
    gen, _, ks, kd, _ = O.mrfa_forward(src, drv, P, size=src.shape[-1], train=train, prior="mtia")
    for d in (ks, kd):
        for v in d.values():
            v.retain_grad()
    loss = (gen - drv).abs().mean()
    loss.backward()
    kps = {"kp_s": ks["kp"], "jac_s": ks["jacobian"], "kp_d": kd["kp"], "jac_d": kd["jacobian"]}
    dkps = {k: v.grad for k, v in kps.items()}
    grads = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in P.items() if p.is_floating_point()}
    return float(loss.detach()), gen.detach(), {k: v.detach() for k, v in kps.items()}, dkps, grads, P


def check_against_oracle(names, grads, ograds, what, band=None):
    """whole gradient vectors, per sub-network, against the oracle's autograd: 1 - cos and relative norm.  `band`: {sub-network:
    (cos distance, relative norm)} of the fp32 reference from fp64 (group_table on the golden) -- the same allowance as above."""
    lines, worst, fails = [], 0.0, []
    ff = _floor_factor(names, np.array([float(ograds[n].detach().double().norm()) for n in names]))
    for grp in sorted({subnet(n) for n in names}):
        ns = [n for n in names if subnet(n) == grp]
        a = torch.cat([(grads[n] if grads.get(n) is not None else torch.zeros_like(ograds[n])).detach().reshape(-1).double().cpu() for n in ns])
        b = torch.cat([ograds[n].detach().reshape(-1).double() for n in ns])
        cd = 1.0 - float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
        nr = abs(float(a.norm()) - float(b.norm())) / max(float(b.norm()), 1e-300)
        a_c = (COS_BASE + COS_MULT * (band[grp][0] if band else 0.0)) * ff[grp] ** 2
        a_n = (NORM_BASE + NORM_MULT * (band[grp][1] if band else 0.0)) * ff[grp]
        lines.append(f"  {grp:30s} 1-cos {cd:.2e} (allow {a_c:.2e})  |g| rel {nr:.2e} (allow {a_n:.2e})  elements {a.numel()}")
        worst = max(worst, cd / a_c, nr / a_n)
        if not (cd <= a_c and nr <= a_n):
            fails.append((grp, cd, a_c, nr, a_n))
    print(f"{what} vs the oracle's autograd (whole vectors):\n" + "\n".join(lines))
    assert not fails, (what, fails)
    return worst


def whole_vector_band(names, grads_a, grads_b):
    """{sub-network: (1 - cos, relative norm difference)} between two fp32 realisations of the ORACLE (whole gradient vectors): the sampled
    entries of the golden under-estimate the rounding noise of vectors with 10^7 entries (decoder.generator: 3e-6 on the samples, 7e-5 on the
    whole vector), so the whole-vector comparison gets its own, directly measured band"""
    out = {}
    for grp in sorted({subnet(n) for n in names}):
        ns = [n for n in names if subnet(n) == grp]
        a = torch.cat([grads_a[n].detach().reshape(-1).double() for n in ns])
        b = torch.cat([grads_b[n].detach().reshape(-1).double() for n in ns])
        cd = 1.0 - float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
        out[grp] = (cd, abs(float(a.norm()) - float(b.norm())) / max(float(b.norm()), 1e-300), float(b.norm()))
    return out


def merge_bands(a, b):
    return {k: (max(a[k][0], b[k][0]), max(a[k][1], b[k][1]), a[k][2]) for k in a}


def reference_band(g, names, numels, sfx):
    """{sub-network: (1 - cos, relative norm difference)}: how far a correct fp32 evaluation of this program may sit from the fp64 run -- per entry
    the largest of (a) the reference's two fp32 realisations (the golden's main run and `*_alt`) and (b), train mode, the Monte-Carlo-arithmetic
    band of tools/mca_band.py (tests/golden/chain_mtia_mca.npz: one unit roundoff per operation result)"""
    segs, _ = sample_segments(names, numels)
    tru = (g[f"{sfx}_pgrad_samples_fp64"], g[f"{sfx}_pgrad_norms_fp64"])
    band = group_table(names, segs, g[f"{sfx}_pgrad_samples"].astype(np.float64), g[f"{sfx}_pgrad_norms"], *tru)
    if f"{sfx}_pgrad_samples_alt" in g.files:
        alt = group_table(names, segs, g[f"{sfx}_pgrad_samples_alt"].astype(np.float64), g[f"{sfx}_pgrad_norms_alt"], *tru)
        band = {k: (max(band[k][0], alt[k][0]), max(band[k][1], alt[k][1]), band[k][2]) for k in band}
    mca_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_mtia_mca.npz")
    if sfx == "train" and os.path.exists(mca_path):
        mca = np.load(mca_path)
        groups = json.load(open(mca_path.replace(".npz", "_groups.json")))
        for i, grp in enumerate(groups):
            if grp in band:
                band[grp] = (max(band[grp][0], float(mca["train_cos"][i])), max(band[grp][1], float(mca["train_norm"][i])), band[grp][2])
    return band


def _noise(g, key_ref, key_truth):
    """max |reference - fp64| of a stored tensor, over the reference's fp32 realisations"""
    truth = g[key_truth]
    n = np.abs(g[key_ref] - truth)
    if key_ref + "_alt" in g.files:
        n = np.maximum(n, np.abs(g[key_ref + "_alt"] - truth))
    return n
