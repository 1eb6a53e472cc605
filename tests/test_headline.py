"""The headline program of the benchmark -- the hipGraph-REPLAYED MTIA-prior training step (BASELINE config 2: ONE batched TokenPose_B pass over source + driving (per-call BatchNorm statistics),
DenseMotion, RaftFlow, surrogate L1 loss, backward with deferred weight gradients) -- held to the reference's own
`MRFA(prior_model='mtia')` (tests/golden/chain_mtia.npz, tools/make_goldens.py:g12_chain_mtia: model.py:185-210, train.py:58-64) and to
the oracle's autograd.  VERDICT r3 item 2: what meets the reference here is the replayed graph, not an eager twin of it.

CPU leg: the oracle itself against that golden (pins the oracle's backward of the MTIA chain)."""
import numpy as np
import pytest
import torch

from tests import cases
from tests import headline_checks as H


def _numels(sds):
    return {pfx + k: v.numel() for pfx, sd in sds.items() for k, v in sd.items()}


def _sds():
    from mrfa_amd.train import VOX1, HotPath
    m = HotPath(VOX1, prior="mtia")
    return cases.mtia_chain_weights(m.encoder.state_dict(), m.dense_motion.state_dict(), m.decoder.state_dict())


def test_oracle_autograd_of_the_mtia_chain_vs_reference_golden(golden_dir):
    """eval-mode BatchNorm (the well-conditioned leg; the train-mode leg of the oracle runs on the GPU box next to the HIP path)"""
    g, names = H.load_golden(golden_dir)
    sds = _sds()
    src, drv = cases.images("g12/src_eval", 2, 256), cases.images("g12/drv_eval", 2, 256)
    loss, gen, kps, dkps, grads, _ = H.oracle_run(sds, src, drv, train=False)
    H.check_against_reference(g, names, "eval", loss, gen, kps, dkps, grads, _numels(sds), "oracle (fp32, CPU)")


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True], ids=["eval_bn", "train_bn"])
def test_replayed_mtia_training_step_vs_reference_and_oracle_autograd(golden_dir, train):
    """HotPath(VOX1, prior='mtia') from known weights at B=2 -> GraphedTrainStep (FlatAdam, the batched encoder pass with HRNet branch lanes, deferred weight
    gradients: everything the benchmark's step uses) -> ONE replay of graph A -> loss, generated frames, keypoints, d loss / d keypoints,
    per-sub-network gradient vectors (direction + length), BatchNorm running buffers: against the reference's MRFA golden and against the
    oracle's autograd run here on the host.  Then a second replay must reproduce the first inside the same gates (replays are the
    object that is timed)."""
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import VOX1, HotPath, make_optimizer
    dev = torch.device("cuda", 0)
    sfx = "train" if train else "eval"
    g, names = H.load_golden(golden_dir)
    model = HotPath(VOX1, prior="mtia")
    sds = cases.mtia_chain_weights(model.encoder.state_dict(), model.dense_motion.state_dict(), model.decoder.state_dict())
    numels = _numels(sds)
    for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
        mod.load_state_dict(sds[pfx])
    model.to(dev).train(train)
    model.probe = {}
    src, drv = cases.images(f"g12/src_{sfx}", 2, 256), cases.images(f"g12/drv_{sfx}", 2, 256)
    opt = make_optimizer(model, fused=True)
    step = GraphedTrainStep(model, opt, src.to(dev), drv.to(dev), clip=10.0, world=1)
    if train:
        assert model.batched_encoder and model.defer_decoder_wgrads              # the benchmark's schedule, not a simplified one
    P = {pfx + n: p for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder))
         for n, p in mod.named_parameters()}
    before = {n: p.detach().clone() for n, p in P.items()}
    band = H.reference_band(g, names, numels, sfx)
    # the oracle's autograd on the host, same weights and inputs
    o_loss, o_gen, o_kps, o_dkps, o_grads, o_P = H.oracle_run(sds, src, drv, train=train, threads=min(32, torch.get_num_threads()))
    H.check_against_reference(g, names, sfx, o_loss, o_gen, o_kps, o_dkps, o_grads, numels, "oracle (fp32, host)")
    # a second fp32 realisation of the oracle (ATen-native convolutions): its whole-vector distance from the first is the rounding noise of fp32
    # on THESE vectors, measured without the implementation under test
    o2_grads = H.oracle_run(sds, src, drv, train=train, threads=min(32, torch.get_num_threads()), native_convs=True)[4]
    band = H.merge_bands(band, H.whole_vector_band(names, o_grads, o2_grads))

    def replay():
        step.g_fb.replay()                                # graph A only: zero, pack, forward, loss, backward; the weights stay put
        torch.cuda.synchronize()
        grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in P.items()}
        kps = {k: model.probe[k].detach().clone() for k in ("kp_s", "jac_s", "kp_d", "jac_d")}
        dkps = {k: model.probe["d" + k].detach().clone() for k in ("kp_s", "jac_s", "kp_d", "jac_d")}
        return float(step.loss), step.gen.detach().clone(), kps, dkps, grads

    for k in range(2):
        loss, gen, kps, dkps, grads = replay()
        assert all(torch.equal(P[n].detach(), before[n]) for n in P), "graph A changed a weight"
        w1 = H.check_against_reference(g, names, sfx, loss, gen, kps, dkps, grads, numels, f"hipGraph replay {k}")
        w2 = H.check_against_oracle(names, grads, {n: o_grads[n] for n in names}, f"hipGraph replay {k} ({sfx})", band)
        # output and keypoint gradients against the oracle run too (whole tensors)
        assert abs(loss - o_loss) <= 2e-5
        assert float((gen.cpu() - o_gen).abs().mean()) <= 1e-4 + 3.0 * float(H._noise(g, f"{sfx}_gen_s4", f"{sfx}_gen_s4_fp64").mean())
        print(f"replay {k} ({sfx}): worst measured / allowed = {max(w1, w2):.2f}")
        if train and k == 0:
            # one replayed step = one source pass then one driving pass of BatchNorm running-statistic updates (model.py:185-186 order)
            bufs = {pfx + n: v for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder))
                    for n, v in mod.named_buffers()}
            for key in [f for f in g.files if f.startswith("train_buf_")]:
                n = key[len("train_buf_"):]
                e = float((bufs[n].cpu() - torch.from_numpy(g[key])).abs().max())
                assert e <= 1e-5 * max(1.0, float(np.abs(g[key]).max())), (n, e)
            # the oracle's encoder buffers agree as well (its decoder restatement does not update running statistics)
            for n in ("encoder.pre_feature.bn1.running_mean",):
                assert float((bufs[n].cpu() - o_P[n]).abs().max()) <= 1e-5
    # the full step (graph A + clip + Adam) then moves every weight that has a gradient
    step(src.to(dev), drv.to(dev))
    torch.cuda.synchronize()
    moved = sum(int(not torch.equal(P[n].detach(), before[n])) for n in names if o_grads[n].abs().max() > 0)
    assert moved >= 0.99 * sum(int(o_grads[n].abs().max() > 0) for n in names)


@pytest.mark.gpu
def test_replayed_step_at_the_bench_batch_vs_oracle_autograd(golden_dir):
    """the headline parity ran at B = 2, the benchmark runs B = 8.  The same program at the BENCH batch -- HotPath(VOX1, prior='mtia'),
    8 source + 8 driving frames (one batched encoder pass of 16 with two statistic groups), train-mode BatchNorm, GraphedTrainStep -- one replay of graph A
    against the oracle's fp32 autograd of that batch on the host: loss, generated frames, keypoints, every sub-network's whole gradient vector (direction and
    length).  The reference itself cannot travel and an fp64 run of this batch needs 54 GB, so there is no golden at B = 8: the band per sub-network is the
    larger of the B = 2 band (reference fp32 vs fp64 + Monte-Carlo arithmetic, tests/golden/chain_mtia*.npz) and the distance between two fp32 realisations of
    the oracle AT B = 8 (oneDNN / ATen-native convolutions), neither of which contains the implementation under test."""
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import VOX1, HotPath, make_optimizer
    dev = torch.device("cuda", 0)
    B = 8
    g, names = H.load_golden(golden_dir)
    model = HotPath(VOX1, prior="mtia")
    sds = cases.mtia_chain_weights(model.encoder.state_dict(), model.dense_motion.state_dict(), model.decoder.state_dict())
    numels = _numels(sds)
    for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
        mod.load_state_dict(sds[pfx])
    model.to(dev).train(True)
    model.probe = {}
    src, drv = cases.images("g13/src_train", B, 256), cases.images("g13/drv_train", B, 256)
    step = GraphedTrainStep(model, make_optimizer(model, fused=True), src.to(dev), drv.to(dev), clip=10.0, world=1)
    assert model.batched_encoder and model.defer_decoder_wgrads
    P = {pfx + n: p for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder))
         for n, p in mod.named_parameters()}
    thr = min(32, torch.get_num_threads())
    o_loss, o_gen, o_kps, _, o_grads, _ = H.oracle_run(sds, src, drv, train=True, threads=thr)
    o2_grads = H.oracle_run(sds, src, drv, train=True, threads=thr, native_convs=True)[4]
    band = H.merge_bands(H.reference_band(g, names, numels, "train"), H.whole_vector_band(names, o_grads, o2_grads))
    for k in range(2):
        step.g_fb.replay()
        torch.cuda.synchronize()
        grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in P.items()}
        worst = H.check_against_oracle(names, grads, {n: o_grads[n] for n in names}, f"hipGraph replay {k} (B = {B}, train)", band)
        assert abs(float(step.loss) - o_loss) <= 2e-5, (float(step.loss), o_loss)
        assert float((step.gen.detach().cpu() - o_gen).abs().mean()) <= 1e-4
        for kk in ("kp_s", "jac_s", "kp_d", "jac_d"):
            e = float((model.probe[kk].detach().cpu() - o_kps[kk]).abs().max())
            assert e <= 2e-4 * max(1.0, float(o_kps[kk].abs().max())), (kk, e)
        print(f"B = {B} replay {k}: loss {float(step.loss):.7f} (oracle {o_loss:.7f}); worst measured / allowed = {worst:.2f}")
