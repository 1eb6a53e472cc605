"""Signed per-sub-network gradient-norm deviation of the hipGraph-replayed MTIA training step (B = 2, train-mode BatchNorm: the program of
tests/test_headline.py) from the reference's fp32 golden AND from its fp64 run, for the batched encoder pass and for separate encoder calls:
separates a systematic offset (same sign / size in every replay) from summation-order noise.    python tools/headline_probe.py [replays=4] [detail]
`detail` = substring of parameter names whose own signed deviations are printed as well (e.g. kp_head: weight and bias apart).  MRFA_MFMA=f32 / bf16x6 /
bf16x3 selects the matrix pipe: independent roundings of the same program -- an offset that keeps its sign across them is not rounding."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases, headline_checks as H  # noqa: E402
from mrfa_amd.graph import GraphedTrainStep  # noqa: E402
from mrfa_amd.train import VOX1, HotPath, make_optimizer  # noqa: E402

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 4
detail = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
g, names = H.load_golden(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
for batched in (True, False, True, False):
    model = HotPath(VOX1, prior="mtia")
    sds = cases.mtia_chain_weights(model.encoder.state_dict(), model.dense_motion.state_dict(), model.decoder.state_dict())
    for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
        mod.load_state_dict(sds[pfx])
    model.to(dev).train(True)
    model.batched_encoder = batched
    src, drv = cases.images("g12/src_train", 2, 256), cases.images("g12/drv_train", 2, 256)
    opt = make_optimizer(model, fused=True)
    step = GraphedTrainStep(model, opt, src.to(dev), drv.to(dev), clip=10.0, world=1)
    P = {pfx + n: p for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)) for n, p in mod.named_parameters()}
    ref_n, tru_n = g["train_pgrad_norms"], g["train_pgrad_norms_fp64"]
    groups = sorted({H.subnet(n) for n in names})
    rows = {grp: [] for grp in groups}
    per_name = {n: [] for n in names if detail and detail in n}
    losses = []
    for k in range(replays):
        step.g_fb.replay()
        torch.cuda.synchronize()
        got = H.norms_of({n: p.grad for n, p in P.items()}, names)
        losses.append(float(step.loss))
        for i, n in enumerate(names):
            if n in per_name:
                per_name[n].append((got[i] / ref_n[i] - 1.0, got[i] / tru_n[i] - 1.0))
        for grp in groups:
            idx = [i for i, n in enumerate(names) if H.subnet(n) == grp]
            na = np.sqrt((got[idx] ** 2).sum())
            rows[grp].append((na / np.sqrt((ref_n[idx] ** 2).sum()) - 1.0, na / np.sqrt((tru_n[idx] ** 2).sum()) - 1.0))
    print(f"batched_encoder={batched}: loss " + " ".join(f"{x:.7f}" for x in losses) + f" (reference {float(g['train_loss'][0]):.7f}, fp64 {float(g['train_loss_fp64'][0]):.7f})")
    for grp in groups:
        print(f"  {grp:28s} |g|/|g_ref|-1: " + " ".join(f"{a:+.2e}" for a, _ in rows[grp]) + "   vs fp64: " + " ".join(f"{b:+.2e}" for _, b in rows[grp]))
    for n, v in per_name.items():
        print(f"    {n:40s} |g|/|g_ref|-1: " + " ".join(f"{a:+.2e}" for a, _ in v) + "   vs fp64: " + " ".join(f"{b:+.2e}" for _, b in v))
    del step, opt, model
    torch.cuda.empty_cache()
