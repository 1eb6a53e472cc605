"""Run-to-run gradient noise of the whole training path at fixed weights and inputs (eager launches): relative L2 distance of
each parameter group's gradient between two identical forward+backward passes.  Atomic accumulation order is the only
non-determinism; the distance measures how strongly the randomly initialised, train-mode-BN model amplifies it."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mrfa_amd.train import VOX1, HotPath, l1_loss  # noqa: E402
from mrfa_amd.utils.prng import det_uniform  # noqa: E402

prior = sys.argv[1] if len(sys.argv) > 1 else "mtia"
dev = torch.device("cuda", 0)
model = HotPath(VOX1, prior=prior)
bench.init_weights(model)
model.to(dev).train(True)
src = det_uniform("bench/src/r0", (8, 3, 256, 256), 0, 1).to(dev)
drv = det_uniform("bench/drv/r0", (8, 3, 256, 256), 0, 1).to(dev)
saved = [b.clone() for b in model.buffers()]
runs = []
for k in range(3):
    for b, sv in zip(model.buffers(), saved):
        b.copy_(sv)
    model.zero_grad(set_to_none=True)
    loss = l1_loss(model(src, drv), drv)
    loss.backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.double().clone() for n, p in model.named_parameters() if p.grad is not None})
    print(f"run {k}: loss {float(loss):.8f}")
for grp in ("encoder", "dense_motion", "decoder"):
    names = [n for n in runs[0] if n.startswith(grp + ".")]
    for k in (1, 2):
        num = sum(float((runs[k][n] - runs[0][n]).pow(2).sum()) for n in names) ** 0.5
        den = sum(float(runs[0][n].pow(2).sum()) for n in names) ** 0.5
        print(f"{grp:13s} run {k} vs 0: rel L2 {num / den:.4f}   |g| {den:.4e}  max|g| {max(float(runs[0][n].abs().max()) for n in names):.3e}")
