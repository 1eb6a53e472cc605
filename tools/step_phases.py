"""Phase times of the hipGraph-REPLAYED training step, measured by timestamp kernels captured into the graph (engine.Marks, mrfa_timestamp):
no profiler, so the graph's parallel branches overlap exactly as they do in bench.py.

    python tools/step_phases.py [B=8] [prior=mtia] [replays=20]
prints, per mark, the mean time since the first mark of the step (us) and the step's wall time.
"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd import engine  # noqa: E402
from mrfa_amd.graph import GraphedTrainStep  # noqa: E402
from mrfa_amd.train import HotPath, VOX1, make_optimizer  # noqa: E402
from mrfa_amd.utils.prng import det_uniform  # noqa: E402
from bench import init_weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prior = sys.argv[2] if len(sys.argv) > 2 else "mtia"
replays = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
model = HotPath(VOX1, prior=prior)
init_weights(model)
model = model.to(dev).train(True)
opt = make_optimizer(model, fused=True)
src = det_uniform("src", (B, 3, 256, 256), 0, 1).to(dev)
drv = det_uniform("drv", (B, 3, 256, 256), 0, 1).to(dev)
engine.MARKS = engine.Marks(dev)
g = GraphedTrainStep(model, opt, src, drv, clip=VOX1["train_params"]["clip"], world=1)
names = list(engine.MARKS.names)
for _ in range(3):
    g(src, drv)
torch.cuda.synchronize()
acc = None
t0 = time.perf_counter()
for _ in range(replays):
    g(src, drv)
    torch.cuda.synchronize()
    engine.MARKS.names = names
    r = engine.MARKS.read()
    acc = [a + v for a, (_, v) in zip(acc, r)] if acc else [v for _, v in r]
wall = (time.perf_counter() - t0) / replays * 1e3
t0 = time.perf_counter()
for _ in range(replays):
    g(src, drv)
torch.cuda.synchronize()
print(f"B={B} prior={prior}: {(time.perf_counter() - t0) / replays * 1e3:.2f} ms per replayed step ({wall:.2f} with a sync + read-back per step)")
rows = sorted(zip(names, [a / replays for a in acc]), key=lambda x: x[1])
for n, v in rows:
    print(f"{v / 1e3:9.3f} ms  {n}")
