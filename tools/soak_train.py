"""Stability soak of the default training path: N hipGraph steps on fresh synthetic pairs each step; prints loss trajectory,
memory before / after and checks that nothing turns non-finite."""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd.graph import GraphedTrainStep  # noqa: E402
from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step  # noqa: E402
from mrfa_amd.utils.prng import fill_state_dict  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prior = sys.argv[2] if len(sys.argv) > 2 else "mtia"
dev = torch.device("cuda", 0)
import bench  # noqa: E402
model = HotPath(VOX1, prior=prior)
bench.init_weights(model)
model.to(dev).train()
opt = make_optimizer(model, fused=True)
g = torch.Generator(device=dev).manual_seed(0)
base = torch.rand(8, 3, 256, 256, device=dev, generator=g)
src = base.clone()
drv = base.roll(shifts=(3, 5), dims=(2, 3))
step = GraphedTrainStep(model, opt, src, drv)      # captured and verified at the INITIAL weights (FlatAdam needs no step before): after one Adam step of a
step.verify()                                      # randomly initialised model two eager passes differ by tens of percent and the check compares noise with noise
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
losses = []
t0 = time.perf_counter()
for i in range(steps):
    # a learnable toy task: driving = source shifted by a few pixels (new random source every step)
    s = torch.rand(8, 3, 256, 256, device=dev, generator=g)
    s = torch.nn.functional.avg_pool2d(s, 9, 1, 4)          # smooth images
    d = s.roll(shifts=(3, 5), dims=(2, 3))
    losses.append(step(s, d).clone())      # step() returns the graph's static loss buffer
    if i % 25 == 24:
        print(f"step {i + 1}: loss {float(losses[-1]):.5f}", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ls = [float(x) for x in losses]
ok = all(l == l and abs(l) < 1e6 for l in ls) and all(torch.isfinite(p).all() for p in model.parameters())
print(f"{steps} steps in {dt:.1f} s ({8 * steps / dt:.1f} pairs/s incl. input generation); loss {ls[0]:.4f} -> {ls[-1]:.4f}; "
      f"memory {m0 / 2**30:.2f} -> {torch.cuda.memory_allocated() / 2**30:.2f} GiB; finite: {ok}")
sys.exit(0 if ok and ls[-1] < ls[0] else 1)
