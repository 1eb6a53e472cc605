"""Which Python call sites issue ATen / rocBLAS device kernels during ONE eager training step at the bench configuration (torch.profiler with
stacks): the torch arithmetic still on the product path.   python tools/trace_aten.py [mtia|fomm]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step  # noqa: E402
from mrfa_amd.utils.prng import det_uniform  # noqa: E402

prior = sys.argv[1] if len(sys.argv) > 1 else "mtia"
dev = torch.device("cuda:0")
model = HotPath(VOX1, prior=prior)
bench.init_weights(model)
model.to(dev).train(True)
opt = make_optimizer(model, fused=True)
src, drv = det_uniform("t/s", (8, 3, 256, 256), 0, 1).to(dev), det_uniform("t/d", (8, 3, 256, 256), 0, 1).to(dev)
for _ in range(2):
    train_step(model, opt, src, drv)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(model, opt, src, drv)
    torch.cuda.synchronize()
sites = collections.Counter()
tsum = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = getattr(ev, "self_cuda_time_total", 0)
    if not dt:
        continue                                           # (no device kernel of its own)
    own = [s_ for s_ in (ev.stack or []) if "/mrfa_amd/" in s_ or "/repo/bench.py" in s_][:2]
    key = (ev.name, " < ".join(s_.split("/repo/")[-1].split(":")[0] + ":" + s_.split("(")[-1].split(")")[0] + "@" + s_.split("/repo/")[-1].split("(")[0].split(",")[0] for s_ in own) or "(no mrfa_amd frame)")
    sites[key] += 1
    tsum[key] += dt
for key, n in sites.most_common(70):
    print(f"{n:5d} {tsum[key]:9.0f} us  {key[0]:26s} {key[1]}")
print(sum(sites.values()), "ATen ops with device time,", sum(tsum.values()) / 1e3, "ms of device time")
