"""Python call sites of zero fills (torch.zeros / zeros_like / Tensor.zero_ / fill_) during ONE eager training step on the GPU, with
element counts: the source of the ~255 ATen fill launches per step.   python tools/trace_fills.py [mtia|fomm]"""
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
sites, elems = collections.Counter(), collections.Counter()


def wrap(mod, name, numel):
    orig = getattr(mod, name)

    def f(*a, **k):
        fr, chain = sys._getframe(1), []
        while fr is not None and len(chain) < 3:
            fn = fr.f_code.co_filename
            if "/repo/" in fn:
                chain.append(f"{fn.split('/repo/')[-1]}:{fr.f_lineno}")
            fr = fr.f_back
        r = orig(*a, **k)
        key = (name, " < ".join(chain))
        sites[key] += 1
        elems[key] += numel(a, r)
        return r
    setattr(mod, name, f)


for n in ("zeros", "zeros_like", "full", "ones"):
    wrap(torch, n, lambda a, r: r.numel())
for n in ("zero_", "fill_"):
    wrap(torch.Tensor, n, lambda a, r: a[0].numel())
train_step(model, opt, src, drv)
torch.cuda.synchronize()
for key, c in sites.most_common(45):
    print(f"{c:4d} x {elems[key] / max(c, 1) / 1e3:10.1f} K elements  {key[0]:10s} {key[1]}")
print(sum(sites.values()), "fills")
