"""Per-launch table of every MFMA conv / dgrad / wgrad / GEMM launch of ONE training step at the bench configuration
(hipEvent-timed through the engine's profiling hook): shape, tile config, ms, TFLOP/s.   python tools/profile_step.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd.engine import Ctx  # noqa: E402
from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step  # noqa: E402
from mrfa_amd.utils.prng import det_uniform, fill_state_dict  # noqa: E402

dev = torch.device("cuda:0")
PRIOR = sys.argv[2] if len(sys.argv) > 2 else "fomm"          # python tools/profile_step.py 8 mtia [rows]
model = HotPath(VOX1, prior=PRIOR)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
bench.init_weights(model)
model.to(dev).train(True)
opt = make_optimizer(model)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
src = det_uniform("p/src", (B, 3, 256, 256), 0, 1).to(dev)
drv = det_uniform("p/drv", (B, 3, 256, 256), 0, 1).to(dev)
for _ in range(2):
    train_step(model, opt, src, drv)
torch.cuda.synchronize()
Ctx.profile = []
train_step(model, opt, src, drv)
torch.cuda.synchronize()
prof, Ctx.profile = Ctx.profile, None
rows = [(e0.elapsed_time(e1), f, cfg, d) for cfg, f, e0, e1, d in prof]
tot = sum(r[0] for r in rows)
print(f"{len(rows)} MFMA launches, {tot:.1f} ms, {sum(r[1] for r in rows) / tot / 1e9:.1f} TF/s overall")
agg = {}
for ms, f, cfg, d in rows:
    a = agg.setdefault(d, [0.0, 0.0, 0, cfg])
    a[0] += ms; a[1] += f; a[2] += 1
NROWS = int(sys.argv[3]) if len(sys.argv) > 3 else 70
for d, (ms, f, n, cfg) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:NROWS]:
    tile = "wgrad" if cfg < 0 else ("halo " if cfg & (1 << 28) else "") + f"{(cfg >> 16) & 0xfff}x{(cfg >> 4) & 0xfff}{'f' if cfg & 2 else ''}{'s' if cfg & 1 else ''}{'b' if cfg & 4 else ''}"
    print(f"{ms:7.3f} ms  x{n:<2d} {f / ms / 1e9:6.1f} TF/s  {tile:10s} {d}")
