"""Which ops still trigger a lazy zero fill of a fresh gradient buffer (first writer that cannot overwrite), by bytes."""
import collections, os, sys, traceback
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mrfa_amd import engine
from mrfa_amd.train import VOX1, HotPath, l1_loss
from mrfa_amd.utils.prng import det_uniform
stats = collections.Counter()
orig = engine.Storage.grad_buf
def patched(self):
    if self.grad is not None and self.fresh:
        fr = [f for f in traceback.extract_stack()[:-1] if "engine.py" in f.filename or "modules" in f.filename or "losses" in f.filename]
        key = " <- ".join(f"{f.name}:{f.lineno}" for f in fr[-3:])
        stats[key] += self.data.numel() * 4
    return orig(self)
engine.Storage.grad_buf = patched
dev = torch.device("cuda", 0)
model = HotPath(VOX1, prior=sys.argv[1] if len(sys.argv) > 1 else "mtia"); bench.init_weights(model); model.to(dev).train(True)
src = det_uniform("a/s", (8, 3, 256, 256), 0, 1).to(dev); drv = det_uniform("a/d", (8, 3, 256, 256), 0, 1).to(dev)
l1_loss(model(src, drv), drv).backward()
torch.cuda.synchronize()
print(f"lazily zero-filled: {sum(stats.values()) / 2**20:.0f} MiB")
for k, v in stats.most_common(14):
    print(f"  {v / 2**20:8.0f} MiB  {k}")
