"""How much activation-gradient memory each engine program zero-fills before its backward (and how much it may leave uninitialised
because a single BatchNorm backward overwrites it): one training forward+backward of the default workload."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mrfa_amd import engine
from mrfa_amd.train import VOX1, HotPath, l1_loss
from mrfa_amd.utils.prng import det_uniform
orig = engine.Ctx.run_backward
def patched(self):
    todo = [st for st in self.storages if st.grad is None]
    lazy = [st for st in todo if st.grad_noinit or st.data.numel() >= engine.FRESH_MIN_ELEMS]
    arena = [st for st in todo if st not in lazy]
    r = orig(self)
    untouched = [st for st in lazy if st.fresh]                   # never written: dead branches, skipped by their producers
    print(f"program: {len(todo)} gradient buffers, {sum(st.data.numel() for st in todo) * 4 / 2**20:.0f} MiB; zero arena {len(arena)} buffers "
          f"{sum(st.data.numel() for st in arena) * 4 / 2**20:.0f} MiB; allocated uninitialised {len(lazy)} buffers "
          f"{sum(st.data.numel() for st in lazy) * 4 / 2**20:.0f} MiB, of which never written {sum(st.data.numel() for st in untouched) * 4 / 2**20:.0f} MiB")
    return r
engine.Ctx.run_backward = patched
dev = torch.device("cuda", 0)
model = HotPath(VOX1, prior="mtia"); bench.init_weights(model); model.to(dev).train(True)
src = det_uniform("a/s", (8, 3, 256, 256), 0, 1).to(dev); drv = det_uniform("a/d", (8, 3, 256, 256), 0, 1).to(dev)
l1_loss(model(src, drv), drv).backward()
torch.cuda.synchronize()
