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
    noinit = sum(st.data.numel() for st in self.storages if st.grad is None and st.grad_noinit)
    zero = sum(st.data.numel() for st in self.storages if st.grad is None and not st.grad_noinit)
    big = sorted((st.data.numel() for st in self.storages if st.grad is None and not st.grad_noinit), reverse=True)[:5]
    print(f"program: zero-filled {zero*4/2**20:8.1f} MiB in {sum(1 for st in self.storages if st.grad is None and not st.grad_noinit)} storages, "
          f"uninitialised {noinit*4/2**20:8.1f} MiB; biggest zeroed: {[round(b*4/2**20) for b in big]} MiB")
    return orig(self)
engine.Ctx.run_backward = patched
dev = torch.device("cuda", 0)
model = HotPath(VOX1, prior="mtia"); bench.init_weights(model); model.to(dev).train(True)
src = det_uniform("a/s", (8, 3, 256, 256), 0, 1).to(dev); drv = det_uniform("a/d", (8, 3, 256, 256), 0, 1).to(dev)
l1_loss(model(src, drv), drv).backward()
torch.cuda.synchronize()
