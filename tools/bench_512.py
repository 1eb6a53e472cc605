"""BASELINE config 5: vox1 512x512 inference-only generator path, bs=4, one MI355X -- the HBM-bound grid_sample stress.
  (a) RaftFlow forward at size=512, B=4 (eval, hipGraph replay): ms and pairs/s;
  (b) the six-level feature warps on their own: every grid_sample of a warp set timed with HIP events, GB/s of ALGORITHMIC
      bytes 4*(C*Hi*Wi + C*Ho*Wo + 2*Ho*Wo) per sample against the 8 TB/s HBM peak.
      python tools/bench_512.py [B]"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd.engine import Ctx  # noqa: E402
from mrfa_amd.modules import RaftFlow  # noqa: E402
from mrfa_amd.train import VOX1  # noqa: E402
from mrfa_amd.utils.prng import det_uniform, fill_state_dict  # noqa: E402
import copy  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = 512
dev = torch.device("cuda", 0)
cfg = copy.deepcopy(VOX1["raft_flow"])
cfg["size"] = size
rf = RaftFlow(**cfg)
sd = fill_state_dict(rf.state_dict(), tag="decoder.")
for k in list(sd):
    if k.endswith(("refine.conv2.weight", "refine.convo2.weight")):
        sd[k] = sd[k] * 0.3
rf.load_state_dict(sd)
rf.to(dev).eval()
h = size // 4
img_full = det_uniform("c5/img", (B, 3, size, size), 0, 1).to(dev)
img = torch.nn.functional.avg_pool2d(img_full, 4)
kp_s = det_uniform("c5/ks", (B, 10, 2), -0.8, 0.8).to(dev)
kp_d = det_uniform("c5/kd", (B, 10, 2), -0.8, 0.8).to(dev)
ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, h), indexing="ij")
deform = (torch.stack([xs, ys], dim=-1)[None].expand(B, h, h, 2) + det_uniform("c5/d", (B, h, h, 2), -0.1, 0.1)).contiguous().to(dev)
occ = det_uniform("c5/o", (B, 1, h, h), -2, 2).to(dev)
dm = {"deformation": deform, "occlusion": occ}


def fwd():
    with torch.no_grad():
        return rf(kp_s, kp_d, dm, img, img_full)[0]


out = fwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    g.replay()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
gf = (4 * (362.73 - 8.59) + 137.4) * B                     # 4x the conv FLOPs of RaftFlow at 256^2 + the 16x correlation GEMM
print(f"RaftFlow forward 512x512 B={B}: {ms:.2f} ms = {B / ms * 1e3:.1f} pairs/s  ({gf / ms:.1f} TFLOP/s algorithmic = "
      f"{gf / ms / 157.3:.2f} of the fp32 MFMA peak), "
      f"memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, out finite: {bool(torch.isfinite(out).all())}")

# (b) one six-level warp set (raft.py:247/260/271), both sampling conventions
ectx = Ctx(dev, train=False, record=False)
tot_b = tot_ms = 0.0
for C_, r in ((512, 16), (512, 32), (512, 64), (256, 128), (128, 256), (64, 512)):
    f = ectx.new(B, r, r, C_)
    f.st.data.normal_()
    for mode, name in ((1, "flow px / align_corners=True"), (0, "normalised / align_corners=False")):
        grid = ectx.new(B, r, r, 2)
        grid.st.data.uniform_(-3, 3) if mode == 1 else grid.st.data.uniform_(-1.05, 1.05)
        o = ectx.new(B, r, r, C_)
        for _ in range(3):
            ectx.grid_sample(f, grid, mode, out=o)
        torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            ectx.grid_sample(f, grid, mode, out=o)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20
        byt = 4.0 * B * (2 * C_ * r * r + 2 * r * r)
        if mode == 1:
            tot_b, tot_ms = tot_b + byt, tot_ms + t
        print(f"  grid_sample C={C_:3d} @{r:3d}^2 ({name}): {t * 1e3:7.1f} us  {byt / t / 1e6:6.0f} GB/s = {byt / t / 1e6 / 8000:.2f} of HBM peak")
print(f"  one warp set (6 levels, B={B}): {tot_b / 1e9:.2f} GB algorithmic in {tot_ms:.3f} ms = {tot_b / tot_ms / 1e6:.0f} GB/s "
      f"({tot_b / tot_ms / 1e6 / 8000:.2f} of the 8 TB/s HBM peak)")
