"""Per-shape MFMA conv / wgrad time of ONE TokenPose_B forward+backward pass (B=8, 256x256), eager launches with a HIP event pair
around every launch (engine.Ctx.profile): which layer shapes the ~30 ms of small convolutions of the MTIA encoder go to.
    python tools/profile_encoder_convs.py [B] [mfma mode]"""
import collections
import copy
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd import hip  # noqa: E402
from mrfa_amd.engine import Ctx  # noqa: E402
from mrfa_amd.modules.transformer import get_pose_net  # noqa: E402
from mrfa_amd.modules.util import convert_dict_to_attrit_dict  # noqa: E402
from mrfa_amd.train import VOX1  # noqa: E402
from mrfa_amd.utils.prng import det_uniform, fill_tokenpose_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:
    hip.set_mfma_mode(sys.argv[2])
net = get_pose_net(convert_dict_to_attrit_dict(copy.deepcopy(VOX1["mtia_kp_detector"])), is_train=True)
net.load_state_dict(fill_tokenpose_state_dict(net.state_dict(), "encoder."))
net.to("cuda:0").train(True)
x = det_uniform("tpb/x", (B, 3, 256, 256), 0, 1).to("cuda:0")


def fb():
    net.zero_grad(set_to_none=True)
    o = net(x)
    (o["kp"].sum() + o["jacobian"].sum()).backward()


for _ in range(2):
    fb()
torch.cuda.synchronize()
Ctx.profile = []
reps = 3
for _ in range(reps):
    fb()
torch.cuda.synchronize()
prof, Ctx.profile = Ctx.profile, None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0])
for cfg, flops, e0, e1, desc in prof:
    key = desc.split(" ldx=")[0] + (f" tile={cfg >> 16}x{(cfg >> 4) & 0xfff}{' splitK' if cfg & 1 else ''}{' bf16x6' if cfg & 4 else ''}" if cfg >= 0 else "")
    a = agg[key]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += flops
tot = sum(a[1] for a in agg.values()) / reps
print(f"mode {hip.mfma_mode()}  B={B}: {len(prof) // reps} MFMA launches per pass, {tot:.2f} ms of launch-to-launch event time")
print(f"{'ms/pass':>8} {'n':>4} {'us each':>8} {'TF/s':>6}  shape")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{a[1] / reps:8.3f} {a[0] // reps:4d} {1e3 * a[1] / a[0]:8.1f} {a[2] / a[1] / 1e9:6.1f}  {key}")
