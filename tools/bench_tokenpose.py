"""TokenPose_B (MTIA prior) alone at B=8, 256x256: eager forward / forward+backward timing; run under
rocprofv3 --kernel-trace --stats for the per-kernel breakdown."""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd import hip  # noqa: E402
from mrfa_amd.modules.transformer import get_pose_net  # noqa: E402
from mrfa_amd.modules.util import convert_dict_to_attrit_dict  # noqa: E402
from mrfa_amd.train import VOX1  # noqa: E402
from mrfa_amd.utils.prng import det_uniform, fill_tokenpose_state_dict  # noqa: E402
import copy  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if len(sys.argv) > 3:
    hip.set_mfma_mode(sys.argv[3])
net = get_pose_net(convert_dict_to_attrit_dict(copy.deepcopy(VOX1["mtia_kp_detector"])), is_train=True)
net.load_state_dict(fill_tokenpose_state_dict(net.state_dict(), "encoder."))
net.to("cuda:0").train(True)
x = det_uniform("tpb/x", (B, 3, 256, 256), 0, 1).to("cuda:0")


def fwd():
    with torch.no_grad():
        return net(x)


def fb():
    net.zero_grad(set_to_none=True)
    o = net(x)
    (o["kp"].sum() + o["jacobian"].sum()).backward()


for name, fn in (("forward", fwd), ("forward+backward", fb)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    gf = 15.69 * B * (1 if name == "forward" else 3)
    print(f"{name}: {1e3 * dt:.2f} ms  ({gf / dt / 1e3:.1f} TFLOP/s algorithmic, mode {hip.mfma_mode()})")
