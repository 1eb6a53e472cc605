"""End-to-end error of the two matrix paths against the CPU oracle (KPDetector -> DenseMotion -> RaftFlow, 256^2, B=2, eval):
prints max / mean |out - oracle| for MRFA_MFMA=f32 and bf16x6.  (Test infrastructure: imports the oracle.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from mrfa_amd import hip  # noqa: E402
from mrfa_amd.modules import DenseMotionNetwork, KPDetector, RaftFlow  # noqa: E402
from oracle import mrfa_oracle as O  # noqa: E402

DEV = torch.device("cuda", 0)
b, size = 2, 256
src, drv = cases.images("acc/src", b, size), cases.images("acc/drv", b, size)
kp, dm, rf = KPDetector(**cases.KP_DETECTOR_CFG), DenseMotionNetwork(**cases.DENSE_MOTION_CFG), RaftFlow(**cases.raft_cfg(size))
sds = {}
for n, m in (("kp", kp), ("dm", dm), ("rf", rf)):
    sds[n] = cases.weights_for(m.state_dict(), n)
    m.load_state_dict(sds[n])
    m.to(DEV).eval()
with torch.no_grad():
    P = {"encoder." + k: v for k, v in sds["kp"].items()}
    oks, okd = O.kp_detector(src, P, "encoder."), O.kp_detector(drv, P, "encoder.")
    od = O.dense_motion(src, okd, oks, {"dm." + k: v for k, v in sds["dm"].items()}, "dm.")
    img = torch.nn.functional.avg_pool2d(src, 4)
    oout, owarp, _ = O.raft_flow(oks["kp"], okd["kp"], od, img, src, {"rf." + k: v for k, v in sds["rf"].items()}, "rf.", size=size)
    for mode in ("f32", "bf16x6", "bf16x3", "bf16"):
        hip.set_mfma_mode(mode)
        ks, kd = kp(src.to(DEV)), kp(drv.to(DEV))
        d = dm(src.to(DEV), kd, ks)
        out, warp, _ = rf(ks["kp"], kd["kp"], d, img.to(DEV), src.to(DEV))
        e = (out.cpu() - oout).abs()
        ed = (d["deformation"].cpu() - od["deformation"]).abs()
        ek = (ks["kp"].cpu() - oks["kp"]).abs()
        print(f"{mode:7s} out: max {e.max():.3e} mean {e.mean():.3e} | deformation: max {ed.max():.3e} | kp: max {ek.max():.3e}")
