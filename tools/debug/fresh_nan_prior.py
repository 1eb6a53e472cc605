"""debug: KPDetector / DenseMotionNetwork backward in default vs all-fresh-NaN gradient-buffer mode on the GPU"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mrfa_amd import engine
from mrfa_amd.modules import KPDetector, DenseMotionNetwork
from tests import cases

dev = "cuda:0"


def run_kp(fresh):
    engine.FRESH_MIN_ELEMS = 0 if fresh else (4 << 20) // 4
    engine.FRESH_NAN = fresh
    m = KPDetector(**cases.KP_DETECTOR_CFG)
    m.load_state_dict(cases.weights_for(m.state_dict(), "kp"))
    m.to(dev).train(True)
    x = cases.images("dbg/x", int(os.environ.get("B","2")), 256).to(dev)
    o = m(x)
    ((o["kp"] * torch.arange(20, device=dev).view(1, 10, 2)).sum() + (o["jacobian"] ** 2).sum()).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters()}


def run_dm(fresh):
    engine.FRESH_MIN_ELEMS = 0 if fresh else (4 << 20) // 4
    engine.FRESH_NAN = fresh
    m = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
    m.load_state_dict(cases.weights_for(m.state_dict(), "dm"))
    m.to(dev).train(True)
    x = cases.images("dbg/x", int(os.environ.get("B","2")), 256).to(dev)
    kd, ks = cases.keypoints("dbg/kd", int(os.environ.get("B","2"))), cases.keypoints("dbg/ks", int(os.environ.get("B","2")))
    kd = {k: v.to(dev).requires_grad_(True) for k, v in kd.items()}
    ks = {k: v.to(dev).requires_grad_(True) for k, v in ks.items()}
    o = m(x, kd, ks)
    ((o["deformation"] ** 2).sum() + o["occlusion"].sum() + (o["mask"] ** 2).sum()).backward()
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in m.named_parameters()}
    g.update({"kd." + k: v.grad.clone() for k, v in kd.items()})
    g.update({"ks." + k: v.grad.clone() for k, v in ks.items()})
    return g


for name, fn in (("KPDetector", run_kp), ("DenseMotion", run_dm)):
    c = fn(True); a, b = fn(False), fn(False)
    worst = []
    big = max(v.norm().item() for v in a.values())
    for n in a:
        sc = max(a[n].norm().item(), 1e-3 * big)
        worst.append(((a[n] - c[n]).norm().item() / sc, (a[n] - b[n]).norm().item() / sc, n, bool(torch.isfinite(c[n]).all())))
    worst.sort(reverse=True)
    print(name, "worst relative |default - freshnan| (second: default vs default rerun)")
    for w in worst[:8]:
        print("   %.3e  %.3e  %s finite=%s" % w)
