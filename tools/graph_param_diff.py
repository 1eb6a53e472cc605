"""Which parameters' gradients differ between replays of the captured forward+backward graph? (debugging aid)"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mrfa_amd.graph import GraphedTrainStep  # noqa: E402
from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step  # noqa: E402
from mrfa_amd.utils.prng import det_uniform  # noqa: E402

prior = sys.argv[1] if len(sys.argv) > 1 else "mtia"
dev = torch.device("cuda", 0)
model = HotPath(VOX1, prior=prior)
bench.init_weights(model)
model.to(dev).train(True)
opt = make_optimizer(model, fused=True)
src = det_uniform("bench/src/r0", (8, 3, 256, 256), 0, 1).to(dev)
drv = det_uniform("bench/drv/r0", (8, 3, 256, 256), 0, 1).to(dev)
train_step(model, opt, src, drv)
step = GraphedTrainStep(model, opt, src, drv)
names = {id(p): n for n, p in model.named_parameters()}
snaps = []
for k in range(3):
    step.g_fb.replay()
    torch.cuda.synchronize()
    snaps.append({names[id(p)]: p.grad.double().clone() for p in step.grads.params})
    print("replay", k, "loss", float(step.loss))
# eager reference at the same weights
from mrfa_amd import engine  # noqa: E402
from mrfa_amd.train import l1_loss  # noqa: E402
opt.zero_grad()
with engine.direct_param_grads():
    loss = l1_loss(model(src, drv), drv)
    loss.backward()
torch.cuda.synchronize()
eager = {names[id(p)]: p.grad.double().clone() for p in step.grads.params}
print("eager loss", float(loss))
opt.zero_grad()
with engine.direct_param_grads():
    loss = l1_loss(model(src, drv), drv)
    loss.backward()
torch.cuda.synchronize()
eager2 = {names[id(p)]: p.grad.double().clone() for p in step.grads.params}
print("eager2 loss", float(loss))
for label, a, b in (("replay1 vs replay0", snaps[1], snaps[0]), ("replay2 vs replay0", snaps[2], snaps[0]), ("replay0 vs eager", snaps[0], eager), ("eager2 vs eager", eager2, eager)):
    rows = sorted(((float((a[n] - b[n]).norm()), float(b[n].norm()), n) for n in a), reverse=True)
    tot = (sum(float((a[n] - b[n]).pow(2).sum()) for n in a) / sum(float(b[n].pow(2).sum()) for n in a)) ** 0.5
    print(f"== {label}: total rel L2 {tot:.4f}")
    for r in rows[:6]:
        print("   |diff| %.3e  |g| %.3e  %s" % r)
