"""Eager timing of the training forward+backward with the pieces of the reference's objective switched on one at a time."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mrfa_amd.losses import GeneratorFullLoss
from mrfa_amd.train import VOX1, HotPath, l1_loss
from mrfa_amd.utils.prng import det_uniform
prior = sys.argv[1] if len(sys.argv) > 1 else "mtia"
dev = torch.device("cuda", 0)
model = HotPath(VOX1, prior=prior); bench.init_weights(model); model.to(dev).train(True)
src = det_uniform("bench/src/r0", (8, 3, 256, 256), 0, 1).to(dev); drv = det_uniform("bench/drv/r0", (8, 3, 256, 256), 0, 1).to(dev)
def mk(perc, eq, eqj):
    return GeneratorFullLoss(dict(scales=[1, 0.5, 0.25, 0.125], transform_params=dict(sigma_affine=0.05, sigma_tps=0.005, points_tps=5),
                                  loss_weights=dict(perceptual=perc, equivariance=eq, equivariance_jacobian=eqj))).to(dev)
def run(name, fn, n=5):
    for _ in range(2):
        model.zero_grad(set_to_none=True); fn().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        model.zero_grad(set_to_none=True); fn().backward()
    torch.cuda.synchronize(); print(f"{name:45s} {(time.perf_counter() - t0) / n * 1e3:8.1f} ms")
def with_loss(full):
    def f():
        kp_s, kp_d = model.encode_pair(src, drv); gen = model.decode(src, kp_s, kp_d)
        return sum(v.mean() for v in full(model.encoder, drv, gen, kp_d).values()) + l1_loss(gen, drv)
    return f
run("surrogate only", lambda: l1_loss(model(src, drv), drv))
run("+ equivariance (third encoder pass)", with_loss(mk([0] * 5, 10, 0)))
run("+ equivariance + jacobian", with_loss(mk([0] * 5, 10, 10)))
run("+ perceptual only", with_loss(mk([10] * 5, 0, 0)))
run("+ everything", with_loss(mk([10] * 5, 10, 10)))

# ---- the same variants as hipGraph replays (forward + backward + clip + Adam)
from mrfa_amd.graph import GraphedTrainStep
from mrfa_amd.train import make_optimizer, train_step
opt = make_optimizer(model, fused=True)
train_step(model, opt, src, drv)
def graphed(name, loss_fn):
    g = GraphedTrainStep(model, opt, src, drv, loss_fn=loss_fn)
    for _ in range(2):
        g(src, drv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        g(src, drv)
    torch.cuda.synchronize(); print(f"graph: {name:38s} {(time.perf_counter() - t0) / 5 * 1e3:8.1f} ms")
def lf(full):
    def f(m, s, d):
        kp_s, kp_d = m.encode_pair(s, d); gen = m.decode(s, kp_s, kp_d)
        return sum(v.mean() for v in full(m.encoder, d, gen, kp_d).values()) + l1_loss(gen, d)
    return f
graphed("surrogate only", None)
graphed("+ equivariance (third encoder pass)", lf(mk([0] * 5, 10, 0)))
graphed("+ equivariance + jacobian", lf(mk([0] * 5, 10, 10)))
graphed("+ perceptual only", lf(mk([10] * 5, 0, 0)))
graphed("+ everything", lf(mk([10] * 5, 10, 10)))
# bench.py's VGG initialisation (He-uniform, positive biases: dense activations through all 16 layers)
from mrfa_amd.utils.prng import fill_state_dict as _fill
full = mk([10] * 5, 10, 10)
vsd = full.perceptual.vgg.state_dict()
vnew = _fill({k: v for k, v in vsd.items() if k not in ("mean", "std")}, tag="vgg")
vnew.update({k: (v.abs() * 0.5) for k, v in vnew.items() if k.endswith(".bias")})
vnew["mean"], vnew["std"] = vsd["mean"], vsd["std"]
full.perceptual.vgg.load_state_dict({k: v.to(dev) for k, v in vnew.items()})
graphed("+ everything, bench VGG weights", lf(full))
