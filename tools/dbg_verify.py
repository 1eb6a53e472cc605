"""Distances that GraphedTrainStep.verify() measures (replay vs eager per parameter group, and the eager noise band) for the default training step:
    MRFA_ENC_WGRAD_FANOUT=0|4 python tools/dbg_verify.py     (forces the failure message, which prints them)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mrfa_amd.graph import GraphedTrainStep
from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step
from mrfa_amd.utils.prng import det_uniform
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = HotPath(VOX1, prior="mtia"); bench.init_weights(m); m.to(dev).train(True)
opt = make_optimizer(m, fused=True)
src, drv = det_uniform("d/s", (8, 3, 256, 256), 0, 1).to(dev), det_uniform("d/d", (8, 3, 256, 256), 0, 1).to(dev)
train_step(m, opt, src, drv)
step = GraphedTrainStep(m, opt, src, drv, world=1)
for _ in range(2):
    try:
        step.verify(band_mult=0.0, tol=0.0)
    except RuntimeError as e:
        print(str(e)[:700])
