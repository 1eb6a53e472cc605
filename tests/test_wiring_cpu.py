"""CPU-only checks of the host side (`-m "not gpu"`): state_dict layout, C-ABI surface, no-fallback guarantees, and the
engine programs of the product modules executed through the C-ABI emulator against the REFERENCE goldens -- i.e. the
host logic (buffer layout, zero-copy concatenation, tape order, algebraic shortcuts of the correlation pyramid) is
verified here; the kernels themselves are verified against the same emulator on the GPU (tests/test_kernels_gpu.py)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from mrfa_amd import hip
from mrfa_amd.modules import DenseMotionNetwork, KPDetector, RaftFlow
from mrfa_amd.modules.manifest import manifest_of
from tests import cases
from tests.emu import emulated_hip
from tests.test_oracle_golden import raft_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_state_dict_layout_matches_reference(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    assert manifest_of(KPDetector(**cases.KP_DETECTOR_CFG)) == man["KPDetector"]
    assert manifest_of(DenseMotionNetwork(**cases.DENSE_MOTION_CFG)) == man["DenseMotionNetwork"]
    assert manifest_of(RaftFlow(**cases.raft_cfg(256))) == man["RaftFlow"]
    assert manifest_of(RaftFlow(**cases.raft_cfg(256, True))) == man["RaftFlow_prior_only"]


def test_reference_style_checkpoint_loads():
    """checkpoint layout of the reference: {'model': {'module.<name>': tensor}} (logger.py:50-58, train.py:94)"""
    from mrfa_amd.train import HotPath
    m = HotPath()
    ck = {"module." + k: v.clone() for k, v in m.state_dict().items()}
    m2 = HotPath()
    missing = torch.nn.DataParallel(m2).load_state_dict(ck, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


def test_c_abi_exports_every_declared_symbol():
    from mrfa_amd.build import build
    lib_path = build(verbose=False)
    header = open(os.path.join(ROOT, "include", "mrfa_hip.h")).read()
    declared = set(re.findall(r"\b(mrfa_[a-z0-9_]+)\s*\(", header))
    declared = {d for d in declared if not d.endswith("_params")}
    lib = ctypes.CDLL(lib_path)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mrfa_hip.h but not exported"
    assert declared == set(hip.EXPORTED_SYMBOLS), declared ^ set(hip.EXPORTED_SYMBOLS)
    # the library, the header and the ctypes binding state ONE ABI version (a stale library is refused by hip.lib())
    assert lib.mrfa_version() == int(re.search(r"#define MRFA_ABI_VERSION (\d+)", header).group(1)) == hip.ABI_VERSION


def test_product_fails_loudly_without_the_hip_library(monkeypatch):
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libmrfa_hip.so")
    with pytest.raises(hip.HipLibraryMissing):
        hip.lib()
    kp = KPDetector(**cases.KP_DETECTOR_CFG)
    with pytest.raises(hip.HipLibraryMissing):
        kp(torch.zeros(1, 3, 256, 256))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mrfa_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("CPU oracle", ""), f"{f} mentions oracle"
                assert "capi_emulator" not in src


@pytest.mark.parametrize("train", [False, True])
def test_prior_modules_through_emulator(golden_dir, train):
    g = _g(golden_dir, "prior.npz")
    sfx = "train" if train else "eval"
    x = cases.images("g3/src", 2, 256)
    with emulated_hip():
        kp = KPDetector(**cases.KP_DETECTOR_CFG)
        kp.load_state_dict(cases.weights_for(kp.state_dict(), "kp"))
        kp.train(train)
        with torch.no_grad():
            o = kp(x)
        assert np.abs(o["kp"].numpy() - g[f"kp_{sfx}"]).max() < 5e-5
        assert np.abs(o["jacobian"].numpy() - g[f"jac_{sfx}"]).max() < 5e-5
        dm = DenseMotionNetwork(**cases.DENSE_MOTION_CFG)
        dm.load_state_dict(cases.weights_for(dm.state_dict(), "dm"))
        dm.train(train)
        with torch.no_grad():
            o = dm(x, cases.keypoints("g3/kd", 2), cases.keypoints("g3/ks", 2))
        assert np.abs(o["deformation"].numpy() - g[f"dm_deformation_{sfx}"]).max() < 5e-5
        assert np.abs(o["occlusion"].numpy() - g[f"dm_occlusion_{sfx}"]).max() < 5e-5
        assert np.abs(o["mask"][:, :, ::4, ::4].numpy() - g[f"dm_mask_{sfx}_s4"]).max() < 5e-5
        assert o["sparse_deformed"].shape == (2, 11, 3, 64, 64) and o["logit_mask"].shape == (2, 11, 64, 64)


@pytest.mark.parametrize("run_backward", [False, True])
def test_autograd_node_is_released(run_backward):
    """The per-module autograd node holds the engine context (arenas, tapes); it must die with the outputs -- with or
    without a backward pass -- and must not keep the parameters' AccumulateGrad nodes (and their stream) alive, which
    is also what makes the step capturable into a hipGraph on another stream."""
    import gc
    import weakref
    with emulated_hip():
        kp = KPDetector(**cases.KP_DETECTOR_CFG)
        kp.load_state_dict(cases.weights_for(kp.state_dict(), "kp"))
        kp.train(True)
        o = kp(cases.images("leak/src", 1, 256))
        node = weakref.ref(o["kp"].grad_fn)
        assert node() is not None
        if run_backward:
            (o["kp"].sum() + o["jacobian"].sum()).backward()
        del o
        gc.collect()
        assert node() is None


@pytest.mark.parametrize("prior_only", [False, True])
def test_raft_flow_through_emulator(golden_dir, prior_only):
    size, b = 64, 2
    g = _g(golden_dir, "raft_64.npz")
    with emulated_hip():
        rf = RaftFlow(**cases.raft_cfg(size, prior_only))
        rf.load_state_dict(cases.weights_for(rf.state_dict(), "rf"))
        rf.eval()
        kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, "g3/raft64")
        with torch.no_grad():
            o, w, s = rf(kp_s, kp_d, dmo, img, img_full)
    sfx = ("prior_" if prior_only else "") + "eval"
    assert np.abs(o.numpy() - g[f"out_{sfx}"]).max() < 1e-4
    assert np.abs(w.numpy() - g[f"warp_{sfx}"]).max() < 1e-4
    assert np.abs(s[:, :, ::2, ::2].numpy() - g[f"strip_{sfx}"]).max() < 1e-4
    assert s.shape == (b, 1, size, (6 if prior_only else 7) * size)


@pytest.mark.parametrize("direct", [False, True])
def test_raft_flow_backward_through_emulator(golden_dir, direct, fresh_mode):
    """the hand-written backward tape (every op's gradient wiring) against the reference's autograd gradients.
    direct=True: engine.direct_param_grads() -- the backward kernels accumulate straight into pre-bound .grad tensors
    (the flat gradient buffer of the hipGraph step) and autograd is handed None for the parameters."""
    import contextlib
    from mrfa_amd import engine
    from mrfa_amd.graph import FlatGradients
    g = _g(golden_dir, "grads_64.npz")
    names = json.load(open(os.path.join(golden_dir, "grads_64_param_names.json")))
    size, b = 64, 2
    with emulated_hip():
        rf = RaftFlow(**cases.raft_cfg(size))
        rf.load_state_dict(cases.weights_for(rf.state_dict(), "rf"))
        rf.train(True)
        kp_s, kp_d, dmo, img, img_full = raft_inputs(size, b, "g4/raft")
        leaves = [t.clone().requires_grad_(True) for t in (kp_s, kp_d, dmo["deformation"], dmo["occlusion"])]
        driving = cases.images("g4/drv", b, size)
        if direct:
            fg = FlatGradients(rf.parameters())
            fg.bind()
        with (engine.direct_param_grads() if direct else contextlib.nullcontext()):
            o, _, _ = rf(leaves[0], leaves[1], {"deformation": leaves[2], "occlusion": leaves[3]}, img, img_full)
            loss = (o - driving).abs().mean()
            loss.backward()
        assert not direct or fg.bound()
    assert abs(loss.item() - float(g["loss"][0])) < 1e-6
    for n, t in zip(("kp_s", "kp_d", "deformation", "occlusion"), leaves):
        assert torch.isfinite(t.grad).all(), n
        assert np.abs(t.grad.numpy() - g[f"grad_{n}"]).max() < 1e-4, n
    P = dict(rf.named_parameters())
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in P.values())
    norms = np.array([0.0 if P[n].grad is None else P[n].grad.norm().item() for n in names], np.float32)
    ref = g["param_grad_norms"]
    assert np.abs(norms - ref).max() <= 1e-4 + 1e-3 * np.abs(ref).max()
    for key in g:
        if key.startswith("pgrad_"):
            assert np.abs(P[key[6:]].grad.numpy() - g[key]).max() <= 1e-5 + 1e-3 * np.abs(g[key]).max(), key


SYNCBN_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tests.emu import emulated_hip
from mrfa_amd.modules.util import Hourglass
from mrfa_amd.utils.prng import det_uniform, fill_state_dict
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
with emulated_hip():
    m = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
    m.load_state_dict(fill_state_dict(m.state_dict(), "sbn"))
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)       # reference train.py:43
    m.train(True)
    # (DDP refuses CPU modules that contain SyncBatchNorm, so its gradient averaging is done by hand here)
    x = det_uniform("sbn/x", (4, 5, 8, 8))[rank * 2:(rank + 1) * 2]
    y = m(x)
    (y * det_uniform("sbn/w", (4, 13, 8, 8))[rank * 2:(rank + 1) * 2]).sum().div(4.0).mul(world).backward()
    for p in m.parameters():
        dist.all_reduce(p.grad)
        p.grad.div_(world)
    if rank == 0:
        torch.save({"grads": {n: p.grad for n, p in m.named_parameters()}, "y": y.detach(),
                    "rm": m.encoder.down_blocks[0].norm.running_mean.clone()}, out)
dist.destroy_process_group()
"""


def test_sync_batchnorm_world2_matches_big_batch(tmp_path):
    """SyncBatchNorm semantics (reference train.py:43): 2 ranks x 2 samples with statistics all-reduced inside the
    engine == 1 process on the 4-sample batch, for outputs, running stats and every gradient (gloo, CPU, emulated)."""
    from mrfa_amd.modules.util import Hourglass
    from mrfa_amd.utils.prng import det_uniform, fill_state_dict
    script = tmp_path / "worker.py"
    script.write_text(SYNCBN_WORKER)
    out = tmp_path / "res.pt"
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, str(out)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = torch.load(out)
    with emulated_hip():
        m = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
        m.load_state_dict(fill_state_dict(m.state_dict(), "sbn"))
        m.train(True)
        y = m(det_uniform("sbn/x", (4, 5, 8, 8)))
        (y * det_uniform("sbn/w", (4, 13, 8, 8))).sum().div(4.0).backward()
    assert torch.allclose(got["y"], y.detach()[:2], atol=1e-5)
    assert torch.allclose(got["rm"], m.encoder.down_blocks[0].norm.running_mean, atol=1e-6)
    for n, p in m.named_parameters():
        assert torch.allclose(got["grads"][n], p.grad, atol=2e-5, rtol=1e-4), n


SYNCBN_HRNET_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
torch.set_num_threads(2)
from tests.emu import emulated_hip
from tests.test_wiring_cpu import small_hrnet
from mrfa_amd.utils.prng import det_uniform
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
with emulated_hip():
    deep = int(os.environ.get("SBH_DEEP", "0"))
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(small_hrnet(deep))       # reference train.py:43
    m.train(True)
    y = m(det_uniform("sbh/x", (4, 3, 32, 32))[rank * 2:(rank + 1) * 2])
    (y * det_uniform("sbh/w", (4, 32, 8, 8))[rank * 2:(rank + 1) * 2]).sum().div(4.0).mul(world).backward()
    for p in m.parameters():
        dist.all_reduce(p.grad)
        p.grad.div_(world)
    if rank == 0:
        from mrfa_amd import engine
        torch.save({"grads": {n: p.grad for n, p in m.named_parameters()}, "y": y.detach(), "bufs": {n: b.clone() for n, b in m.named_buffers()},
                    "collectives": engine.SYNCBN_COLLECTIVES}, out)
dist.destroy_process_group()
"""


def small_hrnet(deep=0):
    """the MTIA prior's HRNet trunk with one module and one block per branch (every BatchNorm call pattern of the encoder: conv-epilogue
    statistics, the separate statistics pass behind stride-2 convolutions, residual-closing BatchNorm, 1x1 fuse layers); deep: two stage-3 modules
    (the first with all three fuse outputs: the two-step 1/4 -> 1/16 term, running sums through BatchNorm passes) and two blocks per branch"""
    from mrfa_amd.modules.transformer.hr_base import HRNET_base
    from mrfa_amd.utils.prng import fill_state_dict
    cfg = {"MODEL": {"EXTRA": {"PRETRAINED_LAYERS": [],
                               "STAGE2": dict(NUM_MODULES=1, NUM_BRANCHES=2, BLOCK="BASIC", NUM_BLOCKS=[1, 1], NUM_CHANNELS=[32, 64], FUSE_METHOD="SUM"),
                               "STAGE3": dict(NUM_MODULES=2 if deep else 1, NUM_BRANCHES=3, BLOCK="BASIC", NUM_BLOCKS=[2 if deep else 1] * 3,
                                              NUM_CHANNELS=[32, 64, 128], FUSE_METHOD="SUM")}}}
    m = HRNET_base(cfg)
    m.load_state_dict(fill_state_dict(m.state_dict(), "sbh"))
    return m


@pytest.mark.parametrize("lockstep,deep", [(1, 0), (0, 0), (1, 1)], ids=["depth_batched_collectives", "one_collective_per_layer", "depth_batched_two_modules"])
def test_sync_batchnorm_hrnet_world2_matches_big_batch(tmp_path, lockstep, deep):
    """SyncBatchNorm through the MTIA encoder's BatchNorm call patterns (transformer/hr_base.py): 2 gloo ranks x 2 samples == 1 process
    x 4 samples for the output, every running statistic / batch counter and every gradient -- with the statistics of independent layers (the
    resolution branches' blocks, the terms of a fuse layer) travelling in one collective per depth (engine.Ctx.sync_stats, the default) and with
    one collective per layer and direction (MRFA_SYNCBN_LOCKSTEP=0): 32 layers x 2 directions = 64 collectives against 48."""
    from mrfa_amd.utils.prng import det_uniform
    script = tmp_path / "worker.py"
    script.write_text(SYNCBN_HRNET_WORKER)
    out = tmp_path / "res.pt"
    port = str(37500 + os.getpid() % 2000 + 2000 * lockstep)
    env = dict(os.environ, MRFA_SYNCBN_LOCKSTEP=str(lockstep), SBH_DEEP=str(deep))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, str(out)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = torch.load(out)
    n_sync = sum(isinstance(mod, torch.nn.BatchNorm2d) for mod in small_hrnet(deep).modules())
    if lockstep and deep:
        # stage 2 as below; stage 3: two modules x (two block depths x three branches: 12 -> 4), fuse layers 7 -> 2 (three outputs) and 2 -> 1
        assert got["collectives"] == 2 * (n_sync - 3 - 16 - 5 - 1), (got["collectives"], n_sync)
    elif lockstep:
        # stage 2: two branch blocks 4 -> 2 collectives, its fuse layer 2 -> 1; stage 3: three branch blocks 6 -> 2, its (single-output) fuse layer 2 -> 1
        assert got["collectives"] == 2 * (n_sync - 8), (got["collectives"], n_sync)
    else:
        assert got["collectives"] == 2 * n_sync, (got["collectives"], n_sync)
    with emulated_hip():
        m = small_hrnet(deep)
        m.train(True)
        y = m(det_uniform("sbh/x", (4, 3, 32, 32)))
        (y * det_uniform("sbh/w", (4, 32, 8, 8))).sum().div(4.0).backward()
    assert (got["y"] - y.detach()[:2]).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    for n, b in m.named_buffers():
        if b.dtype.is_floating_point:
            assert (got["bufs"][n] - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), n
        else:
            assert int(got["bufs"][n]) == int(b) == 1, n
    for n, p in m.named_parameters():
        sc = max(p.grad.abs().max().item(), 1e-6)
        assert (got["grads"][n] - p.grad).abs().max().item() <= 2e-3 * sc + 1e-6, (n, (got["grads"][n] - p.grad).abs().max().item(), sc)


DDP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tests.emu import emulated_hip
from mrfa_amd.modules.util import Hourglass
from mrfa_amd.utils.prng import det_uniform, fill_state_dict
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
with emulated_hip():
    m = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
    m.load_state_dict(fill_state_dict(m.state_dict(), "ddp"))
    m.eval()                                  # eval-mode BN: samples are independent, so DP == big batch exactly
    ddp = torch.nn.parallel.DistributedDataParallel(m)
    x = det_uniform("ddp/x", (4, 5, 8, 8))[rank * 2:(rank + 1) * 2]
    y = ddp(x)
    (y * det_uniform("ddp/w", (4, 13, 8, 8))[rank * 2:(rank + 1) * 2]).sum().div(4.0).mul(world).backward()
    if rank == 0:
        torch.save({n: p.grad for n, p in m.named_parameters()}, out)
dist.destroy_process_group()
"""


def test_ddp_gloo_world2_matches_single_process(tmp_path):
    """Data-parallel gradient sync through the engine's single-autograd-node bridge: 2 ranks x 2 samples (gloo, CPU,
    emulated kernels) must reproduce the 1-process gradients on the 4-sample batch."""
    from mrfa_amd.modules.util import Hourglass
    from mrfa_amd.utils.prng import det_uniform, fill_state_dict
    script = tmp_path / "worker.py"
    script.write_text(DDP_WORKER)
    out = tmp_path / "grads.pt"
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, str(out)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = torch.load(out)
    with emulated_hip():
        m = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
        m.load_state_dict(fill_state_dict(m.state_dict(), "ddp"))
        m.eval()
        y = m(det_uniform("ddp/x", (4, 5, 8, 8)))
        (y * det_uniform("ddp/w", (4, 13, 8, 8))).sum().div(4.0).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(got[n], p.grad, atol=1e-5, rtol=1e-4), n


FLAT_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tests.emu import emulated_hip
from mrfa_amd.graph import FlatGradients
from mrfa_amd.modules.util import Hourglass
from mrfa_amd.utils.prng import det_uniform, fill_state_dict
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
with emulated_hip():
    m = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
    m.load_state_dict(fill_state_dict(m.state_dict(), "ddp"))
    m.eval()
    fg = FlatGradients(m.parameters())
    fg.bind()
    for it in range(2):                          # second pass: the views must have survived backward + exchange
        fg.flat.zero_()
        x = det_uniform("ddp/x", (4, 5, 8, 8))[rank * 2:(rank + 1) * 2]
        y = m(x)
        (y * det_uniform("ddp/w", (4, 13, 8, 8))[rank * 2:(rank + 1) * 2]).sum().div(4.0).mul(world).backward()
        assert fg.bound()
        fg.all_reduce()
        fg.flat.mul_(1.0 / world)
    if rank == 0:
        torch.save({n: p.grad.clone() for n, p in m.named_parameters()}, out)
dist.destroy_process_group()
"""


def test_flat_gradient_exchange_world2_matches_single_process(tmp_path):
    """The hipGraph step's data-parallel exchange (mrfa_amd/graph.py: every .grad a view of one buffer, ONE all-reduce,
    1/world) on 2 gloo ranks x 2 samples == the 1-process gradients on the 4-sample batch."""
    from mrfa_amd.modules.util import Hourglass
    from mrfa_amd.utils.prng import det_uniform, fill_state_dict
    script = tmp_path / "worker.py"
    script.write_text(FLAT_WORKER)
    out = tmp_path / "grads.pt"
    port = str(33500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, str(out)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = torch.load(out)
    with emulated_hip():
        m = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
        m.load_state_dict(fill_state_dict(m.state_dict(), "ddp"))
        m.eval()
        y = m(det_uniform("ddp/x", (4, 5, 8, 8)))
        (y * det_uniform("ddp/w", (4, 13, 8, 8))).sum().div(4.0).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(got[n], p.grad, atol=1e-5, rtol=1e-4), n


OVERLAP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
torch.set_num_threads(4)
from tests.emu import emulated_hip
from tests import cases
from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step_overlapped
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
with emulated_hip():
    model = HotPath(VOX1, prior="fomm")
    for mod, tag in ((model.encoder, "kp"), (model.dense_motion, "dm"), (model.decoder, "rf")):
        mod.load_state_dict(cases.weights_for(mod.state_dict(), tag))
    model.eval()                                     # running statistics: 2 ranks x 1 sample == 1 process x 2 samples exactly
    opt = make_optimizer(model, fused=True)
    src, drv = cases.images("ovl/src", 2, 256)[rank:rank + 1], cases.images("ovl/drv", 2, 256)[rank:rank + 1]
    loss = train_step_overlapped(model, opt, src, drv, world=world)
    torch.save({"grad_sum": opt.grads.flat.clone(), "w": opt.flat_w.clone(), "loss": float(loss)}, out + f".{rank}")
dist.destroy_process_group()
"""


def test_overlapped_exchange_world2_matches_single_process(tmp_path):
    """The hipGraph step's data-parallel schedule (mrfa_amd/graph.py: backward cut at the keypoint encoder, async all-reduce of the
    decoder / dense-motion gradient ranges beside the encoder's backward, encoder range after it, 1/world folded into FlatAdam), issued
    eagerly by train.train_step_overlapped on 2 gloo ranks x 1 sample through the emulated ABI: summed gradients / 2 and the weights after
    the step equal a 1-process step on the 2-sample batch, both ranks end with identical weights, and the split backward equals the
    single backward() of train_step."""
    from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step, train_step_overlapped
    script = tmp_path / "worker.py"
    script.write_text(OVERLAP_WORKER)
    out = str(tmp_path / "res.pt")
    port = str(35500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["grad_sum"], r1["grad_sum"]) and torch.equal(r0["w"], r1["w"])       # replicas stay replicas

    def single(step):
        with emulated_hip():
            model = HotPath(VOX1, prior="fomm")
            for mod, tag in ((model.encoder, "kp"), (model.dense_motion, "dm"), (model.decoder, "rf")):
                mod.load_state_dict(cases.weights_for(mod.state_dict(), tag))
            model.eval()
            opt = make_optimizer(model, fused=True)
            loss = step(model, opt, cases.images("ovl/src", 2, 256), cases.images("ovl/drv", 2, 256))
            return opt.grads.flat.clone(), opt.flat_w.clone(), float(loss)
    g1, w1, l1 = single(lambda m, o, s, d: train_step_overlapped(m, o, s, d, world=1))
    g2, w2, l2 = single(lambda m, o, s, d: train_step(m, o, s, d))
    assert abs(l1 - l2) <= 1e-7 and abs(0.5 * (r0["loss"] + r1["loss"]) - l1) <= 1e-6
    sc = float(g1.abs().max())
    assert float((g1 - g2).abs().max()) <= 1e-6 * sc                       # split backward == one backward
    assert float((0.5 * r0["grad_sum"] - g1).abs().max()) <= 2e-5 * sc     # mean over ranks == gradient of the global batch
    # Adam's first step moves every weight by ~lr * sign(g): equal except where |g| is at the noise level
    assert float((r0["w"] - w1).abs().mean()) <= 2e-6 and float((w1 - w2).abs().max()) <= 4.1e-4


def test_deferred_decoder_wgrads_equal_inline():
    """HotPath.defer_decoder_wgrads (engine.DeferredWgrads: the dense-motion / decoder weight-gradient launches collected during their
    backward, issued when the backward reaches the keypoint encoder, un-packed into .grad behind them, joined by HotPath.join()):
    same gradients and the same weights after the step as the in-line order; the collection really held launches."""
    from mrfa_amd import engine
    from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step
    seen = {}

    def run(defer, batch=False):
        with emulated_hip():
            model = HotPath(VOX1, prior="fomm")
            for mod, tag in ((model.encoder, "kp"), (model.dense_motion, "dm"), (model.decoder, "rf")):
                mod.load_state_dict(cases.weights_for(mod.state_dict(), tag))
            model.train(True)
            model.defer_decoder_wgrads = defer
            model._wdefer.batch = batch                  # plain weight-gradient launches as ONE parameter array (mrfa_conv2d_wgrad_multi)
            opt = make_optimizer(model, fused=True)
            if defer:
                orig = engine.DeferredWgrads.flush

                def spy(self, dev):
                    seen["thunks"] = max(seen.get("thunks", 0), len(self.thunks))
                    return orig(self, dev)
                engine.DeferredWgrads.flush = spy
            try:
                loss = train_step(model, opt, cases.images("dw/src", 1, 256), cases.images("dw/drv", 1, 256))
            finally:
                if defer:
                    engine.DeferredWgrads.flush = orig
            return opt.grads.flat.clone(), opt.flat_w.clone(), float(loss)
    g0, w0, l0 = run(False)
    g1, w1, l1 = run(True)
    assert seen.get("thunks", 0) > 50, seen                 # ~90 convolutions + the un-packing
    assert l0 == l1
    assert float((g1 - g0).abs().max()) <= 1e-6 * float(g0.abs().max())
    assert float((w1 - w0).abs().max()) <= 1e-7
    # the batched form (what the keypoint encoder's collection uses on the GPU): the same launches marshalled into one parameter array per flush
    calls = {"n": 0, "problems": 0}
    from oracle.capi_emulator import Emulator
    orig_multi = Emulator.mrfa_conv2d_wgrad_multi

    def counting(self, stream, ps, n):
        calls["n"] += 1
        calls["problems"] += n
        return orig_multi(self, stream, ps, n)
    Emulator.mrfa_conv2d_wgrad_multi = counting
    try:
        g2, w2, l2 = run(True, batch=True)
    finally:
        Emulator.mrfa_conv2d_wgrad_multi = orig_multi
    assert calls["n"] >= 1 and calls["problems"] > 50, calls
    assert l2 == l0
    assert float((g2 - g0).abs().max()) <= 1e-6 * float(g0.abs().max())
    assert float((w2 - w0).abs().max()) <= 1e-7


def test_flat_adam_matches_torch_adam_and_clip():
    """mrfa_amd.optim.FlatAdam (flat buffers + K20 entry points, emulated here) against torch.optim.Adam(betas=(0.5, 0.999))
    + clip_grad_norm_(norm_type=inf) as the reference's train.py:21,65-70 uses them: same weights after every step, with
    clipping active and inactive, across an LR change, and through state_dict round trips in both directions."""
    import copy
    import math
    from mrfa_amd.modules.util import Hourglass
    from mrfa_amd.optim import FlatAdam
    from mrfa_amd.utils.prng import det_normal, fill_state_dict

    def model():
        m = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
        m.load_state_dict(fill_state_dict(m.state_dict(), "fa"))
        return m
    with emulated_hip():
        ma, mb = model(), model()
        ga_, gb_ = [list(ma.encoder.parameters()), list(ma.decoder.parameters())], [list(mb.encoder.parameters()), list(mb.decoder.parameters())]
        oa = torch.optim.Adam([{"params": ga_[0]}, {"params": ga_[1]}], lr=2e-4, betas=(0.5, 0.999))
        ob = FlatAdam([{"params": gb_[0], "clip": 10.0}, {"params": gb_[1]}], lr=2e-4, betas=(0.5, 0.999))
        assert all(p.data_ptr() >= ob.flat_w.data_ptr() for p in mb.parameters())       # parameters re-homed
        sd0 = {k: v.clone() for k, v in mb.state_dict().items()}
        assert all(torch.equal(v, ma.state_dict()[k]) for k, v in sd0.items())            # ... with their values

        def one(step, gain):
            oa.zero_grad()
            ob.zero_grad()
            for i, (pa, pb) in enumerate(zip(ma.parameters(), mb.parameters())):
                g = det_normal(f"fa/g{step}/{i}", tuple(pa.shape)) * gain
                pa.grad = g.clone()
                pb.grad.add_(g)                                                           # into the bound flat view
            torch.nn.utils.clip_grad_norm_(ga_[0], max_norm=10.0, norm_type=math.inf)
            oa.step()
            ob.step()
            for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
                assert (pa - pb).abs().max().item() <= 2e-7, (step, n)
        one(0, 1.0)
        one(1, 300.0)                                   # |g|_inf >> 10: clipping active on group 0 only
        for g in oa.param_groups + ob.param_groups:     # MultiStepLR-style edit of param_groups
            g["lr"] = 2e-5
        one(2, 1e-3)
        # torch Adam resumes from a FlatAdam checkpoint and vice versa
        oa2 = torch.optim.Adam([{"params": ga_[0]}, {"params": ga_[1]}], lr=1.0, betas=(0.9, 0.9))
        oa2.load_state_dict(copy.deepcopy(ob.state_dict()))
        oa = oa2
        one(3, 1.0)
        ob.load_state_dict(copy.deepcopy(oa.state_dict()))
        one(4, 1.0)
        assert float(ob.state[gb_[0][0]]["step"]) == 5.0


def test_checkpoint_roundtrip_reference_layout(tmp_path):
    """save_checkpoint writes the reference's file layout ({'model': 'module.'-prefixed state_dict, 'optimizer', 'epoch'},
    logger.py:50-66); a torch.optim.Adam training loop and a FlatAdam one both resume from it, and FlatAdam resumes from a
    file written with torch.optim.Adam (the reference's own checkpoints)."""
    from mrfa_amd.modules.util import Hourglass
    from mrfa_amd.optim import FlatAdam
    from mrfa_amd.train import load_checkpoint, save_checkpoint
    from mrfa_amd.utils.prng import det_normal, fill_state_dict

    class Tiny(torch.nn.Module):                     # encoder / decoder / dense_motion attributes as in the reference's MRFA
        def __init__(self, tag):
            super().__init__()
            self.encoder = Hourglass(block_expansion=8, in_features=3, num_blocks=2, max_features=32)
            self.decoder = Hourglass(block_expansion=8, in_features=5, num_blocks=2, max_features=32)
            self.dense_motion = Hourglass(block_expansion=8, in_features=4, num_blocks=2, max_features=32)
            self.load_state_dict(fill_state_dict(self.state_dict(), tag))

    def groups(m):
        return [{"params": list(m.encoder.parameters()), "clip": 10.0}, {"params": list(m.decoder.parameters())},
                {"params": list(m.dense_motion.parameters()), "clip": 10.0}]

    def fake_step(m, opt, k):
        opt.zero_grad()
        for i, p in enumerate(m.parameters()):
            g = det_normal(f"ck/g{k}/{i}", tuple(p.shape))
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        opt.step()
    with emulated_hip():
        a = Tiny("ck")
        oa = FlatAdam(groups(a), lr=2e-4, betas=(0.5, 0.999))
        fake_step(a, oa, 0)
        f1 = str(tmp_path / "00000003-checkpoint.pth")
        save_checkpoint(f1, a, oa, epoch=3)
        raw = torch.load(f1)
        assert set(raw) == {"model", "optimizer", "epoch"} and all(k.startswith("module.") for k in raw["model"])
        # reference-style resume: torch.optim.Adam
        b = Tiny("other")
        ob = torch.optim.Adam([{"params": g["params"]} for g in groups(b)], lr=2e-4, betas=(0.5, 0.999))
        assert load_checkpoint(f1, b, ob) == 3
        # FlatAdam resume
        c = Tiny("other2")
        oc = FlatAdam(groups(c), lr=2e-4, betas=(0.5, 0.999))
        assert load_checkpoint(f1, c, oc) == 3
        for m in (b, c):
            for (n, p), (_, q) in zip(a.named_parameters(), m.named_parameters()):
                assert torch.equal(p, q), n
        fake_step(a, oa, 1)
        for g in ob.param_groups:
            pass
        for m, o in ((b, ob), (c, oc)):
            if o is ob:                                            # the reference clips encoder / dense_motion by hand
                o.zero_grad()
                for i, p in enumerate(m.parameters()):
                    p.grad = det_normal(f"ck/g1/{i}", tuple(p.shape))
                import math
                torch.nn.utils.clip_grad_norm_(m.encoder.parameters(), 10.0, norm_type=math.inf)
                torch.nn.utils.clip_grad_norm_(m.dense_motion.parameters(), 10.0, norm_type=math.inf)
                o.step()
            else:
                fake_step(m, o, 1)
            for (n, p), (_, q) in zip(a.named_parameters(), m.named_parameters()):
                assert (p - q).abs().max().item() <= 2e-7, n      # resumed optimizers continue identically
        # and the other direction: a checkpoint written with torch.optim.Adam (the reference's) into FlatAdam
        f2 = str(tmp_path / "00000004-checkpoint.pth")
        save_checkpoint(f2, b, ob, epoch=4)
        d = Tiny("other3")
        od = FlatAdam(groups(d), lr=2e-4, betas=(0.5, 0.999))
        assert load_checkpoint(f2, d, od) == 4
        fake_step(a, oa, 2)
        fake_step(d, od, 2)
        for (n, p), (_, q) in zip(a.named_parameters(), d.named_parameters()):
            assert (p - q).abs().max().item() <= 4e-7, n


CKPT_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tests.emu import emulated_hip
from mrfa_amd.modules.util import Hourglass
from mrfa_amd.train import save_checkpoint, sync_bn_buffers
from mrfa_amd.utils.prng import fill_state_dict
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
with emulated_hip():
    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = Hourglass(block_expansion=8, in_features=3, num_blocks=2, max_features=32)
    m = Tiny()
    m.load_state_dict(fill_state_dict(m.state_dict(), "ck"))
    for n, b in m.named_buffers():                      # per-rank running statistics that drifted apart
        if n.endswith("running_mean"):
            b.fill_(float(rank + 1))
    opt = torch.optim.Adam(m.parameters())
    before = {n: b.clone() for n, b in m.named_buffers()}
    # (1) the reference's pattern (train.py:89-94): ONLY rank 0 saves.  A collective inside save_checkpoint would hang here.
    if rank == 0:
        save_checkpoint(out + ".local", m, opt, epoch=1)
    assert all(torch.equal(b, before[n]) for n, b in m.named_buffers()), "a rank-local save must not touch the live buffers"
    dist.barrier()
    # (2) the explicit collective form: every rank calls, rank 0 writes the rank-independent file
    # (every rank hands in ITS OWN path, so that a write by a rank other than 0 cannot hide behind rank 0's file)
    save_checkpoint(out + f".coll.rank{rank}", m, opt, epoch=2, collective=True)
    dist.barrier()
    assert os.path.exists(out + ".coll.rank0"), "rank 0 did not write the collective checkpoint"
    assert not any(os.path.exists(out + f".coll.rank{r}") for r in range(1, world)), "a rank other than 0 wrote a checkpoint with collective=True"
    # (3) SyncBatchNorm models: buffers are identical by construction, no collective is issued even with collective=True
    sm = torch.nn.SyncBatchNorm.convert_sync_batchnorm(Tiny())
    if rank == 0:                                         # only rank 0 calls: must not block
        save_checkpoint(out + ".sync", sm, torch.optim.Adam(sm.parameters()), epoch=3, collective=True)
    dist.barrier()
dist.destroy_process_group()
"""


def test_save_checkpoint_is_rank_local_and_the_buffer_average_is_explicit(tmp_path):
    """save_checkpoint must not be a hidden collective (the reference saves under `if local_rank == 0`, train.py:89-94).
    Two gloo ranks: rank 0 alone saves without hanging and without touching its live BatchNorm buffers; collective=True on every rank
    averages the running statistics over the ranks first; SyncBatchNorm models never issue the collective."""
    script = tmp_path / "worker.py"
    script.write_text(CKPT_WORKER)
    out = str(tmp_path / "ck.pt")
    port = str(37500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    local, coll = torch.load(out + ".local"), torch.load(out + ".coll.rank0")
    assert not os.path.exists(out + ".coll.rank1")
    k = next(k for k in local["model"] if k.endswith("running_mean"))
    assert float(local["model"][k].mean()) == 1.0           # rank 0's own statistics
    assert float(coll["model"][k].mean()) == 1.5            # the mean over the two ranks
    assert local["epoch"] == 1 and coll["epoch"] == 2 and os.path.exists(out + ".sync")


def test_baseline_config1_mrfa_mtia_prior_only_plumbing():
    """BASELINE.json configs[0] / SURVEY 8(d)(1): the MRFA counterpart with the MTIA prior, prior_only=True, one 256x256 pair,
    is_train=False, on CPU (kernels = their ABI specification): output tuple shapes incl. the (1,3,256,1792) visualisation
    strip (6 prior-only levels + occlusion), finite L1 / PSNR, and agreement with the oracle's wiring."""
    import copy
    import bench
    from mrfa_amd.modules import MRFA
    from mrfa_amd.modules.util import convert_dict_to_attrit_dict
    from mrfa_amd.train import VOX1
    from mrfa_amd.utils.prng import det_uniform
    from oracle import mrfa_oracle as O
    cfg = copy.deepcopy(VOX1)
    cfg["raft_flow"]["prior_only"] = True
    cfg["train_params"].update(prior_model="mtia", bg_start=1000, num_epochs=100)
    m = MRFA(convert_dict_to_attrit_dict(cfg))
    P = bench.init_weights(m)
    m.eval()
    src, drv = det_uniform("c1/src", (1, 3, 256, 256), 0, 1), det_uniform("c1/drv", (1, 3, 256, 256), 0, 1)
    with emulated_hip(), torch.no_grad():
        gen, warp, losses, kp_s, kp_d = m({"source": src, "driving": drv}, is_train=False)
    assert gen.shape == (1, 3, 256, 256) and warp.shape == (1, 3, 256, 1792) and kp_s.shape == kp_d.shape == (1, 10, 2)
    assert losses == {} and torch.isfinite(gen).all() and torch.isfinite(warp).all()
    l1 = (gen - drv).abs().mean().item()
    psnr = 10 * np.log10(1.0 / ((gen - drv) ** 2).mean().item())
    assert np.isfinite(l1) and np.isfinite(psnr) and 0 <= gen.min() and gen.max() <= 1
    with torch.no_grad():
        g2, w2, k_s, k_d, _ = O.mrfa_forward(src, drv, {k: v.clone() for k, v in P.items()}, size=256, prior_only=True, train=False,
                                             prior="mtia")
    assert (gen - g2).abs().max().item() <= 1e-3 and (gen - g2).abs().mean().item() <= 1e-4
    assert (warp - w2).abs().max().item() <= 1e-3
    assert (kp_s - k_s["kp"]).abs().max().item() <= 1e-5 and (kp_d - k_d["kp"]).abs().max().item() <= 1e-5


# the reference's train_params blocks (config/vox1.yaml:66-101; celebvhq.yaml differs in bg_start only)
_YAML_TRAIN = dict(prior_model="mtia", num_epochs=100, num_repeats=150, epoch_milestones=[60, 90], lr=2.0e-4, batch_size=80,
                   scales=[1, 0.5, 0.25, 0.125], clip_grad=True, clip=10, bg_start=1000, checkpoint_freq=100,
                   transform_params=dict(sigma_affine=0.05, sigma_tps=0.005, points_tps=5),
                   loss_weights=dict(perceptual=[10, 10, 10, 10, 10], equivariance=10, equivariance_jacobian=10))


def _mrfa(**train_overrides):
    import copy
    from mrfa_amd.modules import MRFA
    from mrfa_amd.modules.util import convert_dict_to_attrit_dict
    from mrfa_amd.train import VOX1
    cfg = copy.deepcopy(VOX1)
    cfg["train_params"] = dict(_YAML_TRAIN, **train_overrides)
    return MRFA(convert_dict_to_attrit_dict(cfg))


@pytest.mark.parametrize("name,over", [("MRFA_vox1", {}), ("MRFA_celebvhq", {"bg_start": 0}), ("MRFA_vox1_fomm", {"prior_model": "fomm"})])
def test_whole_mrfa_state_dict_equals_the_reference_mrfa(golden_dir, name, over):
    """The checkpoint key set of the reference's OWN MRFA built from its YAML files (tools/make_goldens.py:g9_mrfa_manifest): same
    names, shapes, dtypes AND order -- `pyramid.*` / `vgg.*` at the top level (model.py:154-157), `encoder.*`, `dense_motion.*`,
    [`bg_predictor.*`,] `decoder.*`, `down.weight`.  demo.py:36-38 and Logger.load_cpk load such files with strict=True."""
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    assert manifest_of(_mrfa(**over)) == man[name]


def test_mrfa_checkpoint_roundtrip_is_strict_and_keeps_the_loss_modules(tmp_path, caplog):
    """a reference MRFA checkpoint's VGG19 / pyramid weights must land in the perceptual loss (not be dropped), files we
    write must carry them under the reference's names, and key mismatches must be reported, not filtered away."""
    import logging
    from mrfa_amd.train import HotPath, load_checkpoint, save_checkpoint
    from mrfa_amd.utils.prng import fill_state_dict
    a = _mrfa(prior_model="fomm")
    a.load_state_dict(fill_state_dict(a.state_dict(), "ckm"))
    opt = torch.optim.Adam([{"params": a.encoder.parameters()}, {"params": a.decoder.parameters()}, {"params": a.dense_motion.parameters()}])
    f = str(tmp_path / "00000007-checkpoint.pth")
    save_checkpoint(f, a, opt, epoch=7)
    raw = torch.load(f)["model"]
    assert "module.vgg.slice1.0.weight" in raw and "module.pyramid.downs.0-5.weight" in raw and not any("losses." in k for k in raw)
    b = _mrfa(prior_model="fomm")
    assert load_checkpoint(f, b) == 7
    # the SAME tensor object serves the perceptual loss: what the checkpoint carried is what the loss will use
    assert b.losses.perceptual.vgg is b.vgg and b.losses.perceptual.pyramid is b.pyramid
    for k, v in a.state_dict().items():
        assert torch.equal(b.state_dict()[k], v), k
    # a network-only model takes the same file, says what it leaves out ...
    h = HotPath(prior="fomm")
    with caplog.at_level(logging.WARNING, logger="mrfa_amd"):
        assert load_checkpoint(f, h) == 7
    assert any("loss-module entries" in r.message for r in caplog.records)
    assert torch.equal(h.decoder.refine.conv1.weight, a.decoder.refine.conv1.weight)
    # ... and anything else that does not line up raises instead of being filtered
    bad = dict(torch.load(f))
    bad["model"] = {k: v for k, v in bad["model"].items() if "refine.conv1" not in k}
    bad["model"]["module.decoder.not_a_layer.weight"] = torch.zeros(1)
    f2 = str(tmp_path / "bad.pth")
    torch.save(bad, f2)
    with pytest.raises(RuntimeError, match="missing keys.*refine.conv1.*unexpected keys.*not_a_layer"):
        load_checkpoint(f2, _mrfa(prior_model="fomm"))
    with caplog.at_level(logging.WARNING, logger="mrfa_amd"):
        load_checkpoint(f2, _mrfa(prior_model="fomm"), strict=False)          # train.py:27-32's fine-tuning load


def test_graph_capture_refuses_a_late_import(tmp_path):
    """replays are only trustworthy when DEBUG_CLR_GRAPH_PACKET_CAPTURE was 0 at HIP start-up.  A process that
    touched the device first gets a warning at import and a RuntimeError at capture time (not silently wrong replays)."""
    import subprocess
    import sys
    code = r"""
import os, sys, types, warnings
os.environ.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
import torch
torch.cuda.is_initialized = lambda: True          # stands for 'the host application already used the GPU'
sys.path.insert(0, sys.argv[1])
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    import mrfa_amd
assert mrfa_amd.GRAPH_REPLAY_UNSAFE and any("hipGraph capture" in str(x.message) for x in w), [str(x.message) for x in w]
try:
    mrfa_amd.graph_replay_safe("probe")
except RuntimeError as e:
    assert "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in str(e)
else:
    raise SystemExit("graph_replay_safe did not raise")
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    # the normal order (import mrfa_amd before any device call, or the flag exported as 0) stays silent
    code2 = "import sys; sys.path.insert(0, sys.argv[1]); import mrfa_amd; assert not mrfa_amd.GRAPH_REPLAY_UNSAFE; mrfa_amd.graph_replay_safe(); print('ok')"
    r = subprocess.run([sys.executable, "-c", code2, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("train,b", [(False, 2), (True, 4)])
def test_prior_stage_gradients_through_emulator(golden_dir, train, b, fresh_mode):
    """KPDetector / DenseMotionNetwork backward (K14-K17 specifications + tape wiring) against the reference's autograd gradients"""
    from tests.grad_checks import check_prior_stage_gradients
    with emulated_hip():
        check_prior_stage_gradients(golden_dir, train, b, "cpu")


@pytest.mark.parametrize("frames", [2, 3])
def test_statistic_groups_equal_separate_calls(frames):
    """engine.stat_groups (include/mrfa_hip.h v7): the reference's separate encoder calls -- encoder(source), encoder(driving)[, encoder(transformed
    driving)], model.py:185-186,234 -- as ONE program over the concatenated batch must give what the separate train-mode calls give: outputs,
    every parameter gradient, the running statistics after `frames` momentum updates in call order, num_batches_tracked == frames.  Through the
    HRNet trunk's BatchNorm call patterns (conv-epilogue statistics with the fused finalize, stride-2 layers, residual-closing BatchNorm, the
    first backward phase inside the consumer's data gradient), host logic + ABI specification on CPU."""
    from mrfa_amd import engine
    from mrfa_amd.utils.prng import det_uniform
    b = 2
    xs = [det_uniform(f"sg/x{i}", (b, 3, 32, 32)) * (1.0 + 0.5 * i) + 0.1 * i for i in range(frames)]       # (different statistics per call)
    ws = [det_uniform(f"sg/w{i}", (b, 32, 8, 8)) for i in range(frames)]
    with emulated_hip():
        ref = small_hrnet()
        ref.train(True)
        ys = [ref(x) for x in xs]
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        got = small_hrnet()
        got.train(True)
        with engine.stat_groups(frames):
            y = got(torch.cat(xs, 0))
        (y * torch.cat(ws, 0)).sum().backward()
    for i in range(frames):
        assert (y[i * b:(i + 1) * b] - ys[i]).abs().max().item() <= 1e-5 * max(1.0, ys[i].abs().max().item()), i
    for (n, a), (_, c) in zip(ref.named_buffers(), got.named_buffers()):
        if a.dtype.is_floating_point:
            assert (a - c).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item()), n
        else:
            assert int(a) == int(c) == frames, n
    for (n, p), (_, q) in zip(ref.named_parameters(), got.named_parameters()):
        sc = max(p.grad.abs().max().item(), 1e-6)
        assert (p.grad - q.grad).abs().max().item() <= 1e-4 * sc + 1e-7, (n, (p.grad - q.grad).abs().max().item(), sc)


@pytest.mark.parametrize("src,kernel,least", [("conv_small", "conv_small_kernel", 6), ("conv_mfma", "conv_mfma_kernel", 8), ("conv_split", "conv_bf16x6_kernel", 8),
                                              ("conv_halo", "conv_halo_kernel", 20), ("conv_lean", "conv_lean_kernel", 48), ("conv_lean", "gemm_lean_kernel", 6)])
def test_fused_finalize_waits_for_its_statistics_atomics_before_the_ticket(tmp_path, src, kernel, least):
    """the fence-free hand-offs to a launch's LAST workgroup (common.h: fused_bn_finalize -- the BatchNorm statistics, device-scope fp64
    atomics into the slots -- and splitk_last_arriver -- the partial tiles of a K split, device-scope fp32 atomics into y): a barrier, then a device-scope
    ticket.  That is only a hand-off if every thread WAITS for its atomics (s_waitcnt vmcnt(0)) before the barrier; neither gfx950's back-off barrier nor a
    relaxed ticket makes the compiler emit the wait (round 4 shipped conv_small.hip without it).  This compiles each file that draws tickets to ISA and checks, in
    every kernel and for every ticket, that a full vmcnt wait sits between the last floating-point atomic before it and the barrier before the ticket."""
    import re
    import shutil
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    out = tmp_path / (src + ".s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "mrfa_amd", "csrc"),
                           "--offload-device-only", "-S", os.path.join(ROOT, "mrfa_amd", "csrc", src + ".hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    kernels = re.split(r"^(_Z\w*" + kernel + r"\w*):[^\n]*$", out.read_text(), flags=re.M)
    checked = 0
    for name, body in zip(kernels[1::2], kernels[2::2]):
        lines = body.split("\n")
        tickets = [i for i, ln in enumerate(lines) if re.search(r"global_atomic_add(_u32)?\s+v\d+, v\d+, v\d+, .*sc0", ln)]       # the returning (ticket) atomics
        for t in tickets:
            atoms = [i for i in range(t) if re.search(r"global_atomic_add_f(64|32)", lines[i])]
            assert atoms, name
            between = lines[atoms[-1] + 1:t]
            bar = max(i for i, ln in enumerate(between) if "s_barrier" in ln)
            assert any(re.search(r"s_waitcnt\s+vmcnt\(0\)", ln) for ln in between[:bar]), \
                f"{name}: no s_waitcnt vmcnt(0) between the atomics and the barrier before the ticket at line {t}"
            checked += 1
    assert checked >= least, checked        # every instantiation of the kernel carries the epilogue


@pytest.mark.parametrize("train", [False, True], ids=["eval_bn", "train_bn"])
def test_kp_detector_occlusion_head_through_emulator_vs_reference_golden(golden_dir, train):
    """the host logic of KPDetector(estimate_occlusion=True) through the ABI emulator against the reference's outputs and autograd (VERDICT r4 missing 7)"""
    from tests import grad_checks
    from tests.emu import emulated_hip
    with emulated_hip():
        grad_checks.check_kp_occlusion_head(golden_dir, torch.device("cpu"), train)


def test_binding_constants_mirror_the_header():
    """the ctypes binding's copies of the header's layout constants (slot counts of the statistics / LayerNorm scratch buffers, ABI version)"""
    import re
    from mrfa_amd import hip
    h = open(os.path.join(ROOT, "include", "mrfa_hip.h")).read()
    const = lambda name: int(re.search(r"#define\s+" + name + r"\s+(\d+)", h).group(1))
    assert hip.ABI_VERSION == const("MRFA_ABI_VERSION")
    assert hip.STATS_SLOTS == const("MRFA_STATS_SLOTS")
    assert hip.FIN_WORDS == const("MRFA_FIN_WORDS")
    assert hip.LN_SLOTS == const("MRFA_LN_SLOTS")
    assert hip.RESIZE_SUM_TERMS == const("MRFA_RESIZE_SUM_TERMS")
