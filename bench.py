#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: 256x256 (source, driving) pairs/s, forward+backward(+clip+Adam), fp32,
BASELINE.json configs[1] = vox1.yaml shapes with the MTIA prior (TokenPose_B; `--prior fomm` = KPDetector) and RAFT refinement,
B=8 per GPU, synthetic inputs resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --size 512 --batch 4 --inference          # BASELINE.json configs[4]: 512x512 inference-only generator path

Rank 0 prints ONE JSON line.  A step = pack + forward + backward + (flat RCCL all-reduce, N > 1) + clip + Adam, replayed
as hipGraphs (mrfa_amd/graph.py; `--no-graph` launches every kernel eagerly, with DistributedDataParallel for N > 1).
`roofline` is measured live with HIP events around every launch of the dominant kernel (the 128x128 implicit-GEMM convolution
tile, forward + data-gradient launches: `conv_bf16x6_kernel` on the bf16 matrix pipe with exactly split fp32 operands by default,
`conv_mfma_kernel<128,128,...>` on the native fp32 matrix pipe with `--mfma f32`): inside the timed region for eager launches,
and on the same step re-issued eagerly right after the timed region when it was a graph replay (events cannot be recorded
inside a replayed graph; the kernels, shapes and stream are identical).  `roofline.traffic` is NOT measured in this run: it is
read from the committed PMC passes of the same command (profiles/README.md) and labelled `traffic_source`.
`cpu_baseline` times the CPU oracle (the reference restated, oracle/mrfa_oracle.py) on a bounded sample on rank 0."""
import argparse
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see mrfa_amd/__init__.py

import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
SUSTAINED_X6_TFLOPS = 257.0            # fp32-equivalent TF/s of the split-operand step on random operand bits (see roofline.sustained_ceiling_random_operands)
PEAK_BF16_MFMA_TFLOPS = 2500.0         # same guide: BF16 dense (not the 2:1-sparsity figure)


def init_weights(model):
    """deterministic random-init weights of the BASELINE architecture (no checkpoints exist offline), name-keyed PRNG"""
    from mrfa_amd.utils.prng import fill_state_dict, fill_tokenpose_state_dict
    out = {}
    for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
        if pfx == "encoder." and model.prior == "mtia":
            sd = fill_tokenpose_state_dict(mod.state_dict(), tag=pfx)
        else:
            sd = fill_state_dict(mod.state_dict(), tag=pfx)
            for k in list(sd):
                if k.endswith("jacobian.weight"):
                    sd[k] = sd[k] * 0.05
                if k.endswith("jacobian.bias"):
                    sd[k] = torch.tensor([1.0, 0.0, 0.0, 1.0]) + sd[k] * 0.5
                if k.endswith(("refine.conv2.weight", "refine.convo2.weight")):
                    sd[k] = sd[k] * 0.3
        mod.load_state_dict(sd)
        out.update({pfx + k: v for k, v in sd.items()})
    if getattr(model, "bg_predictor", None) is not None:
        sd = fill_state_dict(model.bg_predictor.state_dict(), tag="bg_predictor.")
        for k, t in model.bg_predictor.state_dict().items():
            if t.dim() == 1 and k.endswith(".weight"):
                sd[k] = torch.ones_like(sd[k])                                      # BatchNorm scales
            elif k.endswith("fc.weight"):
                sd[k] = sd[k] * 0.02                                                # an affine close to the identity
            elif k.endswith("fc.bias"):
                sd[k] = torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0]) + sd[k] * 0.1
        model.bg_predictor.load_state_dict(sd)
        out.update({"bg_predictor." + k: v for k, v in sd.items()})
    return out


def cpu_baseline(batch: int, max_seconds: float = 60.0, prior: str = "mtia", background: bool = False):
    """The CPU oracle (oracle/mrfa_oracle.py: the reference restated, pinned to it by tests/golden) on the host cores, BASELINE.md section 4: forward +
    backward (train-mode BatchNorm, the same surrogate loss) at B = 1, best of 5 after 2 warm-ups -- `value` --, the forward alone (eval, no_grad) at
    B = 1 likewise, and ONE forward + backward pass at the bench batch if the time bound (~max_seconds of CPU work in all) still allows it."""
    from mrfa_amd.train import VOX1
    from mrfa_amd.utils.prng import det_uniform
    from oracle import mrfa_oracle as O
    from mrfa_amd.train import HotPath
    # oneDNN fp32 convs stop scaling (and collapse with 256 threads on the GPU box's 2-socket host): cap at 32 threads
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    model = HotPath(VOX1, prior=prior, background=background)
    P = {k: (v.clone().requires_grad_(not k.endswith(("running_mean", "running_var", "pos_embedding", "down.weight")))
             if v.is_floating_point() else v.clone()) for k, v in init_weights(model).items()}
    del model
    t_start = time.time()

    def fwd_bwd(b, tag):
        src, drv = det_uniform(f"cpu/src{tag}", (b, 3, 256, 256), 0, 1), det_uniform(f"cpu/drv{tag}", (b, 3, 256, 256), 0, 1)
        t0 = time.time()
        gen, _, _, _, _ = O.mrfa_forward(src, drv, P, size=256, train=True, prior=prior)
        (gen - drv).abs().mean().backward()
        for v in P.values():
            if v.is_floating_point():
                v.grad = None
        return time.time() - t0

    def fwd(b, tag):
        src, drv = det_uniform(f"cpu/src{tag}", (b, 3, 256, 256), 0, 1), det_uniform(f"cpu/drv{tag}", (b, 3, 256, 256), 0, 1)
        t0 = time.time()
        with torch.no_grad():
            O.mrfa_forward(src, drv, P, size=256, train=False, prior=prior)
        return time.time() - t0
    for k in range(2):
        fwd_bwd(1, f"w{k}")
    tb = [fwd_bwd(1, k) for k in range(5)]
    for k in range(2):
        fwd(1, f"fw{k}")
    tf = [fwd(1, f"f{k}") for k in range(5)]
    out = {"value": 1.0 / min(tb), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"B=1 fwd+bwd (train-mode BN, L1 loss) of the CPU oracle: best of 5 after 2 warm-ups ({min(tb):.2f} s; mean {sum(tb) / 5:.2f} s)",
           "forward_only_b1": {"value": round(1.0 / min(tf), 3), "unit": "pairs/s", "sample": f"B=1 forward (eval, no_grad): best of 5 after 2 warm-ups ({min(tf):.2f} s)"}}
    spent = time.time() - t_start
    est = batch * min(tb)                            # (a B-sample pass costs about B single passes)
    if batch > 1 and spent + est <= max_seconds * 1.25:
        t8 = fwd_bwd(batch, "b")
        out[f"fwd_bwd_b{batch}"] = {"value": round(batch / t8, 3), "unit": "pairs/s", "sample": f"1 x (B={batch} fwd+bwd), no warm-up: the time bound of the default run ({t8:.1f} s)"}
    else:
        out[f"fwd_bwd_b{batch}"] = {"value": None, "skipped": f"would take ~{est:.0f} s on top of {spent:.0f} s (bound {max_seconds:.0f} s)"}
    out["value"] = round(out["value"], 4)
    return out


def run_inference(a, emit=True):
    """BASELINE.json configs[4]: vox1 512x512 inference-only generator path (RaftFlow forward: generator encode, 16 384^2 correlation
    volume, 6-level refinement, deformed-feature warps, decode), bs=4, one MI355X -- the HBM-bound grid_sample stress.  One JSON line:
    value = pairs/s of the hipGraph-replayed forward; roofline = the six-level feature-warp set (grid_sample_fwd, the kernel SURVEY 8(d)
    names for this config) in algorithmic GB/s against the 8 TB/s HBM peak, each launch timed with HIP events on the launch stream."""
    import copy
    from mrfa_amd import hip
    from mrfa_amd.engine import Ctx
    from mrfa_amd.graph import GraphedForward
    from mrfa_amd.modules import RaftFlow
    from mrfa_amd.train import VOX1
    from mrfa_amd.utils.prng import det_uniform, fill_state_dict
    # replicas only: no collective exists on the inference path.  With --gpus N every rank runs its own replica on its own GPU; the
    # process group is used for the timing barrier and the max-over-ranks of the elapsed time, nothing else.
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    hip.lib()
    if a.mfma:
        hip.set_mfma_mode(a.mfma)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", init_method="env://")
        world = dist.get_world_size()
    B, size = a.batch, a.size
    h = size // 4
    cfg = copy.deepcopy(VOX1["raft_flow"])
    cfg["size"] = size
    rf = RaftFlow(**cfg)
    sd = fill_state_dict(rf.state_dict(), tag="decoder.")
    for k in list(sd):
        if k.endswith(("refine.conv2.weight", "refine.convo2.weight")):
            sd[k] = sd[k] * 0.3
    rf.load_state_dict(sd)
    rf.to(dev).eval()
    img_full = det_uniform("c5/img", (B, 3, size, size), 0, 1).to(dev)
    img = torch.nn.functional.avg_pool2d(img_full, 4)
    kp_s, kp_d = det_uniform("c5/ks", (B, 10, 2), -0.8, 0.8).to(dev), det_uniform("c5/kd", (B, 10, 2), -0.8, 0.8).to(dev)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, h), indexing="ij")
    deform = (torch.stack([xs, ys], dim=-1)[None].expand(B, h, h, 2) + det_uniform("c5/d", (B, h, h, 2), -0.1, 0.1)).contiguous().to(dev)
    dm = {"deformation": deform, "occlusion": det_uniform("c5/o", (B, 1, h, h), -2, 2).to(dev)}

    class _Fwd(torch.nn.Module):                      # GraphedForward wants model(source, driving): close over the fixed prior inputs
        def __init__(self):
            super().__init__()
            self.rf = rf

        def forward(self, full, quarter):
            return self.rf(kp_s, kp_d, dm, quarter, full)[0]
    m = _Fwd().eval()
    launch = "eager"
    with torch.no_grad():
        step = lambda: m(img_full, img)
        if not a.no_graph:
            gf = GraphedForward(m, img_full, img)     # raises if the capture cannot be trusted (no silent eager fallback)
            step = lambda: gf(gf.src, gf.drv)         # the graph's own static inputs: no per-step copy
            launch = "hipGraph"
        for _ in range(a.warmup):
            out = step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
    ms = 1e3 * dt / a.steps
    scale = (size / 256.0) ** 2
    gflop = (scale * (362.73 - 8.59) + 8.59 * scale * scale) * B          # conv FLOPs x (size/256)^2, correlation GEMM x (size/256)^4
    roof = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
    if not a.no_roofline:
        ectx = Ctx(dev, train=False, record=False)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        tot_b = tot_ms = 0.0
        n = 0
        levels = []
        for C_, r in ((512, size // 32), (512, size // 16), (512, size // 8), (256, size // 4), (128, size // 2), (64, size)):
            f = ectx.new(B, r, r, C_)
            f.st.data.normal_()
            grid = ectx.new(B, r, r, 2)
            grid.st.data.uniform_(-3, 3)
            o = ectx.new(B, r, r, C_)
            for _ in range(3):
                ectx.grid_sample(f, grid, 1, out=o)
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(20):
                ectx.grid_sample(f, grid, 1, out=o)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20
            byt = 4.0 * B * (2 * C_ * r * r + 2 * r * r)                    # read C*H*W, write C*H*W, read the (x,y) grid
            tot_b, tot_ms, n = tot_b + byt, tot_ms + t, n + 1
            levels.append({"C": C_, "res": r, "us": round(t * 1e3, 1), "GBps": round(byt / t / 1e6, 0)})
        ach = tot_b / tot_ms / 1e6
        traffic = tsrc = None
        try:                                          # HBM bytes per six-level set from the committed PMC passes of this command
            tname = next(n for n in ("r6_config5_traffic.json", "r2_config5_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            with open(os.path.join(ROOT, "profiles", tname)) as tf:
                traffic = round(json.load(tf)["hbm_bytes_per_six_level_set"] / 1e9, 3)
            tsrc = f"profiles/{tname}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (B=4, 512), not this run"
        except Exception:
            pass
        if (B, size) != (4, 512):
            traffic = tsrc = None
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
                "traffic_unit": "GB per six-level set (PMC)", "traffic_source": tsrc,
                "kernel": "grid_sample_fwd (one six-level deformed-feature warp set, flow-in-pixels sampling; the forward runs three sets + the image warp)",
                "algorithmic_GB_per_set": round(tot_b / 1e9, 3), "ms_per_set": round(tot_ms, 4), "launches_per_set": n, "levels": levels,
                "whole_forward_algorithmic_tflops": round(gflop / ms, 2),
                "whole_forward_frac_of_fp32_mfma_peak": round(gflop / ms / PEAK_FP32_MFMA_TFLOPS, 4)}
    cpu = None
    if not a.no_cpu_baseline:
        try:
            from oracle import mrfa_oracle as O
            torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
            P = {k: v.detach().cpu().clone() for k, v in rf.state_dict().items()}
            t1 = time.time()
            with torch.no_grad():
                O.raft_flow(kp_s[:1].cpu(), kp_d[:1].cpu(), {k: v[:1].cpu() for k, v in dm.items()}, img[:1].cpu(), img_full[:1].cpu(), P, "",
                            size=size, train=False)
            cdt = time.time() - t1
            cpu = {"value": 1.0 / cdt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"1 x (B=1 RaftFlow forward at {size}x{size}) of the CPU oracle in {cdt:.1f}s"}
        except Exception as ex:
            cpu = {"error": repr(ex)}
    line = {"metric": f"frames/sec ({size}x{size} source+driving pair) inference, generator path", "value": round(world * B * a.steps / dt, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("bf16" if hip.mfma_mode() == "bf16" else "f32"), "data": "synthetic",
            "config": {"workload": f"vox1.yaml RaftFlow generator path (encode + correlation + 6-level refinement + warps + decode), {size}x{size}, "
                                   f"bs={B}, inference only (eval-mode BN, no autograd)", "global_batch": world * B,
                       "parallelism": f"dp{world}" + (" (independent replicas, no collective)" if world > 1 else ""),
                       "launch": launch, "mfma": hip.mfma_mode(), "out_finite": bool(torch.isfinite(out).all()),
                       "memory_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
            "roofline": roof, "cpu_baseline": cpu}
    if not emit:
        return line
    print(json.dumps(line), flush=True)


def roofline_from_profile(prof, nprof, hip):
    """`roofline` of the dominant MFMA kernel from an eager, HIP-event-timed issue of the step (Ctx.profile records): algorithmic FLOPs of
    every conv_halo launch (forward + data gradient) over their summed launch durations, against the dense peak of the active matrix mode."""
    roof = None
    if not prof:
        return roof
    split = hip.mfma_mode() in ("bf16x6", "bf16x3", "bf16")
    nprod = {"bf16x3": 3, "bf16": 1}.get(hip.mfma_mode(), 6)
    # BM=128, BN=128, chunked (bit 0 = split-K launch of the same kernel, bit 2 = bf16x6 split-operand kernel)
    dom = (128 << 16) | (128 << 4) | (4 if split else 0)
    HALO = 1 << 28                            # mrfa_conv2d_last_config bit 28: the patch-tiled 3x3 kernel (conv_halo.hip)
    sel = [(f, e0.elapsed_time(e1)) for cfg, f, e0, e1, _ in prof if cfg >= 0 and (cfg & HALO)]
    row_tiled = [(f, e0.elapsed_time(e1)) for cfg, f, e0, e1, _ in prof if (cfg & ~1) == dom]
    dom_name = "conv_halo_kernel (3x3 patch-tiled split-operand tile, all variants: fwd + dgrad launches)"
    if not sel:                               # --mfma f32 / bf16, MRFA_CONV_HALO=0: the row-tiled 128x128 tile is the dominant kernel
        sel, dom_name = row_tiled, None
    allc = [(f, e0.elapsed_time(e1)) for cfg, f, e0, e1, _ in prof if cfg >= 0]
    if sel:
        fl, ms = sum(f for f, _ in sel), sum(t for _, t in sel)
        achieved = fl / (ms * 1e-3) / 1e12
        traffic, tsrc = None, None
        try:                                  # HBM bytes per launch from the committed PMC passes (profiles/README.md)
            if dom_name:                      # the patch-tiled kernel: the latest round's PMC passes (launch-weighted mean over its variants)
                if hip.mfma_mode() == "bf16":       # config 4 has PMC passes of its own (one plane per operand moves other LDS / HBM traffic)
                    tsrc = "profiles/r6_config4_traffic.json"
                else:
                    tsrc = next(f"profiles/r{r}_traffic.json" for r in (6, 5, 4) if os.path.exists(os.path.join(ROOT, "profiles", f"r{r}_traffic.json")))
                with open(os.path.join(ROOT, tsrc)) as tf:
                    tj = json.load(tf)["kernels"]["conv_halo_kernel"]
            elif split:                       # measured in the default (bf16x6) mode; the other split modes run the same loads / stores
                tsrc = "profiles/r2_traffic.json"
                with open(os.path.join(ROOT, "profiles", "r2_traffic.json")) as tf:
                    tj = json.load(tf)["kernels"]["conv_bf16x6_kernel<false, true, 128, 6>"]
            else:
                tsrc = "profiles/r1_traffic.json"
                with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as tf:
                    tj = json.load(tf)
            traffic = round(tj["hbm_bytes_per_launch"] / 1e9, 4)
        except Exception:
            tsrc = None
        # bf16x6: six bf16 MFMA products per fp32 multiply-add -> ceiling = bf16 dense peak / 6, in fp32-equivalent FLOPs
        peak = PEAK_BF16_MFMA_TFLOPS / nprod if split else PEAK_FP32_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "GB/launch (PMC)",
                "traffic_source": (f"{tsrc}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not this run" if tsrc else None),
                "peak_is": (f"bf16 dense MFMA peak 2500 / {nprod} split products (fp32-equivalent FLOPs)" if split
                            else "fp32 dense MFMA peak"),
                "frac_of_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                # measured, not nominal: what the kernel's [12 ds_read_b128 + 24 MFMA] step sustains on this part with random operand
                # bits (power-limited; the register-only MFMA loop: 291) -- tools/ubench/mfma_lds_mix.hip, profiles/r3_mfma_lds_mix.txt
                "sustained_ceiling_random_operands": ({"tflops": SUSTAINED_X6_TFLOPS, "frac": round(achieved / SUSTAINED_X6_TFLOPS, 4),
                                                       "source": "profiles/r3_mfma_lds_mix.txt"} if (split and nprod == 6) else None),
                "kernel": dom_name or ("conv_bf16x6_kernel 128x128 (fwd + dgrad launches)" if split
                                       else "conv_mfma_kernel<128,128,2,4,false> (fwd + dgrad launches)"),
                "row_tiled_128x128": ({"launches_per_step": len(row_tiled) / nprof, "kernel_ms_per_step": round(sum(t for _, t in row_tiled) / nprof, 2),
                                       "tflops": round(sum(f for f, _ in row_tiled) / max(sum(t for _, t in row_tiled), 1e-9) / 1e9, 2)}
                                      if (dom_name and row_tiled) else None),
                "launches_per_step": len(sel) / nprof, "avg_launch_ms": round(ms / len(sel), 4),
                "algorithmic_gflop_per_launch": round(fl / len(sel) / 1e9, 2),
                "kernel_ms_per_step": round(ms / nprof, 2),
                "timed_on": "HIP events around every launch of an EAGER re-issue of the same step right after the timed region (events cannot be recorded inside a "
                            "replayed hipGraph); profiles/r6_final_bench_b8_kernel_stats.csv is the rocprofv3 --kernel-trace --stats summary of the replays",
                "all_mfma_conv_ms_per_step": round(sum(t for _, t in allc) / nprof, 2),
                "all_mfma_conv_tflops": round(sum(f for f, _ in allc) / (sum(t for _, t in allc) * 1e-3) / 1e12, 2)}
    return roof


def run_config4(dev, steps: int = 6, warmup: int = 2):
    """BASELINE.json configs[3] on ONE GPU, for the record inside the default line (outside its timed region): celebvhq.yaml wiring (MTIA prior +
    BGMotionPredictor -> bg_param), 256 x 256, bs = 16, plain-bf16 matrix products (`--mfma bf16`: operands rounded to nearest-even bf16, fp32
    accumulate; activations are still stored as fp32 -- the bf16 STORAGE path is not built), fwd + bwd + clip + Adam as a hipGraph.
    `roofline` = the conv_halo launches against the 2 500 TF/s dense bf16 peak.  The full line is `python bench.py --background --mfma bf16 --batch 16`."""
    from mrfa_amd import hip
    from mrfa_amd.engine import Ctx
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step
    from mrfa_amd.utils.prng import det_uniform
    prev = hip.mfma_mode()
    hip.set_mfma_mode("bf16")
    try:
        B = 16
        model = HotPath(VOX1, prior="mtia", background=True)
        init_weights(model)
        model.to(dev).train(True)
        opt = make_optimizer(model, lr=VOX1["train_params"]["lr"], capturable=True, fused=True, clip=VOX1["train_params"]["clip"])
        src = det_uniform("bench4/src", (B, 3, 256, 256), 0, 1).to(dev)
        drv = det_uniform("bench4/drv", (B, 3, 256, 256), 0, 1).to(dev)
        g = GraphedTrainStep(model, opt, src, drv, clip=VOX1["train_params"]["clip"], world=1)
        g.verify(loss_tol=5e-3)                        # (bf16 products amplify summation-order noise to ~1e-3 in the loss)
        for _ in range(warmup):
            g(src, drv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = g(src, drv)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        loss_val = float(loss.item())
        verify = dict(g.last_verify)
        del g
        for prm in model.parameters():
            prm.grad = None
        train_step(model, opt, src, drv, clip=VOX1["train_params"]["clip"])
        Ctx.profile = []
        nprof = 2
        for _ in range(nprof):
            train_step(model, opt, src, drv, clip=VOX1["train_params"]["clip"])
        torch.cuda.synchronize()
        prof, Ctx.profile = Ctx.profile, None
        roof = roofline_from_profile(prof, nprof, hip)
        return {"metric": "frames/sec (256x256 source+driving pair) fwd+bwd", "value": round(B / dt, 3), "unit": "pairs/s", "ms_per_step": round(1e3 * dt, 3),
                "batch": B, "dtype": "bf16", "launch": "hipGraph", "loss": float(f"{loss_val:.6f}"), "graph_verify": verify,
                "workload": "celebvhq.yaml (bg_start 0: BGMotionPredictor -> bg_param), MTIA prior + DenseMotion + RaftFlow, 256x256, bs=16, "
                            "fwd+bwd+clip+Adam, plain-bf16 MFMA products on fp32-stored activations",
                "roofline": roof}
    finally:
        Ctx.profile = None
        hip.set_mfma_mode(prev)


def spawn_ranks(n: int, argv, script: str = None, need_gpus: bool = True) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks here, the way the reference gets its ranks
    from torch.distributed.launch (run.py:50-59, train.py:39-48).  This parent NEVER touches the GPU (no HIP call, no
    torch.cuda.is_available(), no device_count(): the GPUs are counted from the KFD sysfs topology, visible_gpus()) -- each rank is a fresh child process with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set, rank 0 prints the one JSON line on the inherited stdout.  Any rank failing ends the others
    (SIGTERM, SIGKILL after KILL_GRACE_S) and the exit code is non-zero: a run that silently used fewer ranks than asked cannot happen."""
    import socket
    import subprocess
    if need_gpus:           # (False: tests/bench_dry_run.py, the CPU rehearsal of this launcher)
        have = visible_gpus()
        if have is not None and have < n:
            print(f"[bench] --gpus {n}: only {have} GPU(s) visible on this node", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    alive = set(range(n))
    deadline = None                                   # set when a rank failed: the others get SIGTERM, then SIGKILL after the grace period
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()                      # exact children of this process, by handle
                deadline = time.monotonic() + KILL_GRACE_S
        if deadline is not None and alive and time.monotonic() > deadline:
            # a rank stuck inside an RCCL collective can ignore SIGTERM for ever: escalate (still the exact child handles)
            for o in sorted(alive):
                print(f"[bench] rank {o} did not exit {KILL_GRACE_S:.0f} s after SIGTERM: SIGKILL", file=sys.stderr)
                procs[o].kill()
            deadline = None
        time.sleep(0.2)
    return rc


KILL_GRACE_S = float(os.environ.get("MRFA_BENCH_KILL_GRACE_S", "20"))


def visible_gpus():
    """GPUs this process could use, WITHOUT initialising HIP in the launcher (it fork/execs the rank children next; on ROCm builds
    without amdsmi torch.cuda.device_count() is hipGetDeviceCount): the KFD topology's nodes with SIMDs, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None = unknown (no KFD sysfs): the ranks then find out at set_device()."""
    import glob
    nodes = 0
    found = False
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        found = True
        try:
            with open(prop) as f:
                for ln in f:
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        nodes += 1
        except OSError:
            pass
    if not found:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            nodes = min(nodes, len([x for x in v.split(",") if x.strip() != ""]))
    return nodes


def main(rank_body=None, script=None, argv=None, child_argv=None):
    """rank_body / script: tests/bench_dry_run.py re-uses this launcher (argument parsing, rank spawning, WORLD_SIZE checks) with a CPU rank
    body of its own -- the benchmark itself never reaches test infrastructure except in the `cpu_baseline` leg"""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (BASELINE configs[1]: bs=8; with --inference: 4)")
    ap.add_argument("--size", type=int, default=256, help="image size; 512 only with --inference (BASELINE configs[4])")
    ap.add_argument("--inference", action="store_true",
                    help="BASELINE configs[4]: inference-only generator path (RaftFlow forward) at --size, bs=--batch; roofline = the HBM-bound "
                         "grid_sample feature-warp set")
    ap.add_argument("--prior", choices=["mtia", "fomm"], default="mtia",
                    help="keypoint prior: mtia = TokenPose_B (BASELINE config 2, `prior_model: mtia` of vox1.yaml), fomm = KPDetector")
    ap.add_argument("--background", action="store_true",
                    help="celebvhq.yaml's `bg_start: 0`: BGMotionPredictor (resnet18) feeds bg_param to the dense-motion network (+ the "
                         "background loss with --loss reference)")
    ap.add_argument("--loss", choices=["surrogate", "reference"], default="surrogate",
                    help="surrogate = mean|gen - driving| (SURVEY 8(d), the headline); reference = the reference's generator objective: VGG19 "
                         "perceptual pyramid + equivariance terms (mrfa_amd/losses.py; VGG19 weights random: no checkpoint offline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DistributedDataParallel (RCCL) even with one rank")
    ap.add_argument("--no-forward", action="store_true", help="skip the extra forward-only (inference) measurement")
    ap.add_argument("--sync-bn", action="store_true", help="SyncBatchNorm (reference train.py:43) instead of per-GPU statistics")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (DDP for N>1) instead of replaying hipGraphs")
    ap.add_argument("--force-exchange", action="store_true", help="graph mode: initialise RCCL and run the flat gradient all-reduce even with one rank")
    ap.add_argument("--overlap-exchange", action="store_true",
                    help="graph mode, N > 1: cut graph A at the keypoint encoder and overlap the all-reduce of the other gradients with its backward (opt-in, see mrfa_amd/graph.py)")
    ap.add_argument("--mfma", choices=["f32", "bf16x6", "bf16x3", "bf16"], default=None,
                    help="matrix pipe of the 128x128 conv tiles: native fp32 MFMA, or exactly split fp32 operands on the bf16 pipe "
                         "(fp32-accurate; default: MRFA_MFMA or the library default)")
    ap.add_argument("--wgrad-stream", action="store_true", help="graph mode: weight-gradient kernels as a parallel graph branch (measured slower)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam + clip_grad_norm_ instead of the flat K20 optimizer kernels")
    ap.add_argument("--allow-eager-fallback", action="store_true",
                    help="N = 1 only: if the hipGraph capture or its verification fails, time eager launches instead of exiting non-zero "
                         "(the line then says config.launch = eager)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="mrfa_set_tuning knob (kernel selection only, never results), repeatable: A/B runs on one box, e.g. --tune conv_halo=0")
    a = ap.parse_args(argv)
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    # ---- ranks.  Under torch.distributed.run (the driver's N > 1 command) WORLD_SIZE is set and must equal --gpus; a bare
    # `python bench.py --gpus N` starts its N ranks itself, BEFORE anything in this process touches the GPU.
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            sys.exit(spawn_ranks(a.gpus, child_argv if child_argv is not None else (sys.argv[1:] if argv is None else argv), script=script, need_gpus=rank_body is None))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch one rank per GPU "
              f"(python bench.py --gpus N, or torch.distributed.run --nproc-per-node N bench.py --gpus N)", file=sys.stderr)
        sys.exit(2)
    if rank_body is not None:
        return rank_body(a)
    if a.batch is None:
        a.batch = 4 if a.inference else 8
    if a.inference:
        return run_inference(a)
    if a.size != 256:
        ap.error("--size other than 256 is the inference configuration: add --inference")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or a.force_ddp or a.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", init_method="env://")
        world = dist.get_world_size()                # what RCCL sees, not what the command line says
        if world != a.gpus:
            raise SystemExit(f"[bench] --gpus {a.gpus} but the RCCL process group has {world} rank(s)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from mrfa_amd import hip
    from mrfa_amd.engine import Ctx
    from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step
    from mrfa_amd.utils.prng import det_uniform, fill_state_dict
    hip.lib()                                       # fail loudly if the HIP library is missing
    if a.mfma:
        hip.set_mfma_mode(a.mfma)
    for kv in a.tune:
        k, v = kv.split("=")
        if hip.lib().mrfa_set_tuning(k.encode(), int(v)) < 0:
            raise SystemExit(f"[bench] --tune {kv}: unknown knob")

    model = HotPath(VOX1, prior=a.prior, background=a.background)
    init_weights(model)
    if a.sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model.to(dev).train(True)
    # SyncBatchNorm issues its statistics collectives per BatchNorm layer (group) and direction.  They CAN be captured into the hipGraph
    # (MRFA_SYNCBN_GRAPH=1; RCCL collectives are capturable) but that schedule has only ever run with a forced one-rank group:
    # until it has run at N > 1 the default for --sync-bn is eager launches (DistributedDataParallel at N > 1)
    use_graph = not (a.no_graph or (a.sync_bn and os.environ.get("MRFA_SYNCBN_GRAPH", "0") != "1") or a.force_ddp)
    ddp = (world > 1 or a.force_ddp) and not use_graph
    if ddp:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                          broadcast_buffers=False, gradient_as_bucket_view=True)
    fused = use_graph and not a.torch_adam           # FlatAdam re-homes parameters/gradients/moments into flat buffers
    opt = make_optimizer(model, lr=VOX1["train_params"]["lr"], capturable=use_graph, fused=fused, clip=VOX1["train_params"]["clip"])
    B = a.batch
    clip = VOX1["train_params"]["clip"]
    # synthetic pairs, different per rank (weak scaling: per-GPU work fixed), resident in HBM before timing
    src = det_uniform(f"bench/src/r{rank}", (B, 3, 256, 256), 0, 1).to(dev)
    drv = det_uniform(f"bench/drv/r{rank}", (B, 3, 256, 256), 0, 1).to(dev)

    def barrier():
        if world > 1 or a.force_ddp or a.force_exchange:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loss_fn = None
    if a.loss == "reference":
        from mrfa_amd.losses import GeneratorFullLoss
        from mrfa_amd.train import reference_loss
        from mrfa_amd.utils.prng import fill_state_dict as _fill
        full = GeneratorFullLoss(dict(scales=[1, 0.5, 0.25, 0.125], transform_params=dict(sigma_affine=0.05, sigma_tps=0.005, points_tps=5),
                                      loss_weights=dict(perceptual=[10, 10, 10, 10, 10], equivariance=10, equivariance_jacobian=10)))
        vsd = full.perceptual.vgg.state_dict()
        vnew = _fill({k: v for k, v in vsd.items() if k not in ("mean", "std")}, tag="vgg")
        vnew.update({k: (v.abs() * 0.5) for k, v in vnew.items() if k.endswith(".bias")})
        vnew["mean"], vnew["std"] = vsd["mean"], vsd["std"]
        full.perceptual.vgg.load_state_dict(vnew)
        full.to(dev)
        loss_fn = lambda m_, s_, d_: reference_loss(m_.module if hasattr(m_, "module") else m_, full, s_, d_)
    step = lambda: train_step(model, opt, src, drv, clip=clip, loss_fn=loss_fn)
    launch = "eager"
    gstep = gstep_info = None
    if use_graph:
        # one eager step (Adam state, scratch buffers, gather tables), then the whole step is captured into hipGraphs
        # (mrfa_amd/graph.py): graph A = pack + fwd + bwd, one flat RCCL all-reduce when N > 1, graph B = clip + Adam
        from mrfa_amd.graph import GraphedTrainStep
        if not fused:
            loss = step()                             # torch.optim.Adam: its state tensors must exist before the capture
        # FlatAdam needs no step before the capture (its state lives in the flat buffers from construction), and the capture is verified
        # at the INITIAL weights on purpose: there two passes at the same weights agree to ~1e-3 (relative L2 per parameter group), so the
        # replay-vs-eager check is sharp.  After one Adam step of the randomly initialised model (every weight moved by +-lr, including
        # those whose gradient is rounding noise) two EAGER passes differ by 16-110 % per group and the check compared noise with noise:
        # 1 of 12 runs failed it by chance (gpurun_out r4d) -- which is what round 3's per-rank retry papered over.
        ok, why = 1, None
        try:
            gstep = GraphedTrainStep(model, opt, src, drv, clip=clip, world=world, exchange=(world > 1 or a.force_exchange),
                                     overlap_exchange=(True if a.overlap_exchange else None),
                                     overlap_wgrad=a.wgrad_stream, loss_fn=loss_fn)
            ltol = 5e-3 if hip.mfma_mode() == "bf16" else 1e-4
            # replays must agree with each other and with eager passes, or the graph is not used.  ONE attempt: a failure fails the run (a
            # retry decided per rank would issue an extra round of collectives on that rank only and desynchronise the communicator)
            replay_noise = gstep.verify(loss_tol=ltol)
            gstep_info = dict(gstep.last_verify)      # (a copy: the step object itself is released before the 512 x 512 leg)
        except Exception as ex:
            ok, why = 0, ex
        if world > 1:                                 # every rank must know before anybody raises (the others sit in a collective)
            flag = torch.tensor([ok], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            step = lambda: gstep(src, drv)
            launch = "hipGraph"
        elif world > 1 or not a.allow_eager_fallback:
            # a fallback would time a different program (at N > 1: DDP buckets instead of the flat exchange): fail instead
            print(f"[bench] hipGraph capture / verification failed on rank {rank}: {why!r}", file=sys.stderr)
            if world > 1:
                torch.distributed.destroy_process_group()
            sys.exit(3)
        else:
            print(f"[bench] hipGraph capture failed: {why!r}; --allow-eager-fallback: timing eager launches", file=sys.stderr)
            for prm in model.parameters():
                prm.grad = None

    # statistics collectives of ONE step.  A captured step issued them while it was being captured (GraphedTrainStep counted them there); an eager step is
    # counted over the first warm-up step -- the step that is timed, with its gradient exchange: no extra weight update on any rank
    # (per_layer_form: what the same step issues with one collective per layer and direction, MRFA_SYNCBN_LOCKSTEP=0 -- independent layers share one here)
    syncbn_per_step = syncbn_per_layer_form = None
    if a.sync_bn and launch == "hipGraph":
        syncbn_per_step, syncbn_per_layer_form = gstep.syncbn_collectives, gstep.syncbn_exchanges
    for i in range(a.warmup):
        if a.sync_bn and launch != "hipGraph" and i == 0:
            from mrfa_amd import engine as _eng
            c0, e0 = _eng.SYNCBN_COLLECTIVES, _eng.SYNCBN_EXCHANGES
            loss = step()
            syncbn_per_step, syncbn_per_layer_form = _eng.SYNCBN_COLLECTIVES - c0, _eng.SYNCBN_EXCHANGES - e0
            continue
        loss = step()
    barrier()
    if launch == "eager" and not a.no_roofline:
        Ctx.profile = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    prof, Ctx.profile = Ctx.profile, None
    loss_val = float(loss.item())
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    nprof = a.steps
    if launch == "hipGraph" and not a.no_roofline and rank == 0:
        # roofline of the dominant kernel: the SAME step launched eagerly on the same stream with a HIP event pair
        # around every MFMA launch (events cannot be recorded inside a replayed graph); no collective in here
        bare = model
        for prm in bare.parameters():
            prm.grad = None
        nprof = max(2, min(a.steps, 4))
        train_step(bare, opt, src, drv, clip=clip, loss_fn=loss_fn)
        Ctx.profile = []
        for _ in range(nprof):
            train_step(bare, opt, src, drv, clip=clip, loss_fn=loss_fn)
        torch.cuda.synchronize()
        prof, Ctx.profile = Ctx.profile, None

    alt = alt3 = None
    if launch == "hipGraph" and hip.mfma_mode() == "bf16x6" and not a.no_forward and world == 1:      # N=1 only: a failed re-capture on ONE rank would leave the others in the all-reduce
        # for the record (outside the timed region): the same step in the opt-in 3-product mode (~1e-5 product error)
        try:
            hip.set_mfma_mode("bf16x3")
            g3 = GraphedTrainStep(model, opt, src, drv, clip=clip, world=world, exchange=(world > 1 or a.force_exchange), loss_fn=loss_fn)
            for _ in range(2):
                g3(src, drv)
            barrier()
            t3 = time.perf_counter()
            n3 = max(3, min(a.steps, 5))
            for _ in range(n3):
                g3(src, drv)
            barrier()
            d3 = (time.perf_counter() - t3) / n3
            alt3 = {"mfma": "bf16x3", "ms_per_step": round(1e3 * d3, 3), "pairs_per_s": round(world * B / d3, 3),
                    "note": "three leading split products only: ~1e-5 relative product error (opt-in; the headline is bf16x6 = fp32-accurate)"}
            del g3
        except Exception as ex:
            print(f"[bench] bf16x3 comparison run failed: {ex!r}", file=sys.stderr)
        finally:
            hip.set_mfma_mode("bf16x6")
    if launch == "hipGraph" and hip.mfma_mode() == "bf16x6" and not a.no_forward and world == 1:      # N=1 only: a failed re-capture on ONE rank would leave the others in the all-reduce
        # for the record (outside the timed region): the same step with every conv on the native fp32 matrix pipe
        try:
            hip.set_mfma_mode("f32")
            g2 = GraphedTrainStep(model, opt, src, drv, clip=clip, world=world, exchange=(world > 1 or a.force_exchange), loss_fn=loss_fn)
            for _ in range(2):
                g2(src, drv)
            barrier()
            t2 = time.perf_counter()
            n2 = max(3, min(a.steps, 5))
            for _ in range(n2):
                g2(src, drv)
            barrier()
            d2 = (time.perf_counter() - t2) / n2
            alt = {"mfma": "f32", "ms_per_step": round(1e3 * d2, 3), "pairs_per_s": round(world * B / d2, 3)}
            del g2
        except Exception as ex:
            print(f"[bench] fp32-MFMA comparison run failed: {ex!r}", file=sys.stderr)
        finally:
            hip.set_mfma_mode("bf16x6")
    fwd = None
    if not a.no_forward:
        # extra (outside the timed region): inference forward only, eval-mode BN, no autograd -- the quantity the
        # north_star's ">= 0.5 x MFMA roofline on the DenseMotion+Generator forward" target is stated on
        m = model.module if hasattr(model, "module") else model
        m.eval()
        with torch.no_grad():
            fstep = lambda: m(src, drv)
            if launch == "hipGraph":
                try:
                    from mrfa_amd.graph import GraphedForward
                    gf = GraphedForward(m, src, drv)
                    fstep = lambda: gf(src, drv)
                except Exception as ex:
                    print(f"[bench] forward hipGraph capture failed: {ex!r}", file=sys.stderr)
            for _ in range(2):
                fstep()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nf = max(3, a.steps)
            for _ in range(nf):
                fstep()
            torch.cuda.synchronize()
            fdt = (time.perf_counter() - t1) / nf
        gflop = (402.4 if a.prior == "mtia" else 375.2) * B    # whole pair incl. 2x encoder (TokenPose_B | KPDetector), forward
        fwd = {"ms_per_batch": round(1e3 * fdt, 3), "pairs_per_s_per_gpu": round(B / fdt, 2),
               "algorithmic_tflops": round(gflop / fdt / 1e3, 2), "frac_of_fp32_mfma_peak": round(gflop / fdt / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4)}
        # ... and the quantity the north_star NAMES: DenseMotionNetwork + RaftFlow forward (modules.dense_motion + modules.raft, 370.98 GF per pair:
        # BASELINE.md section 3) from given keypoints, without the keypoint encoder
        try:
            with torch.no_grad():
                kp_s, kp_d = m.encode_pair(src, drv)
                kp_s, kp_d = {k: v.clone() for k, v in kp_s.items()}, {k: v.clone() for k, v in kp_d.items()}

                class _Decode(torch.nn.Module):
                    def __init__(self, hot):
                        super().__init__()
                        self.hot = hot

                    def forward(self, s_, d_):
                        return self.hot.decode(s_, kp_s, kp_d)
                dec = _Decode(m).eval()
                dstep = lambda: dec(src, drv)
                if launch == "hipGraph":
                    from mrfa_amd.graph import GraphedForward
                    gd = GraphedForward(dec, src, drv)
                    dstep = lambda: gd(src, drv)
                for _ in range(2):
                    dstep()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nf):
                    dstep()
                torch.cuda.synchronize()
                ddt = (time.perf_counter() - t1) / nf
            fwd["densemotion_raftflow_forward"] = {
                "ms_per_batch": round(1e3 * ddt, 3), "pairs_per_s_per_gpu": round(B / ddt, 2), "gflop_per_pair": 370.98,
                "algorithmic_tflops": round(370.98 * B / ddt / 1e3, 2), "frac_of_fp32_mfma_peak": round(370.98 * B / ddt / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                "north_star_target": ">= 0.5 x the fp32-MFMA peak = <= 37.8 ms per batch of 8 = >= 212 pairs/s (BASELINE.md section 3)"}
        except Exception as ex:
            fwd["densemotion_raftflow_forward"] = {"error": repr(ex)}
        m.train(True)
    if rank == 0:
        ms_per_step = 1e3 * dt / a.steps
        value = world * B * a.steps / dt
        roof = roofline_from_profile(prof, nprof, hip)
        c5 = None
        if world == 1 and not a.no_forward and not a.background and a.loss == "surrogate":
            # for the record (outside the timed region, N = 1 only): BASELINE configs[4], the 512x512 inference-only generator path at bs=4, so
            # that the driver's default command times it too (`python bench.py --size 512 --batch 4 --inference` is the full line)
            try:
                import copy as _copy
                a5 = _copy.copy(a)
                a5.size, a5.batch, a5.steps, a5.warmup, a5.inference, a5.no_cpu_baseline = 512, 4, 10, 2, True, True
                gstep = step = None                  # release the training graphs' private pool before the 12 GiB inference run
                torch.cuda.empty_cache()
                l5 = run_inference(a5, emit=False)
                c5 = {"metric": l5["metric"], "value": l5["value"], "unit": l5["unit"], "ms_per_step": l5["ms_per_step"], "batch": 4,
                      "launch": l5["config"]["launch"], "roofline": l5["roofline"]}
            except Exception as ex:
                c5 = {"error": repr(ex)}
        c4 = None
        if world == 1 and not a.no_forward and not a.background and a.loss == "surrogate" and a.prior == "mtia" and hip.mfma_mode() == "bf16x6" and launch == "hipGraph":
            # for the record (outside the timed region, N = 1 only): BASELINE configs[3] on one GPU, so that the driver's default command carries its
            # roofline too (VERDICT r3 'weak' 9)
            try:
                gstep = step = None
                del model, opt
                torch.cuda.empty_cache()
                c4 = run_config4(dev)
                torch.cuda.empty_cache()
            except Exception as ex:
                c4 = {"error": repr(ex)}
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N=1 only: at N>1 the host cores are shared with N-1 busy ranks
            try:
                cpu = cpu_baseline(B, prior=a.prior, background=a.background)
            except Exception as ex:               # the baseline must never take the GPU number down with it
                cpu = {"error": repr(ex)}
        line = {
            "metric": "frames/sec (256x256 source+driving pair) fwd+bwd", "value": round(value, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("bf16" if hip.mfma_mode() == "bf16" else "f32"),
            "data": "synthetic",
            # the headline figures of the sub-records at the END of this line, repeated here so that a truncated tail of the line still carries them
            "summary": {"roofline_frac": (roof or {}).get("frac"),
                        "densemotion_raftflow_forward": ((fwd or {}).get("densemotion_raftflow_forward") if isinstance(fwd, dict) else None),
                        "config5_roofline_frac": ((c5 or {}).get("roofline") or {}).get("frac") if isinstance(c5, dict) else None,
                        "config5_pairs_per_s": (c5 or {}).get("value") if isinstance(c5, dict) else None,
                        "config4_roofline_frac": ((c4 or {}).get("roofline") or {}).get("frac") if isinstance(c4, dict) else None,
                        "config4_pairs_per_s": (c4 or {}).get("value") if isinstance(c4, dict) else None,
                        "cpu_baseline_pairs_per_s": (cpu or {}).get("value") if isinstance(cpu, dict) else None},
            "config": {"workload": ("celebvhq.yaml (bg_start 0: BGMotionPredictor -> bg_param), " if a.background else "vox1.yaml ") +
                                   ("MTIA (TokenPose_B)" if a.prior == "mtia" else "FOMM KPDetector") +
                                   " prior + DenseMotion + RaftFlow refinement, 256x256, "
                                   f"bs={B}/GPU, fwd+bwd+clip+Adam, train-mode BN, " +
                                   ("surrogate L1 loss" if a.loss == "surrogate" else
                                    "the reference's generator losses (VGG19 perceptual pyramid on random-init weights + equivariance, 3 encoder passes)"),
                       "global_batch": world * B, "parallelism": f"dp{world}", "prior": a.prior, "background_predictor": bool(a.background), "sync_bn": bool(a.sync_bn), "sync_bn_collectives_per_step": syncbn_per_step, "sync_bn_collectives_per_step_one_per_layer_form": syncbn_per_layer_form, "launch": launch,
                       "optimizer": "FlatAdam (K20)" if fused else "torch.optim.Adam", "mfma": hip.mfma_mode(), "loss": float(f"{loss_val:.6f}"),
                       "tuning": a.tune or None, "graph_verify": gstep_info,
                       "bn_statistics": ("SyncBatchNorm" if a.sync_bn else "per-rank batch statistics; train.sync_bn_buffers (explicit collective) averages the running "
                                         "buffers over the ranks before a checkpoint is written")},
            "roofline": roof, "cpu_baseline": cpu, "forward_only": fwd, "native_fp32_mfma_path": alt, "bf16x3_path": alt3,
            "config5_512_inference": c5, "config4_celebvhq_bs16_bf16": c4,
        }
    else:
        line = None
    if world > 1 or a.force_ddp or a.force_exchange:
        torch.distributed.barrier()               # rank 0 may arrive ~15 s late (CPU baseline): tear down together
        torch.distributed.destroy_process_group()
    if line is not None:
        # RCCL prints a version banner through C stdio; flush it first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
