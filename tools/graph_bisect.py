"""Debug helper: which part of the training step survives hipGraph capture?  python tools/graph_bisect.py [stage]
Without a stage, runs every stage in its own subprocess and prints one line per stage."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGES = ["m_base", "cls_b1"]


def run_opt(stage):
    import math
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.train import train_step
    model = _hotpath().train()
    src, drv = _pairs(1, "bis")
    kw = {}
    if "noforeach" in stage:
        kw = dict(foreach=False)
    if "fused" in stage:
        kw = dict(fused=True)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True, **kw)
    train_step(model, opt, src, drv)
    train_step(model, opt, src, drv)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if "clip" in stage or "both" in stage:
            torch.nn.utils.clip_grad_norm_(model.encoder.parameters(), max_norm=10.0, norm_type=math.inf)
        if "adam" in stage or "both" in stage:
            opt.step()
    print(stage, "captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(stage, "replayed", flush=True)


def run_cls(stage):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    model = _hotpath().train()
    src, drv = _pairs(1, "bis")
    opt = make_optimizer(model, capturable=True)
    train_step(model, opt, src, drv)
    st = GraphedTrainStep(model, opt, src, drv)
    print(stage, "captured", flush=True)
    print(stage, "replayed", float(st(src, drv)), float(st(src, drv)), flush=True)


def run_mimic(stage):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd import engine
    from mrfa_amd.train import make_optimizer, train_step
    model = _hotpath().train()
    source, driving = _pairs(1, "bis")
    opt = make_optimizer(model, capturable=True)
    if "noadam" not in stage:
        train_step(model, opt, source, driving)
    src, drv = (source, driving) if "noclone" in stage else (source.clone(), driving.clone())
    ps = [p for p in model.parameters() if p.requires_grad]
    dev = ps[0].device
    total = sum((p.numel() + 3) // 4 * 4 for p in ps)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(stream):
        for p in ps:
            p.grad = None
        saved = [b.clone() for b in model.buffers()]
        for _ in range(2 if "2warm" in stage else 1):
            (model(src, drv) - drv).abs().mean().backward()
        if "nobuf" not in stage:
            for b, sv in zip(model.buffers(), saved):
                b.copy_(sv)
        off = 0
        for p in ps:
            p.grad = None if "noflat" in stage else flat[off:off + p.numel()].view_as(p)
            off += (p.numel() + 3) // 4 * 4
    torch.cuda.current_stream(dev).wait_stream(stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    engine.CAPTURE_KEY = 0 if "nokey" in stage else 3
    print(stage, "begin capture", flush=True)
    with torch.cuda.graph(g, stream=stream):
        flat.zero_()
        gen = model(src, drv)
        loss = (gen - drv).abs().mean()
        loss.backward()
    engine.CAPTURE_KEY = 0
    print(stage, "captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(stage, "replayed", float(loss), flush=True)


def run_dbg(stage):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    model = _hotpath().train()
    src, drv = _pairs(1, "g/t")
    opt = make_optimizer(model, capturable=True)
    print("eager step1", float(train_step(model, opt, src, drv)))
    st = GraphedTrainStep(model, opt, src, drv)
    torch.cuda.synchronize()

    def show(tag):
        torch.cuda.synchronize()
        print(f"{tag}: static loss {float(st.loss):.7f}  recomputed from static gen {float((st.gen - st.drv).abs().mean()):.7f}"
              f"  gen mean {float(st.gen.mean()):.6f}  |flat|max {float(st.flat.abs().max()):.3e}", flush=True)
    for k in range(2):
        st.g_fb.replay(); show(f"g_fb #{k}")
    st.g_opt.replay(); torch.cuda.synchronize()
    for k in range(2):
        st.g_fb.replay(); show(f"after g_opt g_fb #{k}")
    with torch.no_grad():
        print("eager loss at these weights", float((model(src, drv) - drv).abs().mean()))
    for k in range(2):
        st.g_fb.replay(); show(f"after eager fwd g_fb #{k}")


def run_adam(stage):
    import math
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    src, drv = _pairs(2, "g/t")
    ma, mb = _hotpath().train(), _hotpath().train()
    oa, ob = make_optimizer(ma, capturable=True), make_optimizer(mb, capturable=True)
    train_step(ma, oa, src, drv)
    train_step(mb, ob, src, drv)
    mb.load_state_dict(ma.state_dict())
    ob.load_state_dict(oa.state_dict())
    step = GraphedTrainStep(mb, ob, src, drv, clip=10.0, world=1)

    def cmp_state(tag):
        worst = {}
        for pa, pb, (n, _) in zip(ma.parameters(), mb.parameters(), ma.named_parameters()):
            sa, sb = oa.state.get(pa, {}), ob.state.get(pb, {})
            for k in set(sa) | set(sb):
                if k not in sa or k not in sb:
                    worst[k + "_missing"] = worst.get(k + "_missing", 0) + 1
                    continue
                d = float((sa[k].float() - sb[k].float()).abs().max())
                if d > worst.get(k, (0, ""))[0]:
                    worst[k] = (d, n)
        print(tag, worst, flush=True)
    cmp_state("state before")
    wd = lambda: max(((pa - pb).abs().max().item(), n) for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()))
    print("weight diff before", wd())
    step.g_fb.replay()
    torch.cuda.synchronize()
    print("weight diff after g_fb", wd())
    pa0 = next(iter(oa.state)); print("oa step", oa.state[pa0]["step"], "ob step", ob.state[next(iter(ob.state))]["step"])
    torch.cuda.synchronize()
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        pa.grad = pb.grad.clone()
    torch.nn.utils.clip_grad_norm_(ma.encoder.parameters(), max_norm=10.0, norm_type=math.inf)
    torch.nn.utils.clip_grad_norm_(ma.dense_motion.parameters(), max_norm=10.0, norm_type=math.inf)
    wa0 = [p.detach().clone() for p in ma.parameters()]
    wb0 = [p.detach().clone() for p in mb.parameters()]
    oa.step()
    step.g_opt.replay()
    torch.cuda.synchronize()
    da = max(float((p - q).abs().max()) for p, q in zip(ma.parameters(), wa0))
    db = max(float((p - q).abs().max()) for p, q in zip(mb.parameters(), wb0))
    print("max |dw| eager", da, "graph", db)
    print("oa step", oa.state[pa0]["step"], "ob step", ob.state[next(iter(ob.state))]["step"])
    cmp_state("state after")
    w = max(((pa - pb).abs().max().item(), n) for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()))
    g = max(((pa.grad - pb.grad).abs().max().item(), n) for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()))
    print("worst weight diff", w, "worst grad diff after", g)


def run_noise(stage):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _fwd_bwd, _hotpath, _pairs
    from mrfa_amd.train import make_optimizer, train_step
    b = 8 if stage.endswith("8") else 2
    src, drv = _pairs(b, "g/t")
    m = _hotpath().train()
    if "step" in stage:
        train_step(m, make_optimizer(m), src, drv)
    l1, g1 = _fwd_bwd(m, src, drv)
    l2, g2 = _fwd_bwd(m, src, drv)
    print("losses", l1, l2)
    rows = []
    for n in g1:
        d = float(((g1[n] - g2[n]) ** 2).sum())
        rows.append((d, (d ** 0.5) / (float(g1[n].norm()) + 1e-30), float(g1[n].norm()), n))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    den = sum(r[2] ** 2 for r in rows)
    print("global rel L2 noise", (tot / den) ** 0.5)
    for r in rows[:14]:
        print("sqdiff %.3e  rel %.3e  norm %.3e  %s" % r)
    rows.sort(key=lambda r: -r[1])
    print("-- by relative noise")
    for r in rows[:10]:
        print("sqdiff %.3e  rel %.3e  norm %.3e  %s" % r)


def run_noise2(stage):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    b = 2
    src, drv = _pairs(b, "g/t")
    m = _hotpath().train()

    def once():
        for p in m.parameters():
            p.grad = None
        kp_s, kp_d = m.encoder(src), m.encoder(drv)
        dm = m.dense_motion(src, kp_d, kp_s)
        keep = {"kp_s": kp_s["kp"], "kp_d": kp_d["kp"], "jac_s": kp_s["jacobian"], "jac_d": kp_d["jacobian"],
                "deformation": dm["deformation"], "occlusion": dm["occlusion"]}
        for t in keep.values():
            t.retain_grad()
        gen, warp, occ = m.decoder(kp_s["kp"], kp_d["kp"], dm, img=m.down(src), img_full=src)
        loss = (gen - drv).abs().mean()
        loss.backward()
        out = {k: t.grad.clone() for k, t in keep.items()}
        out["w:decoder.refine.conv1"] = m.decoder.refine.conv1.weight.grad.clone()
        out["w:decoder.corr_enc.convc1"] = m.decoder.corr_enc.convc1.weight.grad.clone()
        out["w:decoder.kp.enc0"] = m.decoder.kp.encoder.down_blocks[0].conv.weight.grad.clone()
        out["w:decoder.kp_img.enc0"] = m.decoder.kp_img.encoder.down_blocks[0].conv.weight.grad.clone()
        out["w:decoder.generator.first"] = m.decoder.generator.first.conv.weight.grad.clone()
        out["w:dense_motion.mask"] = m.dense_motion.mask.weight.grad.clone()
        out["w:dense_motion.occlusion"] = m.dense_motion.occlusion.weight.grad.clone()
        return out
    a, c = once(), once()
    for k in a:
        print("%-32s rel noise %.3e   norm %.3e" % (k, float((a[k] - c[k]).norm()) / (float(a[k].norm()) + 1e-30), float(a[k].norm())))


def run_noise3(stage):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.engine import Ctx
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    with torch.no_grad():
        kp_s, kp_d = m.encoder(src), m.encoder(drv)
        dm = m.dense_motion(src, kp_d, kp_s)
        img = m.down(src)
    dm = {k: v.detach().requires_grad_(True) for k, v in dm.items()}
    runs = []
    for _ in range(2):
        Ctx.debug_backward = []
        for p in m.parameters():
            p.grad = None
        gen, _, _ = m.decoder(kp_s["kp"], kp_d["kp"], dm, img=img, img_full=src)
        (gen - drv).abs().mean().backward()
        runs.append(Ctx.debug_backward[0])
        Ctx.debug_backward = None
    a, c = runs
    print("closures", len(a), len(c))
    shown = 0
    for (i, d, fa), (_, _, fc) in zip(a, c):
        rel = abs(fa - fc) / (abs(fa) + 1e-30)
        if rel > 1e-6 or i < 12:
            print(f"{i:5d} rel {rel:.2e} fp {fa:.6e}  {d}")
            shown += 1
            if shown > 60:
                break


def run_noise_oracle(stage):
    """run-to-run gradient noise of the torch/MIOpen oracle on the GPU, same weights and inputs (is the noise ours?)"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from oracle import mrfa_oracle as O
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    P = {}
    for pfx, mod in (("encoder.", m.encoder), ("dense_motion.", m.dense_motion), ("decoder.", m.decoder)):
        for k, v in mod.state_dict().items():
            P[pfx + k] = v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.clone()
    outs = []
    for _ in range(2):
        for v in P.values():
            v.grad = None
        Q = {k: (v.detach().clone().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone()) for k, v in P.items()}
        gen = O.mrfa_forward(src, drv, Q, size=256, train=True)[0]
        loss = (gen - drv).abs().mean()
        loss.backward()
        outs.append((float(loss.detach()), {k: v.grad.clone() for k, v in Q.items() if v.is_floating_point() and v.grad is not None}))
    (l1, g1), (l2, g2) = outs
    print("oracle-on-GPU losses", l1, l2)
    tot = sum(float(((g1[n] - g2[n]) ** 2).sum()) for n in g1)
    den = sum(float((g1[n] ** 2).sum()) for n in g1)
    print("oracle-on-GPU global rel L2 gradient noise", (tot / den) ** 0.5)
    # ours against the oracle's, same weights
    from test_graph_gpu import _fwd_bwd
    lo, go = _fwd_bwd(m, src, drv)
    names = dict(m.named_parameters())
    num = sum(float(((g1[n] - go[n]) ** 2).sum()) for n in go if n in g1)
    print("ours loss", lo, "ours-vs-oracle global rel L2", (num / den) ** 0.5)


def run_replays(stage):
    """replay graph A several times at fixed weights: loss and gradients must agree to the run-to-run noise"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _fwd_bwd, _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    fused = "fused" in stage
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    opt = make_optimizer(m, capturable=True, fused=fused)
    train_step(m, opt, src, drv)
    st = GraphedTrainStep(m, opt, src, drv)
    ref = None
    import time
    for k in range(8):
        st.g_fb.replay()
        if "strsync" in stage:
            torch.cuda.current_stream().synchronize()
        else:
            torch.cuda.synchronize()
        if "sleep" in stage:
            time.sleep(0.5)
        g = st.flat.cpu() if "cpu" in stage else st.flat.clone()
        if ref is None:
            ref = g
        if "ptr" in stage:
            lp = st.loss.data_ptr()
            print(f"   clone @0x{g.data_ptr():x}..0x{g.data_ptr() + g.numel() * 4:x}  loss @0x{lp:x} inside clone: {g.data_ptr() <= lp < g.data_ptr() + g.numel() * 4}"
                  f"  flat @0x{st.flat.data_ptr():x} gen @0x{st.gen.data_ptr():x}")
            if k == 2:
                snap = torch.cuda.memory._snapshot()
                for seg in snap["segments"]:
                    a, n = seg["address"], seg["total_size"]
                    if a <= lp < a + n or a <= g.data_ptr() < a + n:
                        print(f"   segment 0x{a:x} size {n} pool {seg.get('segment_pool_id')} stream {seg.get('stream')} blocks {len(seg['blocks'])}")
        print(f"replay {k}: loss {float(st.loss):.7f} rel diff to replay 0: {float((g - ref).norm() / ref.norm()):.4f}", flush=True)
        if k == 1:
            rows = []
            for n, p in m.named_parameters():
                off = (p.grad.data_ptr() - st.flat.data_ptr()) // 4
                a, b = ref[off:off + p.numel()], g[off:off + p.numel()]
                rows.append((float((a - b).norm()), float(a.norm()), float(b.norm()), n))
            rows.sort(reverse=True)
            for r in rows[:25]:
                print("   diff %.3e  |replay0| %.3e |replay1| %.3e  %s" % r)
    for k in range(3):
        if fused:
            from test_graph_gpu import _fwd_bwd_direct
            l, _ = _fwd_bwd_direct(m, opt, src, drv)
            g = st.flat.clone()
        else:
            l, gd = _fwd_bwd(m, src, drv)
            g = torch.cat([torch.nn.functional.pad(gd[n].flatten(), (0, (-gd[n].numel()) % 4)) if n in gd else
                           torch.zeros((p.numel() + 3) // 4 * 4, device=src.device) for n, p in m.named_parameters()])
            st.grads.bind()
        print(f"eager {k}: loss {l:.7f} rel diff to replay 0: {float((g - ref).norm() / ref.norm()):.4f}", flush=True)


def run_canary(stage):
    """fill every cached-but-free block of the default pool with a sentinel, replay graph A, report what was overwritten"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    opt = make_optimizer(m, capturable=True, fused="fused" in stage)
    train_step(m, opt, src, drv)
    st = GraphedTrainStep(m, opt, src, drv)
    torch.cuda.synchronize()
    canaries = []
    for size in [st.flat.numel() * 4, 1 << 32, 1 << 31, 1 << 30, 3 << 28, 1 << 29, 3 << 27, 1 << 28, 3 << 26, 1 << 27, 3 << 25, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10, 512]:
        while True:
            free_cached = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
            if free_cached < size:
                break
            before = torch.cuda.memory_reserved()
            t = torch.full((size // 4,), 12345.0, device=src.device)
            if torch.cuda.memory_reserved() > before:      # came from a fresh hipMalloc, not from the cache: stop this size
                del t
                break
            canaries.append(t)
    print("canaries", len(canaries), "bytes", sum(c.numel() * 4 for c in canaries), "reserved-allocated now", torch.cuda.memory_reserved() - torch.cuda.memory_allocated())
    torch.cuda.synchronize()
    for rep in range(2):
        st.g_fb.replay()
        st.g_opt.replay()
        torch.cuda.synchronize()
        hits = 0
        for c in canaries:
            bad = (c != 12345.0).nonzero().flatten()
            if bad.numel():
                hits += 1
                lo, hi = int(bad.min()), int(bad.max())
                vals = c[lo:lo + 8].tolist()
                print(f"replay {rep}: canary {c.numel() * 4} B @0x{c.data_ptr():x}: {bad.numel()} floats changed in [{lo}, {hi}] span {hi - lo + 1}; first values {vals}")
                c.fill_(12345.0)
        print(f"replay {rep}: {hits} canaries hit")


def run_memhist(stage):
    """which allocations made while the graphs are captured do NOT come from the graph's private pool?"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    import mrfa_amd.graph as G
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    opt = make_optimizer(m, capturable=True)
    train_step(m, opt, src, drv)
    orig = torch.cuda.graph.__enter__

    def enter(self_):
        r = orig(self_)
        torch.cuda.memory._record_memory_history(max_entries=400000, stacks="python")
        return r
    torch.cuda.graph.__enter__ = enter
    st = GraphedTrainStep(m, opt, src, drv)
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    pools = {}
    for seg in snap["segments"]:
        pools[(seg["address"], seg["total_size"])] = seg.get("segment_pool_id")
    def pool_of(addr):
        for (a, n), pid in pools.items():
            if a <= addr < a + n:
                return pid
        return None
    cap_stream = st.stream.cuda_stream
    from collections import Counter
    cnt = Counter()
    shown = 0
    for ev in snap["device_traces"][0]:
        if ev["action"] != "alloc":
            continue
        pid = pool_of(ev["addr"])
        key = (str(pid), ev["stream"] == cap_stream)
        cnt[key] += 1
        if (pid is None or tuple(pid) == (0, 0)) and shown < 12:
            shown += 1
            fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", [])[:14]]
            print(f"DEFAULT-POOL alloc during capture: size {ev['size']} stream {ev['stream']} (capture stream {cap_stream})\n     " + " < ".join(fr))
    print("alloc events by (pool, on capture stream):", dict(cnt))


def run_poison(stage):
    """NaN-fill every cached-free block of the default pool after capture; if a replay's gradients turn NaN the graph reads
    memory it does not own: bisect to the block and print who allocated that address last (allocator history)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    torch.cuda.memory._record_memory_history(max_entries=2000000, stacks="python")
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    opt = make_optimizer(m, capturable=True)
    train_step(m, opt, src, drv)
    st = GraphedTrainStep(m, opt, src, drv)
    torch.cuda.synchronize()
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    canaries = []
    for size in [st.flat.numel() * 4, 1 << 32, 1 << 31, 1 << 30, 3 << 28, 1 << 29, 3 << 27, 1 << 28, 3 << 26, 1 << 27, 3 << 25, 1 << 26, 1 << 24,
                 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10, 512]:
        while True:
            if torch.cuda.memory_reserved() - torch.cuda.memory_allocated() < size:
                break
            before = torch.cuda.memory_reserved()
            t = torch.zeros((size // 4,), device=src.device)
            if torch.cuda.memory_reserved() > before:
                del t
                break
            canaries.append(t)
    print("canaries", len(canaries))

    def nan_count(active):
        for i, c in enumerate(canaries):
            c.fill_(float("nan") if i in active else 0.0)
        st.g_fb.replay()
        torch.cuda.synchronize()
        return int(torch.isnan(st.flat).sum()), float(st.loss)
    allc = set(range(len(canaries)))
    def describe(tag):
        print(tag, "loss", float(st.loss), "gen mean/absmax", float(st.gen.mean()), float(st.gen.abs().max()), "src mean", float(st.src.mean()),
              "drv mean", float(st.drv.mean()), "nan in gen", int(torch.isnan(st.gen).sum()), "nan in flat", int(torch.isnan(st.flat).sum()))
        off = 0
        shown = 0
        for n, p in m.named_parameters():
            k = p.numel()
            if torch.isnan(st.flat[off:off + k]).any() and shown < 6:
                shown += 1
                print("    NaN grad:", n, int(torch.isnan(st.flat[off:off + k]).sum()), "of", k)
            off += (k + 3) // 4 * 4
    print("all poisoned:", nan_count(allc)); describe("  ")
    print("none poisoned:", nan_count(set())); describe("  ")
    print("none poisoned again:", nan_count(set())); describe("  ")
    torch.cuda.synchronize()
    st.g_fb.replay(); torch.cuda.synchronize(); describe("plain replay, no fills before:")
    for c in canaries[:50]:
        c.fill_(0.0)
    st.g_fb.replay(); torch.cuda.synchronize(); describe("50 fills then replay:")
    for c in canaries[:50]:
        c.fill_(0.0)
    torch.cuda.synchronize()
    st.g_fb.replay(); torch.cuda.synchronize(); describe("50 fills, sync, replay:")
    cand = sorted(allc)
    while len(cand) > 1:
        half = cand[:len(cand) // 2]
        n, _ = nan_count(set(half))
        cand = half if n else cand[len(cand) // 2:]
    c = canaries[cand[0]]
    print("culprit canary:", c.numel() * 4, "bytes at", hex(c.data_ptr()), "nan grads with only it poisoned:", nan_count({cand[0]}))
    lo, hi = c.data_ptr(), c.data_ptr() + c.numel() * 4
    last = []
    for ev in snap["device_traces"][0]:
        if ev["action"] in ("alloc", "free_requested", "free_completed") and lo <= ev["addr"] < hi:
            last.append(ev)
    print("history events inside that block:", len(last))
    for ev in last[-12:]:
        fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", [])[:12] if "torch/" not in f["filename"]]
        print(f"  {ev['action']} addr +{ev['addr'] - lo} size {ev['size']} stream {ev['stream']}: " + " < ".join(fr))


def run_seeds(stage):
    """capture fwd+bwd with retained intermediate gradients; replay with / without a big live default-pool tensor"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    keep = {}

    def body():
        for p in m.parameters():
            p.grad = None
        kp_s, kp_d = m.encoder(src), m.encoder(drv)
        dm = m.dense_motion(src, kp_d, kp_s)
        keep.clear()
        keep.update({"kp_s": kp_s["kp"], "kp_d": kp_d["kp"], "jac_s": kp_s["jacobian"], "jac_d": kp_d["jacobian"],
                     "deformation": dm["deformation"], "occlusion": dm["occlusion"]})
        for t in keep.values():
            t.retain_grad()
        gen, warp, occ = m.decoder(kp_s["kp"], kp_d["kp"], dm, img=m.down(src), img_full=src)
        loss = (gen - drv).abs().mean()
        loss.backward()
        return loss.detach()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        loss = body()
    static = dict(keep)
    wname = "encoder.predictor.decoder.up_blocks.1.conv.weight"
    W = dict(m.named_parameters())[wname]

    def snap():
        g.replay()
        torch.cuda.synchronize()
        d = {k: t.grad.detach().cpu().clone() for k, t in static.items()}
        d["w"] = W.grad.detach().cpu().clone()
        d["val:kp_s"] = static["kp_s"].detach().cpu().clone()
        return d
    a = snap()
    b = snap()
    junk = [torch.randn(116_000_000, device=src.device) for _ in range(2)]
    c = snap()
    d = snap()
    del junk
    e = snap()
    for name, x in (("replay1", b), ("with junk", c), ("with junk again", d), ("junk freed", e)):
        print(name, {k: round(float((x[k] - a[k]).norm() / (a[k].norm() + 1e-30)), 4) for k in a})


def run_lossrep(stage):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    opt = make_optimizer(m, capturable=True, fused="fused" in stage)
    train_step(m, opt, src, drv)
    st = GraphedTrainStep(m, opt, src, drv)
    vals = []
    ref = None
    for k in range(12):
        st.g_fb.replay()
        torch.cuda.synchronize()
        f = st.flat.cpu()
        ref = f if ref is None else ref
        vals.append((round(float(st.loss), 5), round(float((f - ref).norm() / ref.norm()), 3)))
    print(os.environ.get("DBG", ""), vals)


def run_guard(stage):
    """is the device clone taken after replay 0 modified by later replays?  where?"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    opt = make_optimizer(m, capturable=True)
    train_step(m, opt, src, drv)
    st = GraphedTrainStep(m, opt, src, drv)
    st.g_fb.replay()
    torch.cuda.synchronize()
    guard = torch.full_like(st.flat, 7.0)
    print(f"flat 0x{st.flat.data_ptr():x} +{st.flat.numel() * 4}  guard 0x{guard.data_ptr():x} (gap {guard.data_ptr() - st.flat.data_ptr() - st.flat.numel() * 4})")
    for k in range(3):
        st.g_fb.replay()
        torch.cuda.synchronize()
        bad = (guard != 7.0).nonzero().flatten()
        if bad.numel():
            lo, hi = int(bad.min()), int(bad.max())
            print(f"replay {k}: guard modified: {bad.numel()} floats in [{lo}, {hi}]; values {guard[lo:lo + 6].tolist()}")
            off = 0
            for n, p in m.named_parameters():
                kk = p.numel()
                c = int(((bad >= off) & (bad < off + kk)).sum())
                if c:
                    print(f"      overlaps slot of {n}: {c}/{kk}")
                    break
                off += (kk + 3) // 4 * 4
            guard.fill_(7.0)
        else:
            print(f"replay {k}: guard intact")


def run_reads(stage):
    import time
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    src, drv = _pairs(2, "g/t")
    m = _hotpath().train()
    opt = make_optimizer(m, capturable=True)
    train_step(m, opt, src, drv)
    st = GraphedTrainStep(m, opt, src, drv)
    keepalive = []
    for k in range(5):
        st.g_fb.replay()
        torch.cuda.synchronize()
        c1 = st.flat.cpu()
        g = st.flat.clone()
        torch.cuda.synchronize()
        c2 = st.flat.cpu()
        gc_ = g.cpu()
        time.sleep(0.2)
        c3 = st.flat.cpu()
        n = float(c1.norm())
        print(f"replay {k}: |cpu2-cpu1| {float((c2 - c1).norm()) / n:.4f} |clone-cpu1| {float((gc_ - c1).norm()) / n:.4f} |cpu3-cpu1| {float((c3 - c1).norm()) / n:.4f}"
              f" loss {float(st.loss):.6f}", flush=True)
        keepalive.append(g)


def run(stage):
    if stage.startswith("reads"):
        return run_reads(stage)
    if stage.startswith("guard"):
        return run_guard(stage)
    if stage.startswith("lossrep"):
        return run_lossrep(stage)
    if stage.startswith("seeds"):
        return run_seeds(stage)
    if stage.startswith("poison"):
        return run_poison(stage)
    if stage.startswith("memhist"):
        return run_memhist(stage)
    if stage.startswith("canary"):
        return run_canary(stage)
    if stage.startswith("replays"):
        return run_replays(stage)
    if stage.startswith("noise_oracle"):
        return run_noise_oracle(stage)
    if stage.startswith("noise3"):
        return run_noise3(stage)
    if stage.startswith("noise2"):
        return run_noise2(stage)
    if stage.startswith("noise"):
        return run_noise(stage)
    if stage.startswith("adam"):
        return run_adam(stage)
    if stage.startswith("dot"):
        return run_dot(stage)
    if stage.startswith("dbg"):
        return run_dbg(stage)
    if stage.startswith("m_"):
        return run_mimic(stage)
    if stage.startswith("opt_"):
        return run_opt(stage)
    if stage.startswith("cls_"):
        return run_cls(stage)
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_gpu import _hotpath, _pairs
    model = _hotpath().train()
    b = 8 if stage.endswith("b8") else 1
    src, drv = _pairs(b, "bis")
    mode = "thread_local" if stage.endswith("_tl") else ("relaxed" if stage.endswith("relaxed") else "global")

    def body():
        if stage == "fwd_train":
            return model(src, drv).sum()
        if stage.startswith("kp_fb"):
            out = model.encoder(src)
            loss = out["kp"].sum() + out["jacobian"].sum()
            loss.backward()
            return loss
        if stage == "dm_fb":
            with torch.no_grad():
                ks, kd = model.encoder(src), model.encoder(drv)
            dm = model.dense_motion(src, kd, ks)
            loss = dm["deformation"].sum() + dm["occlusion"].sum()
            loss.backward()
            return loss
        loss = (model(src, drv) - drv).abs().mean()
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ref = float(body())
    g = torch.cuda.CUDAGraph()
    from mrfa_amd import engine
    for p in model.parameters():
        p.grad = None
    flat = None
    if "flat" in stage:
        ps = [p for p in model.parameters()]
        flat = torch.zeros(sum((p.numel() + 3) // 4 * 4 for p in ps), device=src.device)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += (p.numel() + 3) // 4 * 4
    if "adamfirst" in stage:
        from mrfa_amd.train import make_optimizer, train_step
        opt = make_optimizer(model, capturable=True)
        train_step(model, opt, src, drv)
        torch.cuda.synchronize()
    if "packkey" in stage:
        engine.CAPTURE_KEY = 7
    with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
        if flat is not None and "nozero" not in stage:
            flat.zero_()
        out = body()
    engine.CAPTURE_KEY = 0
    print(stage, "captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(stage, "replayed", float(out), "eager", ref, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for st in STAGES:
            r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, st], capture_output=True, text=True, timeout=600)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if st in l or "Error" in l or "error" in l][-3:]
            print(f"== {st}: rc={r.returncode} {' | '.join(tail)}", flush=True)
