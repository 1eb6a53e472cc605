"""hipGraph capture of the hot path (mrfa_amd/graph.py) must be invisible in the results: a replayed step equals the
eagerly launched step (same kernels, same order), both for inference and for fwd+bwd+clip+Adam."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


def _hotpath(tag="graph"):
    from mrfa_amd.train import VOX1, HotPath
    from mrfa_amd.utils.prng import fill_state_dict
    model = HotPath(VOX1)
    for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
        sd = fill_state_dict(mod.state_dict(), tag=tag + pfx)
        for k in list(sd):
            if k.endswith("jacobian.weight"):
                sd[k] = sd[k] * 0.05
            if k.endswith("jacobian.bias"):
                sd[k] = torch.tensor([1.0, 0.0, 0.0, 1.0]) + sd[k] * 0.5
            if k.endswith(("refine.conv2.weight", "refine.convo2.weight")):
                sd[k] = sd[k] * 0.3
        mod.load_state_dict(sd)
    return model.to(DEV)


def _pairs(b, tag):
    from mrfa_amd.utils.prng import det_uniform
    return (det_uniform(f"{tag}/src", (b, 3, 256, 256), 0, 1).to(DEV), det_uniform(f"{tag}/drv", (b, 3, 256, 256), 0, 1).to(DEV))


def test_graphed_forward_equals_eager():
    from mrfa_amd.graph import GraphedForward
    model = _hotpath().eval()
    src, drv = _pairs(2, "g/a")
    src2, drv2 = _pairs(2, "g/b")
    with torch.no_grad():
        ref1 = model(src, drv).clone()
        ref2 = model(src2, drv2).clone()
    gf = GraphedForward(model, src, drv)
    out1 = gf(src, drv).clone()
    out2 = gf(src2, drv2).clone()            # new inputs through the static buffers
    out1b = gf(src, drv).clone()
    # split-K layers sum with fp32 atomics, so two launches of the same program agree to rounding, not bit for bit
    assert (out1 - ref1).abs().max().item() <= 1e-4
    assert (out2 - ref2).abs().max().item() <= 1e-4
    assert (out1 - out1b).abs().max().item() <= 1e-4
    assert (ref1 - ref2).abs().max().item() > 1e-3        # the two inputs really differ


def _grads(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _fwd_bwd(model, src, drv):
    for p in model.parameters():
        p.grad = None
    from mrfa_amd.train import l1_loss
    loss = l1_loss(model(src, drv), drv)
    loss.backward()
    return float(loss.detach()), _grads(model)


def _fwd_bwd_direct(model, opt, src, drv):
    from mrfa_amd import engine
    opt.zero_grad()
    with engine.direct_param_grads():
        from mrfa_amd.train import l1_loss
        loss = l1_loss(model(src, drv), drv)
        loss.backward()
    return float(loss.detach()), _grads(model)


@pytest.mark.parametrize("fused", [False, True])
def test_graphed_train_step_equals_eager(fused):
    """fused=False: torch.optim.Adam(capturable) + clip_grad_norm_ captured; fused=True: mrfa_amd.optim.FlatAdam (flat
    buffers, K20 kernels, parameter gradients written directly by the backward kernels).
    graph A (pack + fwd + bwd into the flat gradient buffer) against an eager fwd + bwd at the SAME weights, and graph B
    (clip + Adam) against eager clip + Adam on the SAME gradients.  (Whole trajectories are not comparable: the fp32
    atomics of the split reductions make even two eager runs drift apart by ~1e-3 in the loss after one Adam step.)"""
    import copy
    import math
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import make_optimizer, train_step
    src, drv = _pairs(2, "g/t")
    ma, mb = _hotpath().train(), _hotpath().train()
    oa, ob = make_optimizer(ma, capturable=True), make_optimizer(mb, capturable=True, fused=fused)
    init = copy.deepcopy(ma.state_dict())
    l0 = float(train_step(ma, oa, src, drv))
    train_step(mb, ob, src, drv)
    # back to the INITIAL weights (the Adam state of the step stays): one Adam step at lr 2e-4 moves this randomly initialised train-mode model to
    # a point where summation-order noise is tens of percent of the keypoint encoder's gradient, and the comparisons below then compared noise
    # with noise (1 failure in 8 runs in round 4; bench.py's verify() had the same problem and the same fix)
    ma.load_state_dict(init)
    mb.load_state_dict(ma.state_dict())                   # identical weights, BN buffers and Adam state from here on
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))    # load_state_dict shares the tensors it is given
    step = GraphedTrainStep(mb, ob, src, drv, clip=10.0, world=1)
    assert step.verify(replays=4) <= 0.3                  # replays agree with each other (graph nodes are ordered)
    for (n, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        assert torch.equal(ba, bb), f"capture changed buffer {n}"
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(pa, pb), "capture changed a weight"

    # --- graph A vs eager, same weights; eager run twice to measure the run-to-run noise floor of the gradients
    la, ga = _fwd_bwd(ma, src, drv)
    la2, ga2 = _fwd_bwd(ma, src, drv)
    step.g_fb.replay()
    torch.cuda.synchronize()
    lb, gb = float(step.loss), _grads(mb)
    assert abs(la - lb) <= 2e-6 * max(1.0, abs(la)), (la, lb)
    assert set(ga) <= set(gb)
    # The full randomly initialised train-mode model is ill-conditioned: two runs of the torch/MIOpen oracle on this GPU
    # at these weights differ by 4.6 % in the global L2 norm of the gradient, two eager runs of this engine by 3-6 %
    # (measured in round 2 with a since-removed bisection script).  The replay must sit inside that band; a missing or misrouted
    # gradient is an O(1) error.  (Tight gradient parity lives in test_parity_gpu.py on well-conditioned cases.)
    num = sum(float(((ga[n] - gb[n]) ** 2).sum()) for n in ga)
    noise = sum(float(((ga[n] - ga2[n]) ** 2).sum()) for n in ga)
    den = sum(float((ga[n] ** 2).sum()) for n in ga)
    assert math.sqrt(num / den) <= max(3.0 * math.sqrt(noise / den), 0.15), (math.sqrt(num / den), math.sqrt(noise / den))
    big = [n for n in ga if float(ga[n].norm()) >= 1e-3 * math.sqrt(den)]
    assert len(big) >= 20
    for n in big:
        dn, nn_ = float((ga[n] - gb[n]).norm()), float((ga[n] - ga2[n]).norm())
        assert dn <= 3.0 * nn_ + 0.3 * float(ga[n].norm()), (n, dn, nn_, float(ga[n].norm()))
    for n in gb:                                           # parameters eager leaves without a gradient get exact zeros
        if n not in ga:
            assert float(gb[n].abs().max()) == 0.0, n
    lb2, gb2 = _fwd_bwd_direct(mb, ob, src, drv) if fused else (lb, gb)    # eager direct-gradient mode == the replay
    assert abs(lb2 - lb) <= 2e-6
    num2 = sum(float(((gb2[n] - gb[n]) ** 2).sum()) for n in ga)
    assert math.sqrt(num2 / den) <= max(3.0 * math.sqrt(noise / den), 0.15)

    # --- graph B vs eager clip + Adam on the same gradients
    gcur = _grads(mb)                                      # what the flat buffer holds now (fused: the eager direct pass)
    for n, p in ma.named_parameters():
        p.grad = gcur[n].clone() if n in ga else None
    torch.nn.utils.clip_grad_norm_(ma.encoder.parameters(), max_norm=10.0, norm_type=math.inf)
    torch.nn.utils.clip_grad_norm_(ma.dense_motion.parameters(), max_norm=10.0, norm_type=math.inf)
    oa.step()
    step.g_opt.replay()
    torch.cuda.synchronize()
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert (pa - pb).abs().max().item() <= (2e-7 if fused else 1e-7), n
    for (n, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        if n.endswith("num_batches_tracked"):
            assert 0 < int(bb) <= int(ba), n                # the replay counts batches like an eager forward
        elif n.endswith(("running_mean", "running_var")):
            assert torch.isfinite(bb).all()

    # --- second and later replays (with unrelated device work in between) still match an eager pass at the same weights
    for _ in range(3):
        torch.randn(50_000_000, device=DEV).sum().item()
        mb.load_state_dict(ma.state_dict())
        la3, ga3 = _fwd_bwd(ma, src, drv)
        step.g_fb.replay()
        torch.cuda.synchronize()
        gb3 = _grads(mb)
        assert abs(la3 - float(step.loss)) <= 2e-6 * max(1.0, abs(la3))
        num3 = sum(float(((ga3[n] - gb3[n]) ** 2).sum()) for n in ga3)
        den3 = sum(float((ga3[n] ** 2).sum()) for n in ga3)
        assert math.sqrt(num3 / den3) <= 0.2, math.sqrt(num3 / den3)

    # --- and the replayed step trains
    losses = [float(step(src, drv)) for _ in range(3)]
    assert losses[-1] < l0, (l0, losses)


@pytest.mark.parametrize("nframes", [2, 3])
def test_batched_encoder_pass_equals_separate_calls(nframes):
    """MTIA prior, train mode: the encoder calls of a step (source, driving[, the equivariance pass]) as ONE TokenPose_B program over the concatenated
    batch with per-call BatchNorm statistics (HotPath.batched_encoder, engine.stat_groups, include/mrfa_hip.h v7) against the separate calls of the
    reference (model.py:185-186,234): keypoints and Jacobians, every encoder parameter gradient (inside the atomic-order band of two separate-call
    runs), the running statistics after `nframes` momentum updates in call order, num_batches_tracked == nframes."""
    import bench
    from mrfa_amd.train import VOX1, HotPath
    model = HotPath(VOX1, prior="mtia")
    bench.init_weights(model)
    model.to(DEV).train(True)
    frames = [_pairs(2, f"g/bat{k}")[k % 2] * (1.0 - 0.2 * k) for k in range(nframes)]
    ws = [(torch.rand(2, 10, 2, device=DEV), torch.rand(2, 10, 2, 2, device=DEV)) for _ in range(nframes)]
    saved = [b.clone() for b in model.buffers()]

    def run(batched):
        for b, sv in zip(model.buffers(), saved):
            b.copy_(sv)
        model.batched_encoder = batched
        outs = model.encode_many(frames)
        loss = sum((o["kp"] * w[0]).sum() + (o["jacobian"] * w[1]).sum() for o, w in zip(outs, ws))
        for p in model.encoder.parameters():
            p.grad = None
        loss.backward()
        model.join()
        torch.cuda.synchronize()
        return [(o["kp"].detach().clone(), o["jacobian"].detach().clone()) for o in outs], {n: b.clone() for n, b in model.encoder.named_buffers()}, \
            {n: p.grad.double().clone() for n, p in model.encoder.named_parameters() if p.grad is not None}

    run(False)
    k0, b0, g0 = run(False)
    k1, b1, g1 = run(False)
    k2, b2, g2 = run(True)
    for (ka, ja), (kb, jb) in zip(k0, k2):
        assert (ka - kb).abs().max().item() <= 1e-5 and (ja - jb).abs().max().item() <= 1e-5
    for n in b0:
        if b0[n].dtype.is_floating_point:
            assert (b2[n] - b0[n]).abs().max().item() <= 1e-5 + 1e-5 * b0[n].abs().max().item(), n
        else:
            assert int(b2[n]) == int(b0[n]) == nframes, n

    def dist(a, b):
        return (sum(float((a[n] - b[n]).pow(2).sum()) for n in b) / sum(float(b[n].pow(2).sum()) for n in b)) ** 0.5
    band = dist(g1, g0)
    assert dist(g2, g0) <= 4 * band + 1e-3, (dist(g2, g0), band)
    worst = max(float((g2[n] - g0[n]).norm() / (g0[n].norm() + 1e-3 * max(float(v.norm()) for v in g0.values()))) for n in g0)
    assert worst <= 0.05 + 8 * band, (worst, band)


def test_overlapped_exchange_graph_step_equals_the_single_graph_step():
    """Data-parallel schedule of GraphedTrainStep (graph A cut at the keypoint encoder, async RCCL all-reduce of the decoder /
    dense-motion gradient ranges beside the encoder's backward graph, encoder range after it) on a ONE-rank RCCL group, MTIA prior
    with the batched encoder pass: verify() accepts the two-graph replay against eager passes, the gradient ranges cover the flat
    buffer exactly once, and the replayed steps train like the un-split, exchange-free graph step (same first loss; the later ones
    are a chaotic trajectory at random initialisation and are only required to get below the first)."""
    import os
    import bench
    import torch.distributed as dist
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(36000 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        src, drv = _pairs(2, "g/ovl")

        def run(exchange):
            torch.manual_seed(0)
            model = HotPath(VOX1, prior="mtia")
            bench.init_weights(model)
            model.to(DEV).train(True)
            opt = make_optimizer(model, fused=True)          # (FlatAdam: no eager step before the capture -- the first replay starts from the INITIAL weights
            #                                                  in both runs, so its loss is comparable to 1e-4; after one Adam step of the randomly initialised
            #                                                  model it is already a chaotic quantity: 0.3508 vs 0.3444 was seen between the two runs)
            step = GraphedTrainStep(model, opt, src, drv, world=1, exchange=exchange, overlap_exchange=exchange)
            step.verify()
            losses = [float(step(src, drv)) for _ in range(8)]         # (from the initial weights: the first Adam steps of a random model may go up before they go down)
            torch.cuda.synchronize()
            return step, losses, opt.flat_w.clone()
        s1, l1, w1 = run(True)
        assert s1.split is not None and s1.g_tail is not None and s1.model.batched_encoder
        covered = sorted(s1.head_ranges + s1.tail_ranges)
        assert covered[0][0] == 0 and covered[-1][1] == s1.grads.total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        n_enc = sum((p.numel() + 3) // 4 * 4 for p in s1.model.encoder.parameters() if p.requires_grad)
        assert sum(hi - lo for lo, hi in s1.tail_ranges) == n_enc
        s0, l0, w0 = run(False)
        assert s0.split is None
        # the first replayed step starts from the same weights in both runs; later steps of a randomly initialised train-mode
        # model are a chaotic trajectory (step-2 losses of two IDENTICAL runs were seen at 0.444 and 0.471), so they are only required
        # to train, not to coincide
        assert abs(l1[0] - l0[0]) <= 2e-4 * max(1.0, abs(l0[0])), (l1, l0)
        # (seen once: 0.346, 0.282, 0.406 -- the third step of a B=2 run bounced; the best later loss is the robust statement)
        assert min(l1[1:]) < l1[0] and min(l0[1:]) < l0[0], (l1, l0)
        assert torch.isfinite(w1).all() and float((w1 - w0).abs().max()) <= 50 * 2e-4      # a handful of Adam steps of lr 2e-4
    finally:
        if own_group:
            dist.destroy_process_group()


def test_sync_batchnorm_collectives_are_captured_into_the_graph(monkeypatch):
    """SyncBatchNorm (reference train.py:43) no longer drops the step to eager launches: its statistics all-reduces (one per BatchNorm layer
    and direction) are captured into graph A.  One-rank RCCL group with the collectives FORCED (engine.SYNCBN_FORCE: a sum over one rank is
    the identity), FOMM prior: the capture verifies against eager passes (GraphedTrainStep.verify), ~150 collectives are issued while capturing
    and none by the replays, and the step trains like the same model without SyncBatchNorm (identical arithmetic at world 1; both runs have
    taken an eager step, the capture's warm-up and one replay by then -- a chaotic trajectory at random initialisation, so the losses are
    held to 3 %, the weights to a few Adam steps of lr 2e-4)."""
    import os
    import bench
    import torch.distributed as dist
    from mrfa_amd import engine
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(36100 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1)
    monkeypatch.setattr(engine, "SYNCBN_FORCE", True)
    try:
        src, drv = _pairs(2, "g/sbn")

        def run(sync):
            torch.manual_seed(0)
            model = HotPath(VOX1, prior="fomm")
            bench.init_weights(model)
            if sync:
                model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
            model.to(DEV).train(True)
            opt = make_optimizer(model, fused=True)
            train_step(model, opt, src, drv)
            calls = []
            real = dist.all_reduce
            monkeypatch.setattr(dist, "all_reduce", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
            step = GraphedTrainStep(model, opt, src, drv, world=1, exchange=False)
            captured = len(calls)
            step.verify()                                    # (its eager comparison passes issue their collectives from Python)
            before = len(calls)
            loss = float(step(src, drv))
            torch.cuda.synchronize()
            during_replay = len(calls) - before
            monkeypatch.setattr(dist, "all_reduce", real)
            return loss, opt.flat_w.clone(), captured, during_replay
        l1, w1, cap1, rep1 = run(True)
        l0, w0, cap0, rep0 = run(False)
        assert cap1 >= 100 and cap0 == 0, (cap1, cap0)          # ~75 BatchNorm layers x (forward + backward), issued while capturing ...
        assert rep1 == 0, rep1                                    # ... and replayed from the graph, not re-issued from Python
        assert abs(l1 - l0) <= 3e-2 * max(1.0, abs(l0)), (l1, l0)
        assert torch.isfinite(w1).all() and float((w1 - w0).abs().max()) <= 50 * 2e-4
    finally:
        import gc
        gc.collect()                       # graphs that hold captured RCCL kernels go before their communicator does
        torch.cuda.synchronize()
        if own_group:
            dist.destroy_process_group()


def test_depth_batched_sync_batchnorm_collectives_equal_one_per_layer(monkeypatch):
    """engine.Ctx.sync_stats: under SyncBatchNorm the same block position of HRNet's resolution branches and the terms of a fuse
    layer exchange their statistics -- and their backward sums -- in ONE collective (transformer/hr_base.py: BasicBlock.run_lockstep, _fuse_lockstep).
    One-rank RCCL group with the collectives forced, the trunk of tests/test_wiring_cpu.small_hrnet(deep=1) with two statistic groups (the batched source +
    driving pass): the depth-batched walk against one collective per layer and direction -- same kernels on the same data in another issue order, so the
    outputs and running statistics agree to rounding of the statistics atomics and every parameter gradient to what two runs of one form differ by; the
    collective count drops as the world-2 gloo test (test_sync_batchnorm_hrnet_world2_matches_big_batch) states, and the layer passes served stay the same."""
    import os
    import torch.distributed as dist
    from mrfa_amd import engine
    from mrfa_amd.utils.prng import det_uniform
    from tests.test_wiring_cpu import small_hrnet
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(36300 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1)
    monkeypatch.setattr(engine, "SYNCBN_FORCE", True)
    frames, b, size = 2, 2, 64
    x = torch.cat([(det_uniform(f"lk/x{i}", (b, 3, size, size)) * (1.0 + 0.5 * i)).to(DEV) for i in range(frames)], 0)
    w = det_uniform("lk/w", (frames * b, 32, size // 4, size // 4)).to(DEV)

    def run(lockstep):
        monkeypatch.setattr(engine, "SYNCBN_LOCKSTEP", lockstep)
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(small_hrnet(1)).to(DEV)
        m.train(True)
        c0, e0 = engine.SYNCBN_COLLECTIVES, engine.SYNCBN_EXCHANGES
        with engine.stat_groups(frames):
            y = m(x)
        (y * w).sum().backward()
        torch.cuda.synchronize()
        return (y.detach(), {n: p.grad.double().clone() for n, p in m.named_parameters()}, {n: v.clone() for n, v in m.named_buffers()},
                engine.SYNCBN_COLLECTIVES - c0, engine.SYNCBN_EXCHANGES - e0)
    try:
        y0, g0, b0, c0, e0 = run(False)
        y0b, g0b, _, _, _ = run(False)
        y1, g1, b1, c1, e1 = run(True)
    finally:
        torch.cuda.synchronize()
        if own_group:
            dist.destroy_process_group()
    n_bn = len([1 for n in b0 if n.endswith("running_mean")])
    assert c0 == e0 == e1 == 2 * n_bn, (c0, e0, e1, n_bn)
    assert c1 == 2 * (n_bn - 25), (c1, n_bn)               # (the count of the world-2 gloo test's deep case)
    assert (y0 - y1).abs().max().item() <= 1e-5 * max(1.0, y0.abs().max().item())
    for n in b0:
        if b0[n].dtype.is_floating_point:
            assert (b0[n] - b1[n]).abs().max().item() <= 1e-6 * max(1.0, b0[n].abs().max().item()), n
    worst = noise = 0.0
    for n in g0:
        sc = max(g0[n].abs().max().item(), 1e-6)
        worst = max(worst, (g0[n] - g1[n]).abs().max().item() / sc)
        noise = max(noise, (g0[n] - g0b[n]).abs().max().item() / sc)
    print(f"depth-batched SyncBatchNorm collectives: {c1} instead of {c0} per pass pair; worst per-parameter gradient difference {worst:.2e} of the parameter's scale "
          f"(two runs of the per-layer form: {noise:.2e})")
    assert worst <= 2e-4 + 4 * noise, (worst, noise)


def test_encoder_weight_gradients_dealt_onto_side_streams_match_the_inline_order():
    """HotPath._wdefer_enc (the keypoint encoder's ~200 weight-gradient launches collected during its backward chain and dealt onto
    four side streams afterwards, their un-packing last): one eager forward + backward with the fan-out on and off; a randomly initialised
    train-mode model amplifies summation-order noise to percents of the encoder's gradient, so the yardstick is the distance of two runs with the
    fan-out OFF -- the on/off distance of every sub-network must stay within 3 x that noise."""
    import bench
    from mrfa_amd import engine
    from mrfa_amd.train import VOX1, HotPath, make_optimizer, l1_loss
    src, drv = _pairs(2, "g/fan")

    def run(fanout):
        torch.manual_seed(0)
        m = HotPath(VOX1, prior="mtia")
        bench.init_weights(m)
        m.to(DEV).train(True)
        m.defer_decoder_wgrads = True
        m._wdefer_enc.fanout = fanout
        make_optimizer(m, fused=True)                         # flat gradient buffers: the direct-gradient mode the deferral needs
        for p in m.parameters():
            if p.grad is not None:
                p.grad.zero_()
        with engine.direct_param_grads():
            loss = l1_loss(m(src, drv), drv)
            loss.backward()
        m.join()
        torch.cuda.synchronize()
        assert (fanout > 1) == bool(m._wdefer_enc.fan), "the fan-out did not run / ran when it was off"
        groups = {}
        for n, p in m.named_parameters():
            if p.grad is not None:
                groups.setdefault(n.split(".")[0], []).append(p.grad.detach().flatten().double())
        return {k: torch.cat(v) for k, v in groups.items()}
    a, a2, b = run(0), run(0), run(4)
    for k in a:
        noise = float((a[k] - a2[k]).norm() / a[k].norm())
        onoff = float((a[k] - b[k]).norm() / a[k].norm())
        assert onoff <= 3.0 * noise + 1e-4, (k, onoff, noise)


def test_pack_stream_replays_like_the_eager_step():
    """GraphedTrainStep refreshes the non-encoder weight layouts on a side stream beside the keypoint encoder's forward.
    The schedule must replay to the eager step's gradients: verify() at the INITIAL weights (well conditioned: the band is ~1e-2 / 2e-3 / 1e-2 of
    the encoder / decoder / dense-motion gradient), and the stale-layout case -- weights changed between capture and replay -- must still be right,
    i.e. the packs really run inside the graph, before their first reader."""
    import bench
    from mrfa_amd.graph import GraphedTrainStep
    from mrfa_amd.train import VOX1, HotPath, make_optimizer
    src, drv = _pairs(2, "g/pk")
    m = HotPath(VOX1, prior="mtia")
    bench.init_weights(m)
    m.to(DEV).train(True)
    opt = make_optimizer(m, fused=True)
    step = GraphedTrainStep(m, opt, src, drv, clip=10.0, world=1)
    assert step.pack_stream is not None and step.packs_rest.n > 0 and step.packs.n > 0
    step.verify(replays=3)                                  # raises when a replay leaves the eager noise band
    # weights changed behind the capture (a decoder and an encoder convolution scaled in place; not an optimizer step, which would move this
    # randomly initialised model to an ill-conditioned point): graph A must re-pack them before their first reader -- verify() raises otherwise
    with torch.no_grad():
        for mod in (m.decoder, m.encoder):
            w = next(p for n, p in mod.named_parameters() if p.dim() == 4 and p.shape[2] == 3 and p.shape[0] >= 32)
            w.mul_(1.25)
    step.verify(replays=3)


@pytest.mark.parametrize("frames", [1, 2])
def test_prologue_fusion_equals_the_bn_act_path(frames, monkeypatch):
    """engine.PROLOGUE_FUSION (the BatchNorm-apply + ReLU between the two convolutions of a residual block as the second one's prologue: conv_lean.hip /
    wgrad_lean.hip with one prologue vector pair per statistic group, the BatchNorm's first backward phase in the data gradient that writes d(relu(bn(x))))
    against the default path (a bn_act launch and its tensor between the convolutions) on the HRNet trunk: same outputs, running statistics and parameter
    gradients up to summation order, with one statistic group and with two"""
    from mrfa_amd import engine
    from mrfa_amd.utils.prng import det_uniform
    from tests.test_wiring_cpu import small_hrnet
    from mrfa_amd import hip
    b, size = 1, 256                      # (the encoder's own maps: 64^2 x 32, 32^2 x 64, 16^2 x 128 -- the shapes conv_lean.hip / wgrad_lean.hip take)
    x = torch.cat([(det_uniform(f"pf/x{i}", (b, 3, size, size)) * (1.0 + 0.5 * i)).to(DEV) for i in range(frames)], 0)
    w = det_uniform("pf/w", (frames * b, 32, size // 4, size // 4)).to(DEV)
    prev = hip.lib().mrfa_set_tuning(b"conv_lean_min_wgs", 1)           # (two frames do not make the workgroup count the bench batch does)
    request_cleanup = lambda: hip.lib().mrfa_set_tuning(b"conv_lean_min_wgs", prev)

    def run(fused):
        monkeypatch.setattr(engine, "PROLOGUE_FUSION", fused)
        used = []
        real = engine.Ctx.prebn
        monkeypatch.setattr(engine.Ctx, "prebn", lambda self, *a, **k: (used.append(1), real(self, *a, **k))[1])
        m = small_hrnet().to(DEV)
        m.train(True)
        with engine.stat_groups(frames):
            y = m(x)
        (y * w).sum().backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(engine.Ctx, "prebn", real)
        return y.detach(), {n: p.grad.double().clone() for n, p in m.named_parameters()}, {n: v.clone() for n, v in m.named_buffers()}, len(used)
    try:
        y0, g0, b0, n0 = run(False)
        y0b, g0b, _, _ = run(False)
        y1, g1, b1, n1 = run(True)
    finally:
        request_cleanup()
    assert n0 == 0 and n1 > 0, "the fused run must take the prologue path in its residual blocks"
    assert (y0 - y1).abs().max().item() <= 1e-5 * max(1.0, y0.abs().max().item())
    for n in b0:
        if b0[n].dtype.is_floating_point:
            assert (b0[n] - b1[n]).abs().max().item() <= 1e-6 * max(1.0, b0[n].abs().max().item()), n
    worst = noise = 0.0
    for n in g0:
        sc = max(g0[n].abs().max().item(), 1e-6)
        worst = max(worst, (g0[n] - g1[n]).abs().max().item() / sc)
        noise = max(noise, (g0[n] - g0b[n]).abs().max().item() / sc)
    print(f"prologue fusion vs bn_act path: worst per-parameter gradient error {worst:.2e} of the parameter's scale (two default runs: {noise:.2e})")
    assert worst <= 2e-4 + 4 * noise, (worst, noise)


@pytest.mark.parametrize("frames,b", [(2, 2), (3, 2), (2, 8)])
def test_statistic_groups_on_the_gpu_equal_separate_calls_sharply(frames, b):
    """The batched pass against the separate calls where summation-order noise is small enough to see a SYSTEMATIC error: the HRNet trunk with one
    block per branch (tests/test_wiring_cpu.small_hrnet: every BatchNorm call pattern of the encoder -- conv-epilogue statistics with the fused finalize,
    stride-2 layers, residual-closing BatchNorm, the first backward phase inside the consumer's data gradient, the per-group statistics pass behind
    launches whose tiles would straddle two groups) on 32 x 32 and 64 x 64 inputs.  Outputs to 1e-5, running statistics to 1e-6, EVERY parameter gradient
    to 2e-4 of its scale on 32 x 32 (the random-init TokenPose_B of test_batched_encoder_pass_equals_separate_calls amplifies noise to percents and can
    only bound).

    At 8 x 64 x 64 the trunk itself is ill-conditioned -- the CPU fp32 oracle sits 5.7e-3 from its own fp64 run in layer1.3.bn3.bias, a handful
    of ReLU decisions on nearly-dead channels -- and the two forms take different kernels where a launch is small (the separate call of 8 samples splits
    K, the batch of 16 does not), so those decisions need not coincide.  There the gate is the fp64 oracle (oracle/tokenpose_oracle.hrnet, autograd, run
    here on the host): the batched pass may be no further from it than 2e-4 + 2 x what the SEPARATE calls are (measured: both 1.4e-2 .. 1.6e-2, two
    separate-call runs of one process 2e-6 or 1.4e-2 apart depending on the box).  This case is a bound; the sharp statements are the 32 x 32 cases above,
    the forward outputs and running statistics of every case, and the grouped kernel cases of tests/test_kernels_gpu.py at these shapes."""
    from mrfa_amd import engine
    from mrfa_amd.utils.prng import det_uniform
    from tests.test_wiring_cpu import small_hrnet
    size = 32 if b == 2 else 64
    xs = [(det_uniform(f"sgg/x{i}", (b, 3, size, size)) * (1.0 + 0.5 * i) + 0.1 * i).to(DEV) for i in range(frames)]
    ws = [det_uniform(f"sgg/w{i}", (b, 32, size // 4, size // 4)).to(DEV) for i in range(frames)]

    def run(batched):
        m = small_hrnet().to(DEV)
        m.train(True)
        if batched:
            with engine.stat_groups(frames):
                y = m(torch.cat(xs, 0))
            (y * torch.cat(ws, 0)).sum().backward()
            ys = list(y.detach().split(b))
        else:
            ys = [m(x) for x in xs]
            sum((yy * w).sum() for yy, w in zip(ys, ws)).backward()
            ys = [yy.detach() for yy in ys]
        torch.cuda.synchronize()
        return ys, {n: p.grad.double().clone() for n, p in m.named_parameters()}, {n: v.clone() for n, v in m.named_buffers()}
    y0, g0, b0 = run(False)
    y0b, g0b, _ = run(False)
    y1, g1, b1 = run(True)
    for a, c in zip(y0, y1):
        assert (a - c).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())
    for n in b0:
        if b0[n].dtype.is_floating_point:
            assert (b0[n] - b1[n]).abs().max().item() <= 1e-6 * max(1.0, b0[n].abs().max().item()), n
        else:
            assert int(b0[n]) == int(b1[n]) == frames, n
    errs, noise = [], 0.0
    for n in g0:
        sc = max(g0[n].abs().max().item(), 1e-6)
        errs.append((g0[n] - g1[n]).abs().max().item() / sc)
        noise = max(noise, (g0[n] - g0b[n]).abs().max().item() / sc)
    worst, median = max(errs), sorted(errs)[len(errs) // 2]
    print(f"batched vs separate calls: worst per-parameter gradient error {worst:.2e} of the parameter's scale, median {median:.2e} "
          f"(two separate-call runs: {noise:.2e})")
    if size == 32:
        assert worst <= 2e-4 + 4 * noise, (worst, noise)
        return
    from oracle import tokenpose_oracle as TO
    P = {"h." + k: (v.detach().double().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
         for k, v in small_hrnet().state_dict().items()}
    sum((TO.hrnet(x.cpu().double(), P, "h", True) * w.cpu().double()).sum() for x, w in zip(xs, ws)).backward()
    far = lambda g: max((g[n].cpu() - P["h." + n].grad).abs().max().item() / max(P["h." + n].grad.abs().max().item(), 1e-6) for n in g)
    sep, bat = far(g0), far(g1)
    print(f"distance from the fp64 oracle: separate calls {sep:.2e}, batched pass {bat:.2e}")
    assert bat <= 2e-4 + 2.0 * sep, (bat, sep)
