"""hipGraph capture of the hot path.

A training step of the path is ~2 900 kernel launches (the reference issues more: one ATen op per layer op); issued
one by one from Python they cost 100-130 ms of host time, which is the same order as the 135 ms of device time -- on a
slow host core the step becomes launch-bound.  The programs the engine runs are static (fixed shapes, no
data-dependent control flow, no host synchronisation), so a whole step is captured ONCE into hipGraphs
(torch.cuda.CUDAGraph is the HIP graph object on ROCm; torch is only the owner of the capture stream and of the graph's
private memory pool) and replayed with one launch per step:

  graph A   zero the flat gradient buffer, re-pack the weights, forward, loss, backward (every kernel of mrfa_hip.h)
  graph B   inf-norm clipping of the encoder / dense_motion gradients + Adam (reference train.py:21-25, 58-70)
With N > 1 ranks the exchange of the data-parallel path -- the only collective on it -- is ONE RCCL all-reduce of the flat gradient
buffer between graph A and graph B (457 MB; ~2.6 ms of a 113 ms step at N=8 by RCCL's bus bandwidth, not overlapped).  Opt-in
(`overlap_exchange=True` / MRFA_OVERLAP_EXCHANGE=1, see GraphedTrainStep.__init__ for why it is not the default): graph A is cut where
the backward reaches the keypoint encoder (mrfa_amd.train.SplitBackward), and the exchange is issued between the pieces:
  graph A1  ... forward, loss, backward of the decoder / dense motion (/ background) networks
  (eager)   async RCCL all-reduce of their gradient ranges of the flat buffer (408 of 457 MB with the MTIA prior), on RCCL's stream
  graph A2  the encoder's backward (~30 ms), which hides that all-reduce
  (eager)   all-reduce of the encoder's range (49 MB), wait for both
  graph B   1/N, clipping, Adam
(the reference overlaps through DistributedDataParallel's buckets, train.py:45-48; here the flat buffer has two buckets in backward order)

`GraphedForward` is the inference counterpart (one graph, weights packed once outside it).

MEMCPY / MEMSET NODES.  With ROCm 7.2's default "graph packet capture" path (kernel nodes pre-recorded as AQL packets)
a hipMemcpyAsync / hipMemsetAsync recorded by stream capture becomes a memcpy / memset node that is NOT reliably
ordered against the kernel nodes around it when the graph is replayed: the first replay is right, later ones read a
source before its producer ran (observed: the 80-byte clone of d(loss)/d(keypoints) feeding the KPDetector backward, and
the 4-byte semaphore memset of torch's multi-workgroup reductions -- the loss then reads 0.0).  Two defences:
  * mrfa_amd sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the HIP runtime initialises (mrfa_amd/__init__.py; same
    replay speed on this GPU-bound path, replays then agree with eager launches -- measured with a since-removed bisection script);
  * what the engine itself records is kernels only (no memcpy clones in IslandOut.add_grad, train.l1_loss reduces
    without a semaphore memset, zero fills are fill kernels); torch's autograd inside the glue islands still emits a
    few 8-byte memsets, which is why the first defence is needed;
and GraphedTrainStep.verify() replays the graph against itself so a mis-ordered graph is detected, not trusted.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn

from . import engine, graph_replay_safe


def l1_loss(gen, driving):
    from .train import l1_loss as f
    return f(gen, driving)


def _increment_version(tensors):
    """kernels and graph replays write parameters behind autograd's back: bump the version counters so that every
    cache keyed on them (the engine's packed-weight cache) sees the change"""
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is not None:
        inc(tensors)
    else:
        for t in tensors:
            torch._C._increment_version(t)


class FlatGradients:
    """Every parameter's .grad as a view of ONE fp32 buffer (each slice 16-byte aligned): one memset instead of ~700
    zero fills, and the data-parallel exchange of the path is ONE all-reduce (RCCL on the GPU, gloo in the CPU tests)
    of the whole buffer instead of DistributedDataParallel's buckets.  autograd accumulates into existing .grad tensors
    in place, so the views survive backward passes."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.total = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        assert self.flat.data_ptr() % 16 == 0

    def bind(self):
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += (p.numel() + 3) // 4 * 4

    def bound(self) -> bool:
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.total
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params)

    def ranges_of(self, params):
        """merged [lo, hi) element ranges of the flat buffer that hold the gradients of `params`"""
        want = {id(p) for p in params}
        out, off = [], 0
        for p in self.params:
            n = (p.numel() + 3) // 4 * 4
            if id(p) in want:
                if out and out[-1][1] == off:
                    out[-1][1] = off + n
                else:
                    out.append([off, off + n])
            off += n
        return [tuple(r) for r in out]

    def all_reduce(self, ranges=None, async_op: bool = False):
        """sum over ranks (the mean's 1/world is applied by the caller: GraphedTrainStep folds it into graph B), of the whole buffer
        or of the given element ranges (contiguous views: in place); async_op=True returns the work handles"""
        views = [self.flat] if ranges is None else [self.flat[lo:hi] for lo, hi in ranges if hi > lo]
        handles = [torch.distributed.all_reduce(v, async_op=async_op) for v in views]
        return handles if async_op else None


class GraphedForward:
    """Replay of `model(source, driving)` (eval mode, no autograd) as one hipGraph.

    Weights are packed once before capture: call `recapture()` after changing them."""

    def __init__(self, model: nn.Module, source: torch.Tensor, driving: torch.Tensor):
        self.model = model
        self.src = source.clone()
        self.drv = driving.clone()
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out = None
        self.recapture()

    def recapture(self, check: bool = True):
        graph_replay_safe("GraphedForward")
        m = self.model
        was_training = m.training
        m.eval()
        try:
            with torch.no_grad():
                eager = m(self.src, self.drv)           # packs weights / builds gather tables outside the graph
                eager = tuple(t.clone() for t in eager) if isinstance(eager, (tuple, list)) else eager.clone()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.out = m(self.src, self.drv)
            self.graph = g
            if check:
                self._self_check(eager)
        finally:
            m.train(was_training)

    def _self_check(self, eager, replays: int = 3, tol: float = 5e-3, mean_tol: float = 2e-5):
        """the first AND later replays against the eager output at the same inputs (a mis-ordered memcpy / memset node shows from
        the second replay on).  Inference has no batch statistics; the only legitimate differences are fp32 summation order
        (split-K atomics): ~1e-6 typically, ~1e-4 on single border pixels of sharply warped frames, whereas a mis-ordered graph is
        wrong everywhere -- so the mean is gated tightly (`mean_tol`) and the max loosely (`tol`), relative to max |output|"""
        outs = self.out if isinstance(self.out, (tuple, list)) else (self.out,)
        refs = eager if isinstance(eager, (tuple, list)) else (eager,)
        scratch = torch.empty(1 << 20, device=self.src.device)
        for k in range(replays):
            scratch.normal_()                           # unrelated device work between replays
            self.graph.replay()
            torch.cuda.synchronize()
            for i, (o, r) in enumerate(zip(outs, refs)):
                if not torch.is_tensor(o):
                    continue
                diff = (o - r).abs()
                d, dm, sc = float(diff.max()), float(diff.mean()), max(1.0, float(r.abs().max()))
                if not (d <= tol * sc and dm <= mean_tol * sc):
                    raise RuntimeError(f"GraphedForward: replay {k} differs from the eager forward (output {i}: max |diff| {d:.3e}, mean "
                                       f"{dm:.3e}); the captured graph is not trustworthy (see mrfa_amd/graph.py, MEMCPY / MEMSET NODES)")

    def __call__(self, source: torch.Tensor, driving: torch.Tensor):
        if source.data_ptr() != self.src.data_ptr():
            self.src.copy_(source)
        if driving.data_ptr() != self.drv.data_ptr():
            self.drv.copy_(driving)
        self.graph.replay()
        return self.out


def _drain_watchdog(seconds: float = 0.3):
    """before a capture that will hold RCCL collectives: wait until the device is idle and ProcessGroupNCCL's watchdog thread (one pass per 100 ms) has
    retired every eager collective issued so far -- see GraphedTrainStep.__init__"""
    import time
    torch.cuda.synchronize()
    time.sleep(seconds)


class GraphedTrainStep:
    """fwd + bwd + (all-reduce) + clip + Adam of mrfa_amd.train.train_step as two hipGraphs (see module docstring).

    `model` is the bare HotPath (NOT wrapped in DistributedDataParallel: the gradient exchange is the single flat
    all-reduce issued here); `optimizer` must have been built with capturable=True and stepped at least once."""
    _captures = 0

    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, source: torch.Tensor, driving: torch.Tensor,
                 clip: float = 10.0, world: int = 1, exchange: Optional[bool] = None, overlap_wgrad: bool = False,
                 loss_fn=None, overlap_exchange: Optional[bool] = None,
                 defer_wgrads: Optional[bool] = None):
        """loss_fn(model, source, driving) -> scalar loss; None = the surrogate mean|model(source, driving) - driving|.
        overlap_exchange: cut graph A at the encoder boundary and overlap the all-reduce with the encoder's backward (surrogate-loss step
        of a HotPath only).  OPT-IN (argument or MRFA_OVERLAP_EXCHANGE=1): measured on one MI355X with a one-rank RCCL group, cutting the
        graph costs nothing (111.7 vs 112.3 ms) and the two all-reduces issued serially cost nothing (112.6), but the 408 MB all-reduce
        launched beside graph A2 costs +4.5 ms (116.1) although it moves no data -- and no multi-GPU box was available to show that the
        overlap wins against that at N > 1, where the serial all-reduce is ~2.6 ms at N=8 (2 % of the step).  Default: ONE all-reduce."""
        from .train import SplitBackward, exchange_ranges
        graph_replay_safe("GraphedTrainStep")
        self.model, self.opt, self.clip, self.world = model, optimizer, clip, world
        self.loss_fn = loss_fn
        syncbn_collectives = any(isinstance(m, nn.SyncBatchNorm) for m in model.modules()) and (world > 1 or engine.SYNCBN_FORCE)
        # SyncBatchNorm's statistics collectives are captured into the graph (RCCL ops are capturable): every rank must enqueue them in ONE order on
        # the communicator, i.e. on one stream: no branch lanes inside the encoder program (engine.Ctx.lanes), and the encoder's weight gradients stay in
        # line (measured on one rank with the forced collective: 113 -> 123 ms with the fan-out)
        self.no_lanes = syncbn_collectives
        if syncbn_collectives and hasattr(model, "defer_encoder_wgrads"):
            model.defer_encoder_wgrads = False
        self.exchange = (world > 1) if exchange is None else exchange      # all-reduce between the graphs
        if overlap_exchange is None:
            env = os.environ.get("MRFA_OVERLAP_EXCHANGE", "0")     # "force": cut the graph even without an exchange (timing the cut alone)
            overlap_exchange = env == "force" or (self.exchange and env == "1")
        self.split = SplitBackward(model) if (overlap_exchange and SplitBackward.supported(model, loss_fn)) else None
        # decoder / dense-motion weight gradients issued beside the encoder's backward (engine.DeferredWgrads): MTIA prior, direct
        # parameter gradients (FlatAdam); not together with the cut graph, whose first all-reduce needs those gradients final.
        # Measured (same box, 20 steps): 108.6 vs 111.2 ms.  (Giving the capture / encoder streams high priority on top of it made the
        # replay 1.7x SLOWER -- 182 ms -- so stream priorities are left alone.)
        if defer_wgrads is None:
            defer_wgrads = (getattr(model, "prior", "") == "mtia" and os.environ.get("MRFA_DEFER_WGRADS", "1") == "1")
        if hasattr(model, "defer_decoder_wgrads"):
            model.defer_decoder_wgrads = bool(defer_wgrads) and self.split is None and getattr(optimizer, "fused_clip", False)
        self.src, self.drv = source.clone(), driving.clone()
        self.fused = getattr(optimizer, "fused_clip", False)      # mrfa_amd.optim.FlatAdam: owns the flat buffers
        self.grads = optimizer.grads if self.fused else FlatGradients(model.parameters())
        ps, dev = self.grads.params, self.grads.flat.device
        self.flat = self.grads.flat
        # capture stream: autograd's AccumulateGrad nodes remember the stream they were created on, so one eager
        # fwd+bwd is issued on the capture stream first (no optimizer step: the weights are left untouched)
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for p in ps:
                p.grad = None
            saved = [b.clone() for b in model.buffers()]            # BN running statistics: this pass must not count
            engine.WGRAD_STREAM = overlap_wgrad
            try:
                with engine.direct_param_grads():
                    self._head()
                    self._tail()
            finally:
                engine.WGRAD_STREAM = False
            for b, sv in zip(model.buffers(), saved):
                b.copy_(sv)
            self.grads.bind()
        self.head_ranges, self.tail_ranges = exchange_ranges(self.grads, model) if self.split is not None else (None, None)
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize()
        # every layout the warm pass packed, in one launch per 48 convolutions.  The keypoint encoder's (a tenth of the bytes) first; the rest on a side
        # stream beside the encoder's forward chain, whose small kernels leave the HBM idle (HotPath.await_packs() joins it before the first other use)
        enc = getattr(model, "encoder", None)
        self.pack_stream = None
        if enc is not None and hasattr(model, "await_packs") and os.environ.get("MRFA_PACK_STREAM", "1") != "0":
            self.packs = engine.PackPlan(model, only=enc)
            self.packs_rest = engine.PackPlan(model, exclude=enc)
            self.pack_stream = torch.cuda.Stream(device=dev)
        else:
            self.packs = engine.PackPlan(model)
            self.packs_rest = None
        # All three graphs are captured in "thread_local" error mode: ProcessGroupNCCL's watchdog thread queries events of the eager warm pass's RCCL work
        # while the capture runs, and in the default ("global") mode such a query from ANOTHER thread invalidated the capture (hipErrorCapturedEvent: 1 capture
        # in 5 with the SyncBatchNorm collectives captured, round 5 -- then papered over with a 0.5 s pause that protected the first graph only).
        # thread_local is the mode PyTorch itself uses around NCCL: only this thread's calls are checked against the capture.
        # That alone left 1 capture in 3 failing at the end of round 6 (hipErrorStreamCaptureInvalidated in BOTH collective forms): RCCL's stream joins the
        # capture at the first captured collective, and HIP then refuses the watchdog's query of an EAGER work's end event on that stream
        # (hipErrorCapturedEvent) although the event was recorded before the capture began.  So the watchdog is also given the time to retire the eager work
        # first: device idle, then three of its 100 ms polling periods, before each capture (_drain_watchdog).
        drain = (lambda: _drain_watchdog()) if (syncbn_collectives or world > 1) else (lambda: None)
        drain()
        self.g_fb = torch.cuda.CUDAGraph()
        syncbn0, syncex0 = engine.SYNCBN_COLLECTIVES, engine.SYNCBN_EXCHANGES       # (the Python counters run while the step is captured: the collectives of ONE step)
        GraphedTrainStep._captures += 1
        engine.CAPTURE_KEY = GraphedTrainStep._captures
        engine.WGRAD_STREAM = overlap_wgrad
        try:
            with torch.cuda.graph(self.g_fb, stream=self.stream, capture_error_mode="thread_local"), engine.direct_param_grads():
                if engine.MARKS is not None:
                    engine.MARKS.begin()
                engine.mark("step: start")
                self.flat.zero_()
                self.packs.run()
                if self.packs_rest is not None:
                    # (round 6: forking this refresh BEHIND the encoder's stem and layer1 -- which stream 67 MB tensors and take 0.2 + 0.9 ms alone, 1.2 + 1.2 ms
                    # beside it -- moved the same 1.3 ms of HBM contention onto stage 2: 83.6 / 82.2 ms against 83.2 / 81.4, profiles/r6_ab_late_pack.txt; removed)
                    self.pack_stream.wait_stream(self.stream)
                    with torch.cuda.stream(self.pack_stream):
                        self.packs_rest.run()
                        engine.mark("packed (rest)")
                    object.__setattr__(model, "_pack_stream", self.pack_stream)
                engine.mark("packed")
                loss, gen = self._head()
                if self.split is None:
                    self._tail()
                # detached handles: a static output that still referenced its autograd graph would keep the graph (and
                # the parameters' AccumulateGrad nodes bound to the capture stream) alive for the life of this object
                self.loss, self.gen = loss.detach(), (gen.detach() if gen is not None else None)
            del loss, gen
            self.g_tail = None
            if self.split is not None:                                # graph A2: the encoder's backward, from the keypoint gradients of A1
                self.g_tail = torch.cuda.CUDAGraph()
                drain()
                with torch.cuda.graph(self.g_tail, pool=self.g_fb.pool(), stream=self.stream, capture_error_mode="thread_local"), engine.direct_param_grads():
                    self._tail()
        finally:
            engine.CAPTURE_KEY = 0
            engine.WGRAD_STREAM = False
        self.syncbn_collectives = engine.SYNCBN_COLLECTIVES - syncbn0
        self.syncbn_exchanges = engine.SYNCBN_EXCHANGES - syncex0       # = the collectives of the one-per-layer form (engine.SYNCBN_LOCKSTEP off)
        assert self.grads.bound(), "a gradient left the flat buffer"
        self.g_opt = torch.cuda.CUDAGraph()
        if self.fused:
            optimizer.grad_scale = 1.0 / world
            optimizer.sync_lr()
        drain()
        with torch.cuda.graph(self.g_opt, pool=self.g_fb.pool(), stream=self.stream, capture_error_mode="thread_local"):
            engine.mark("optimizer: start")
            if self.fused:
                optimizer.step()                                      # 1/world, clipping and Adam: 6 launches
                engine.mark("optimizer: done")
            else:
                if world > 1:
                    self.flat.mul_(1.0 / world)
                if clip:
                    nn.utils.clip_grad_norm_(model.encoder.parameters(), max_norm=clip, norm_type=math.inf)
                    nn.utils.clip_grad_norm_(model.dense_motion.parameters(), max_norm=clip, norm_type=math.inf)
                    if getattr(model, "bg_predictor", None) is not None:
                        nn.utils.clip_grad_norm_(model.bg_predictor.parameters(), max_norm=clip, norm_type=math.inf)
                optimizer.step()

    def _loss(self):
        if self.loss_fn is not None:
            return self.loss_fn(self.model, self.src, self.drv), None
        gen = self.model(self.src, self.drv)
        return l1_loss(gen, self.drv), gen

    def _join(self):
        if hasattr(self.model, "join"):
            self.model.join()

    def _head(self):
        """forward + loss + backward: all of it, or (overlapped exchange) down to the keypoint encoder's outputs"""
        lanes, engine.BRANCH_STREAMS = engine.BRANCH_STREAMS, engine.BRANCH_STREAMS and not self.no_lanes
        try:
            if self.split is not None:
                return self.split.head(self.src, self.drv)
            loss, gen = self._loss()
            loss.backward()
            return loss, gen
        finally:
            engine.BRANCH_STREAMS = lanes

    def _tail(self):
        if self.split is not None:
            self.split.tail()                                         # encoder backward + join of the side streams
        else:
            self._join()

    def _replay_fwd_bwd(self):
        self.g_fb.replay()
        if self.g_tail is not None:
            self.g_tail.replay()

    def _segments(self):
        """flat-buffer segments to compare separately: the optimizer's parameter groups (encoder / decoder / dense_motion have
        very different conditioning) or, without FlatAdam, the whole buffer"""
        return list(self.opt.segments) if self.fused else [(0, self.flat.numel())]

    def verify(self, replays: int = 3, tol: float = 0.5, band_mult: float = 4.0, loss_tol: float = 1e-4) -> float:
        """Self-check after capture.  Graph A is replayed several times at fixed weights and every replay's gradient is
        compared, per parameter group, with replay 0 and with an eager forward+backward at the same weights.  Identical
        inputs must give identical results up to atomic-order noise, so the allowed relative L2 distance of a group is
        `band_mult` x the largest distance between three EAGER passes (the noise band, measured here: a randomly initialised model in
        train mode amplifies summation-order noise into 0.2 % of the decoder's gradient but into tens of percent of the
        keypoint encoder's after the first Adam steps) + `tol`/25; the best-conditioned group is therefore checked to a few
        percent, and a gradient that is zero or stale (distance >= 1) fails wherever the band is below 1/band_mult.  The
        loss must agree to `loss_tol` (1e-4; plain-bf16 arithmetic amplifies summation-order noise to ~1e-3).  Returns the worst distance seen in the best-conditioned group.  A replay that depends on
        what ran before it means a node of the graph is not ordered (see "kernel nodes only" above)."""
        saved = [b.detach().clone() for b in self.model.buffers()]
        segs = self._segments()
        dev = self.flat.device
        rng_state = torch.cuda.get_rng_state(dev) if self.loss_fn is not None else None      # reseed() below must not leak out

        def dist(a, b):
            return [float((a[lo:hi] - b[lo:hi]).norm() / (b[lo:hi].norm() + 1e-30)) for lo, hi in segs]

        def reseed():
            # a loss with random draws (the equivariance transform of the reference's objective): the same device-generator state
            # before every eager pass and every replay, so that all of them see the same draw
            if self.loss_fn is not None:
                torch.cuda.manual_seed(20261002)

        def eager():
            reseed()
            with torch.cuda.stream(self.stream):
                self.flat.zero_()
                with engine.direct_param_grads():
                    loss = self._head()[0]
                    self._tail()
                loss = float(loss.detach())
            torch.cuda.synchronize()
            return self.flat.double().cpu(), loss

        torch.cuda.synchronize()
        e0, eloss = eager()
        e1, _ = eager()
        e2, _ = eager()
        band = [max(t) for t in zip(dist(e1, e0), dist(e2, e0), dist(e2, e1))]      # three samples of a heavy-tailed quantity
        allow = [band_mult * b + tol / 25.0 for b in band]
        best = min(range(len(segs)), key=lambda k: band[k])
        ref = ref_loss = None
        worst = 0.0
        with torch.no_grad():
            scratch = torch.empty_like(self.flat)
            for k in range(replays):
                scratch.normal_()                       # unrelated device work between replays
                float(scratch.sum())
                reseed()
                self._replay_fwd_bwd()
                torch.cuda.synchronize()
                g, loss = self.flat.double().cpu(), float(self.loss)
                if abs(loss - eloss) > loss_tol * max(1.0, abs(eloss)):
                    raise RuntimeError(f"hipGraph replay {k}: loss {loss} differs from the eager pass {eloss}")
                checks = [("the eager pass", dist(g, e0))]
                if ref is None:
                    ref, ref_loss = g, loss
                else:
                    checks.append(("replay 0", dist(g, ref)))
                for what, d in checks:
                    worst = max(worst, d[best])
                    bad = [(i, x, a) for i, (x, a) in enumerate(zip(d, allow)) if not x <= a]
                    if bad:
                        raise RuntimeError(f"hipGraph replay {k} differs from {what}: per-group gradient distances {d} exceed "
                                           f"the allowed {allow} (eager noise band {band})")
            for b, sv in zip(self.model.buffers(), saved):
                b.copy_(sv)
        if rng_state is not None:
            torch.cuda.set_rng_state(rng_state, dev)      # every rank / run continues from ITS generator state, not from the fixed seed
        # for the record (bench.py prints it): the eager noise band and what was allowed, per parameter group
        self.last_verify = {"eager_band": [float(f"{x:.3e}") for x in band], "allowed": [float(f"{x:.3e}") for x in allow],
                            "worst_in_best_group": float(f"{worst:.3e}"), "replays": replays}
        return worst

    def __call__(self, source: torch.Tensor, driving: torch.Tensor) -> torch.Tensor:
        if source.data_ptr() != self.src.data_ptr():
            self.src.copy_(source)
        if driving.data_ptr() != self.drv.data_ptr():
            self.drv.copy_(driving)
        if self.fused:
            self.opt.sync_lr()                                        # an LR scheduler may have edited param_groups
        self.g_fb.replay()
        if self.g_tail is not None:
            # RCCL's stream waits for A1 (the work is enqueued behind the current stream), A2 runs beside it on the current stream
            handles = self.grads.all_reduce(self.head_ranges, async_op=True) if self.exchange else []
            self.g_tail.replay()
            if self.exchange:
                handles += self.grads.all_reduce(self.tail_ranges, async_op=True)
            for h in handles:
                h.wait()
        elif self.exchange:
            self.grads.all_reduce()
        self.g_opt.replay()
        # the replay changed the weights behind autograd's back: bump the version counters, on which the engine's
        # packed-weight cache is keyed, so that an eager forward after this step re-packs
        _increment_version(self.grads.params)
        return self.loss
