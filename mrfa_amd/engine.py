"""Host-side execution engine for the HIP hot path.

MI355X-first design: instead of a torch autograd graph of ~1000 small ATen nodes per step, each module of the path
(KPDetector, DenseMotionNetwork, RaftFlow) runs as ONE program of explicit kernel launches on the current HIP stream.
Activations are fp32 NHWC `View`s (pointer + leading dimension) so producers write straight into slices of
concatenated buffers; the forward pass records a tape of backward closures (hand-written HIP backward kernels),
replayed in reverse by a single torch.autograd.Function per module, which is what makes `loss.backward()`, DDP
gradient hooks and optimizers work unchanged on top.

No CPU path exists here: every op goes through libmrfa_hip.so (mrfa_amd.hip.lib()).
"""
from __future__ import annotations

import contextlib
import os

import ctypes as C
from typing import Callable, List, Optional, Sequence

import torch

from . import hip

BN_EPS = 1e-5
FORCE_TILE = 0          # tuning hook: non-zero forces mrfa_conv_params.tile for every MFMA conv launch
BN_MOMENTUM = 0.1
# non-zero while mrfa_amd.graph captures a TRAINING step into a hipGraph: part of every weight-pack cache key, so that
# each pack kernel is recorded (once) inside the graph and re-runs on replay, after the optimizer changed the weights
CAPTURE_KEY = 0
# True (see direct_param_grads()): parameter gradients are accumulated by the backward kernels straight into existing
# .grad tensors (the flat gradient buffer of mrfa_amd.graph / mrfa_amd.optim) and autograd is handed None for them: no
# per-parameter staging tensor, zero fill and AccumulateGrad add (~900 small launches per step).  Gradient hooks do not
# fire in this mode, so DistributedDataParallel must not be wrapped around the model (the flat all-reduce replaces it).
DIRECT_PARAM_GRADS = False
# True: weight-gradient kernels are issued on a second HIP stream (forked after dY is final, joined at the end of each
# program's backward).  The weight gradients are a side chain of the backward pass -- nothing downstream reads them before
# the optimizer -- so the ~130 wgrad launches of a step can overlap the data-gradient / BN / elementwise chain; captured
# into a hipGraph this becomes a parallel branch.  Correct (tests/test_graph_gpu.py passes with it) but measured SLOWER on
# the benchmark step (97.0 vs 93.3 ms: the big kernels already fill the chip and then compete for L2), so it is off by default.
WGRAD_STREAM = False


class Marks:
    """Measurement aid (tools/step_phases.py): device timestamps at named points of a step, stored by one-thread kernels (mrfa_timestamp) on
    whatever stream reaches the point, so they are captured into the hipGraph and a REPLAYED step reports its own phase times without a profiler
    serialising its branches.  engine.MARKS = Marks(dev) switches it on; None (default): mark() does nothing."""

    def __init__(self, dev, n: int = 256):
        self.buf = torch.zeros(n, dtype=torch.int64, device=dev)
        self.names: List[str] = []

    def begin(self):
        self.names = []

    def read(self):
        """[(name, microseconds since the first mark)] of the last pass"""
        t = self.buf[:len(self.names)].cpu().tolist()
        return [(n, (v - t[0]) / 100.0) for n, v in zip(self.names, t)]


MARKS: Optional[Marks] = None


def mark(name: str):
    m = MARKS
    if m is None or len(m.names) >= m.buf.numel():
        return
    hip.check(hip.lib().mrfa_timestamp(hip.stream_ptr(), m.buf.data_ptr() + 8 * len(m.names)), "timestamp")
    m.names.append(name)


class DeferredWgrads:
    """The weight-gradient launches of the programs recorded under `defer_wgrads(d)` (dense motion + RaftFlow: ~100 launches, 22 ms of
    kernels that fill the chip) are not issued where the backward tape reaches them but collected, and issued on ONE side stream when
    the backward pass arrives at the next program that is NOT deferring -- the keypoint encoder, ~2 x 2 000 small latency-bound kernels
    that leave most of the GPU idle for ~20 ms.  Nothing downstream reads a weight gradient before the optimizer, so the only ordering
    needed is: after their dY (they are issued after the whole decoder backward), before `join()` (HotPath.join(), called after
    backward() by train_step / GraphedTrainStep).  The thunks hold the activations and gradient buffers they read until join().
    Direct-gradient mode only (DIRECT_PARAM_GRADS: the un-packing into .grad is deferred with them); anything else runs in line."""

    def __init__(self, fanout: int = 1, manual: bool = False, batch: bool = False):
        # manual: flushed only by join() (never by the backward of the next program): the keypoint encoder's collection -- its programs run their
        # backward on several streams, and only HotPath.join() has ordered all of them before the stream the flush forks from
        self.manual = manual
        self.thunks: List[Callable[[], None]] = []
        self.finals: List[Callable[[], None]] = []          # run after every thunk (the un-packing of the accumulators they add into)
        self.kept: Optional[List[Callable[[], None]]] = None
        self.stream: Optional["torch.cuda.Stream"] = None
        self.fan: List["torch.cuda.Stream"] = []
        # > 1: the launches are dealt round-robin onto that many side streams (the keypoint encoder's ~400 small weight-gradient launches,
        # independent of each other, issued after its backward chain instead of inside it); the finals wait for all of them
        self.fanout = max(1, int(fanout))
        self.flushed = False
        self.batch = batch                                   # plain weight-gradient launches are issued as parameter arrays (mrfa_conv2d_wgrad_multi)
        self._rr = 0
        self._used: List["torch.cuda.Stream"] = []           # side streams that received launches since the last join
        self._srcs: List["torch.cuda.Stream"] = []           # streams on which the pending launches were collected
    def add(self, fn: Callable[[], None], final: bool = False):
        (self.finals if final else self.thunks).append(fn)
        if torch.cuda.is_available():
            # the launch reads what the CURRENT stream has produced (dY of this layer): a program may run part of its backward on a branch
            # stream (Ctx.branch), so the side streams wait for every stream that contributed, not only for the one that flushes
            st = torch.cuda.current_stream()
            if st not in self._srcs:
                self._srcs.append(st)

    def reset(self):
        """start of a step: drop whatever an aborted previous step left behind (a backward that raised between the decoder and the join
        would otherwise have its stale weight-gradient / un-pack thunks flushed into the NEXT step's freshly zeroed .grad)"""
        if self.flushed and self.stream is not None:
            cur = torch.cuda.current_stream(self.stream.device)
            for st in self._used:
                cur.wait_stream(st)                          # launches already issued finish first
        self.thunks, self.finals, self.kept, self.flushed, self._used, self._srcs = [], [], None, False, [], []
        if self in _PENDING_DEFERRED:
            _PENDING_DEFERRED.remove(self)

    def _lanes(self, dev):
        if self.stream is None or self.stream.device != dev:
            self.stream = torch.cuda.Stream(device=dev)
            self.fan = []
        while len(self.fan) < self.fanout - 1:
            self.fan.append(torch.cuda.Stream(device=dev))
        return [self.stream] + self.fan[:self.fanout - 1]

    def _deal(self, dev: torch.device):
        """issue the collected launches on the side streams, ordered after the current stream; the streams are joined by flush()"""
        if not self.thunks:
            return
        if dev.type != "cuda":                               # (the ABI emulator: no streams; the same marshalling as below, so that it is tested without a GPU)
            todo = self.thunks
            if self.batch:
                todo = [fn for fn in self.thunks if getattr(fn, "params", None) is None]
                self._issue_plain([fn for fn in self.thunks if getattr(fn, "params", None) is not None], dev)
            for fn in todo:
                fn()
            self.thunks = []
            return
        cur = torch.cuda.current_stream(dev)
        lanes = self._lanes(dev) if len(self.thunks) > self.fanout else self._lanes(dev)[:1]
        for st in lanes:
            st.wait_stream(cur)
            for src in self._srcs:
                if src != cur:
                    st.wait_stream(src)
            if st not in self._used:
                self._used.append(st)
        todo = self.thunks
        if self.batch:
            # launches that are plain weight-gradient calls travel as ONE parameter array per side stream (mrfa_conv2d_wgrad_multi: the problems the
            # small-problem kernel takes then run up to 28 per launch); the two-stage scratch is per stream, so it is re-pointed like launch() does
            plain = [fn for fn in todo if getattr(fn, "params", None) is not None]
            todo = [fn for fn in todo if getattr(fn, "params", None) is None]
            for k, st in enumerate(lanes):
                mine = plain[k::len(lanes)]
                if not mine:
                    continue
                with torch.cuda.stream(st):
                    self._issue_plain(mine, dev)
        for fn in todo:
            with torch.cuda.stream(lanes[self._rr % len(lanes)]):
                fn()
            self._rr += 1
        # the closures own the buffers the side streams are still reading -- ALL flushes of the step (a collection can be flushed more than once:
        # the third encoder pass of the reference objective finishes its backward before the decoder's starts) until join() / reset()
        self.kept = (self.kept or []) + self.thunks
        self.thunks = []
        self.flushed = True

    @staticmethod
    def _issue_plain(fns, dev: torch.device):
        """the parameter blocks of plain weight-gradient launches as ONE array on the current stream (mrfa_conv2d_wgrad_multi)"""
        if not fns:
            return
        sp = hip.stream_ptr()
        ws = Ctx._wgrad_ws.get((dev, sp))
        if ws is None:
            ws = Ctx._wgrad_ws[(dev, sp)] = torch.empty(16 << 20, dtype=torch.float32, device=dev)
        arr = (hip.WgradParams * len(fns))()
        for i, fn in enumerate(fns):
            C.memmove(C.byref(arr, i * C.sizeof(hip.WgradParams)), C.byref(fn.params), C.sizeof(hip.WgradParams))
            arr[i].ws, arr[i].ws_bytes = ws.data_ptr(), ws.numel() * 4
        hip.check(hip.lib().mrfa_conv2d_wgrad_multi(sp, arr, len(fns)), "wgrad_multi(deferred)")

    def flush(self, dev: torch.device):
        """issue everything collected so far on the side streams, ordered after the current stream; then the finals on the first of them, after all"""
        if not self.thunks and not self.finals:
            return
        if dev.type != "cuda":
            self._deal(dev)
            for fn in self.finals:
                fn()
            self.finals = []
            return
        tag = "enc wgrads" if self.manual else "dec wgrads"
        mark(tag + ": flush")
        self._deal(dev)
        self._lanes(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        for st in self._used + self._srcs:
            if st is not self.stream:
                self.stream.wait_stream(st)
        self._srcs = []
        if self.stream not in self._used:
            self._used.append(self.stream)
        with torch.cuda.stream(self.stream):
            for fn in self.finals:
                fn()
            mark(tag + ": done")
        self.kept = (self.kept or []) + self.finals
        self.finals = []
        self.flushed = True

    def join(self, dev: torch.device):
        self.flush(dev)                                   # (no non-deferring program ran its backward: nothing overlapped, still correct)
        if self.flushed and dev.type == "cuda":
            cur = torch.cuda.current_stream(dev)
            for st in self._used:
                cur.wait_stream(st)
        self.kept, self.flushed, self._used = None, False, []


WGRAD_DEFER: Optional[DeferredWgrads] = None           # programs whose forward runs under defer_wgrads(d) defer into d
_PENDING_DEFERRED: List[DeferredWgrads] = []            # collections with thunks not yet issued


class defer_wgrads:
    def __init__(self, d: Optional[DeferredWgrads]):
        self.d = d

    def __enter__(self):
        global WGRAD_DEFER
        self.prev, WGRAD_DEFER = WGRAD_DEFER, self.d
        return self.d

    def __exit__(self, *exc):
        global WGRAD_DEFER
        WGRAD_DEFER = self.prev
        return False


SYNCBN_FORCE = os.environ.get("MRFA_SYNCBN_FORCE_COLLECTIVE", "0") == "1"
SYNCBN_COLLECTIVES = 0          # statistics all-reduces issued by this process so far (bench.py --sync-bn reports the number per step)
SYNCBN_EXCHANGES = 0            # ... and the layer passes (forward / backward) they served: the collective count of the one-collective-per-layer form


def _syncbn_all_reduce(t: torch.Tensor, layers: int = 1):
    global SYNCBN_COLLECTIVES, SYNCBN_EXCHANGES
    SYNCBN_COLLECTIVES += 1
    SYNCBN_EXCHANGES += layers
    torch.distributed.all_reduce(t)


# SyncBatchNorm layers whose inputs do not depend on each other's outputs -- the same block position of HRNet's parallel resolution branches, the terms of a
# fuse layer (transformer/hr_base.py) -- exchange their statistics in ONE collective (Ctx.sync_stats; the backward sums likewise): 422 -> 245 collectives
# per training step with the MTIA prior.  A collective of a few KB costs its latency, and that latency sits on the step's critical chain once per layer and
# direction, so at N = 8 the count is what matters; at one rank (a forced collective) it measures nothing.  MRFA_SYNCBN_LOCKSTEP=0: one collective per layer.
SYNCBN_LOCKSTEP = os.environ.get("MRFA_SYNCBN_LOCKSTEP", "1") != "0"


def _syncbn_all_reduce_many(ts):
    """all-reduce the 1-D fp64 blocks `ts` (slices of the zero-filled pool chunks of Pool.take) with ONE collective per run of blocks that lie back to back in a
    chunk.  The words between two blocks of a run (alignment padding, the unused ticket / barrier words behind a block) travel along: they are the same on
    every rank and nothing reads them in a SyncBatchNorm step."""
    runs = []
    for t in sorted(ts, key=lambda t: t.data_ptr()):
        base = t._base if t._base is not None else t
        lo, hi = t.storage_offset(), t.storage_offset() + (t.numel() + 3) // 4 * 4
        if runs and runs[-1][0] is base and runs[-1][2] == lo:
            runs[-1][2] = hi
            runs[-1][3] += 1
        else:
            runs.append([base, lo, hi, 1])
    for base, lo, hi, n in runs:
        _syncbn_all_reduce(base.view(-1)[lo:min(hi, base.numel())], n)
SYNCBN_DIRECT_BYTES = int(os.environ.get("MRFA_SYNCBN_DIRECT_KIB", "128")) << 10      # statistics blocks up to this size are all-reduced whole (all slots)
PHASE_UPCONV = os.environ.get("MRFA_PHASE_UPCONV", "1") != "0"        # forward / data gradient of fused-upsample 3x3 layers in phase form
# BatchNorm finalize of the conv -> BatchNorm pairs of the keypoint encoder inside the convolution's launch (last workgroup; mrfa_conv_params.fin_*)
BN_FIN_FUSED = os.environ.get("MRFA_BN_FIN_FUSED", "1") != "0"
# first phase of the BatchNorm backward of single-consumer BatchNorm outputs inside the consumer's data-gradient launch (mrfa_conv_params.bst_*)
BN_BWD_IN_DGRAD = os.environ.get("MRFA_BN_BWD_IN_DGRAD", "1") != "0"
FUSED_RESIZES = os.environ.get("MRFA_FUSED_RESIZES", "1") != "0"      # copies / resizes issued inside Ctx.fused_resizes() travel as ONE launch per direction (kept for same-box A/B runs)
FUSED_SPLITK = os.environ.get("MRFA_FUSED_SPLITK", "1") != "0"        # K splits that finish inside their launch (Ctx._conv_out; kept for same-box A/B runs)
# BatchNorm-apply + ReLU between the two convolutions of a residual block as the second one's prologue (Ctx.prologue_ok; hr_base.BasicBlock): 56 launches and
# 56 x 8 MB tensors fewer per forward pass, exact (tests/test_graph_gpu.py::test_prologue_fusion_equals_the_bn_act_path) -- and measured SLOWER in the
# training step: 81.61 ms (81.4-81.8) against 80.30 (79.5-80.9) without it over three alternating rounds of 40 steps on one box
# (profiles/r6_ab_prologue_wgrad_lean.txt): the split + prologue arithmetic lands in the staging of conv_lean / wgrad_lean, whose one or two waves per SIMD
# have no slack for it, while the bn_act launches it removes ran beside other lanes' kernels.  OFF by default (MRFA_PROLOGUE_FUSION=1 switches it on).
PROLOGUE_FUSION = os.environ.get("MRFA_PROLOGUE_FUSION", "0") == "1"
LAZY_FP32_PACKS = os.environ.get("MRFA_LAZY_FP32_PACKS", "1") != "0"      # fp32 weight layouts only for the launches that read them (Ctx._fp32_weights)
RELU_IN = os.environ.get("MRFA_RELU_IN", "1") != "0"                  # ReLU backward of single-consumer tensors inside the consumer's data gradient
# gradient buffers of at least this many floats are not zero-filled before the backward pass (Storage.fresh); smaller ones share one
# zero arena (one fill instead of hundreds of tiny ones).  9 MiB (round 2: 4): the TokenPose_B encoder's 4 and 8 MiB buffers (32 / 64
# channels @64^2, B = 8) are mostly first touched by accumulating kernels (residual gradients, sub-sampling, GELU, attention), i.e. each
# paid its own fill launch on a latency-bound chain -- 259 fill launches per step; A/B on one box 89.2 -> 88.5 ms.
FRESH_MIN_ELEMS = (int(os.environ.get("MRFA_FRESH_MIN_MIB", "9")) << 20) // 4
# debug switch (MRFA_FRESH_NAN=1; tests set it together with FRESH_MIN_ELEMS = 0): every lazily initialised gradient buffer starts
# as NaN instead of whatever the allocator hands out, so a kernel that READS a buffer no writer has covered -- or a first writer that
# does not cover all of it -- shows up as a non-finite gradient instead of passing by luck
FRESH_NAN = os.environ.get("MRFA_FRESH_NAN", "0") == "1"
# HRNet's resolution branches (three independent chains of BasicBlocks per stage-3 module) as parallel branches of the captured hipGraph (Ctx.lanes()).
# On since round 5: with ONE batched encoder pass (statistic groups below) the forward chain takes 8.2 instead of 10.3 ms with them (round 4, beside a
# second encoder pass on its own stream, they lost: more than four live branches are folded together by the graph executor).  Never with
# SyncBatchNorm collectives (every rank must enqueue them on ONE stream in ONE order): mrfa_amd.graph.GraphedTrainStep switches them off then.
BRANCH_STREAMS = os.environ.get("MRFA_BRANCH_STREAMS", "1") != "0"        # (kept as a switch for same-box A/B runs)


# ---- statistic groups (include/mrfa_hip.h v7): the reference's separate encoder calls -- encoder(source), encoder(driving), encoder(transformed driving),
# model.py:185-186,234 -- travel through ONE program as a batch of groups x B samples.  In train mode every BatchNorm keeps its quantities per group
# (batch statistics, scale / shift, the backward's sums; `groups` momentum updates of the running statistics in group order; num_batches_tracked += groups);
# everything else has no cross-sample coupling.  Half / a third of the launches of a latency-bound chain, twice / three times the rows per launch.
STAT_GROUPS = 1


class stat_groups:
    """with stat_groups(g): programs started inside treat their batch as g statistic groups of N / g consecutive samples"""

    def __init__(self, groups: int):
        self.groups = int(groups)

    def __enter__(self):
        global STAT_GROUPS
        self.prev, STAT_GROUPS = STAT_GROUPS, self.groups

    def __exit__(self, *exc):
        global STAT_GROUPS
        STAT_GROUPS = self.prev
        return False


def prepare_packs(module: torch.nn.Module) -> bool:
    """Refresh, on the current stream, every packed-weight layout the convolutions of `module` have built so far; False if
    the module has not run forward AND backward yet (layouts and gather tables are then still created lazily, inside the
    passes) -- the caller must not fork streams in that case."""
    cws = [m._mrfa_convw for m in module.modules() if getattr(m, "_mrfa_convw", None) is not None]
    if not cws or any(cw._fwd is None and cw._fo is None for cw in cws) or all(cw._dg is None and cw._fi is None for cw in cws):
        return False
    for cw in cws:
        if cw._fwd is not None:
            cw.fwd_pack(getattr(cw, "_fwd_padded", False))
        if cw._dg is not None:
            cw.dgrad_pack(getattr(cw, "_dg_padded", False))
        if getattr(cw, "_fwd_s", None) is not None:
            cw.split_pack("f", getattr(cw, "_fwd_padded", False))
        if getattr(cw, "_dg_s", None) is not None:
            cw.split_pack("d", getattr(cw, "_dg_padded", False))
        if getattr(cw, "_fwd_r", None) is not None:
            cw.split_pack("f", getattr(cw, "_fwd_padded", False), rne=True)
        if getattr(cw, "_dg_r", None) is not None:
            cw.split_pack("d", getattr(cw, "_dg_padded", False), rne=True)
        if getattr(cw, "_fwd_ph", None) is not None:
            cw.phase_pack()
        if getattr(cw, "_dg_ph", None) is not None:
            cw.phase_pack(dgrad=True)
        if cw._fo is not None:
            cw.fewout_pack()
        if cw._fi is not None:
            cw.fewin_dgrad_pack()
    return True


class direct_param_grads:
    def __enter__(self):
        global DIRECT_PARAM_GRADS
        self.prev, DIRECT_PARAM_GRADS = DIRECT_PARAM_GRADS, True

    def __exit__(self, *exc):
        global DIRECT_PARAM_GRADS
        DIRECT_PARAM_GRADS = self.prev


def _direct_ok(*params) -> bool:
    return DIRECT_PARAM_GRADS and all(p is None or (p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32)
                                      for p in params)


def _r4(c: int) -> int:
    return (c + 3) // 4 * 4


# ------------------------------------------------------------------------------------------------- storage / views
class Storage:
    """[rows, ld] fp32 buffer + lazily allocated gradient buffer of the same geometry."""
    __slots__ = ("data", "grad", "rows", "ld", "grad_noinit", "fresh", "bwd_masked", "bn_hint", "seq", "plan")

    def __init__(self, data: torch.Tensor):
        assert data.dim() == 2 and data.dtype == torch.float32 and data.is_contiguous()
        self.data = data
        self.rows, self.ld = data.shape
        self.grad: Optional[torch.Tensor] = None
        # True: the gradient buffer has exactly one writer that overwrites all of it (a raw conv output whose only consumer
        # is its BatchNorm): allocated uninitialised, outside the zero arena
        self.grad_noinit = False
        # True: `grad` is allocated but NOT yet initialised and nobody has written it (big gradient buffers are not zero-filled up
        # front).  The first writer either overwrites all of it (Ctx._claim: conv data-gradients, BatchNorm backward) or, touching
        # it through grad_buf() / View.gptr like every other op, gets it zero-filled first.  A producer that finds its output still
        # fresh knows that no consumer sent a gradient (View.has_grad is False) and skips its backward.
        self.fresh = False
        # True: the buffer holds ReLU outputs whose ONLY consumer is a conv called with relu_in=True: that consumer's data gradient
        # applies the ReLU mask (fused into its epilogue where the library can, a masking pass of its own otherwise), so the
        # producers skip their ReLU-backward pass
        self.bwd_masked = False
        # not None: the buffer is the output of a BatchNorm + activation with ONE consumer, a convolution (Ctx.bn_act(out_sole=True)): what that
        # convolution's data gradient needs to accumulate the first phase of the BatchNorm's backward in its epilogue (Ctx._conv_dgrad)
        self.bn_hint = None
        # position of this buffer among its program's forward activations and that program's zero plan (Ctx.zero_plan): a gradient buffer that had to be
        # zero-filled on first touch is remembered there, and the next run of the same program zero-fills all such buffers in ONE multi-tensor launch
        self.seq = -1
        self.plan = None

    def grad_buf(self) -> torch.Tensor:
        if self.grad is None:
            self.grad = torch.zeros_like(self.data)
        elif self.fresh:
            self.grad.zero_()
            self.fresh = False
            if self.plan is not None and self.seq >= 0:
                self.plan.add(self.seq)
        return self.grad


class View:
    """NHWC activation view: C channels starting at channel offset `coff` of a Storage with leading dimension ld."""
    __slots__ = ("st", "N", "H", "W", "C", "coff", "zpad")

    def __init__(self, st: Storage, N: int, H: int, W: int, C_: int, coff: int = 0, zpad: bool = False):
        assert N * H * W == st.rows and coff + C_ <= st.ld, (N, H, W, C_, coff, st.rows, st.ld)
        self.st, self.N, self.H, self.W, self.C, self.coff = st, N, H, W, C_, coff
        # zpad: the channels [C, roundup32(C)) exist inside ld and hold zeros, so a conv may read the view as if it had
        # roundup32(C) channels (with zero-padded weights) and stay on the vectorised "chunked" MFMA path
        self.zpad = zpad

    @property
    def rows(self) -> int:
        return self.st.rows

    @property
    def ld(self) -> int:
        return self.st.ld

    @property
    def ptr(self) -> int:
        return self.st.data.data_ptr() + 4 * self.coff

    @property
    def gptr(self) -> int:
        return self.st.grad_buf().data_ptr() + 4 * self.coff

    @property
    def has_grad(self) -> bool:
        return self.st.grad is not None and not self.st.fresh

    def slice(self, c0: int, c1: int) -> "View":
        assert 0 <= c0 < c1 <= self.C
        return View(self.st, self.N, self.H, self.W, c1 - c0, self.coff + c0)

    def tensor(self) -> torch.Tensor:
        """(N,H,W,C) torch view of the data (strided when the storage is wider)."""
        return self.st.data.view(self.N, self.H, self.W, self.ld)[..., self.coff:self.coff + self.C]

    def grad_tensor(self) -> torch.Tensor:
        return self.st.grad_buf().view(self.N, self.H, self.W, self.ld)[..., self.coff:self.coff + self.C]


class ZeroPool:
    """Hands out zero-filled 1-D slices carved from large zero chunks: one memset per chunk instead of one tiny fill
    kernel per statistics / gradient-accumulator buffer (~1000 fills per training step otherwise)."""

    def __init__(self, dev, dtype, chunk_elems: int):
        self.dev, self.dtype, self.chunk = dev, dtype, chunk_elems
        self.cur: Optional[torch.Tensor] = None
        self.off = 0

    def take(self, n: int) -> torch.Tensor:
        n4 = (n + 3) // 4 * 4
        if n4 > self.chunk // 4:
            return torch.zeros(n, dtype=self.dtype, device=self.dev)
        if self.cur is None or self.off + n4 > self.chunk:
            self.cur = torch.zeros(self.chunk, dtype=self.dtype, device=self.dev)
            self.off = 0
        v = self.cur[self.off:self.off + n]
        self.off += n4
        return v


# ------------------------------------------------------------------------------------------------- parameters
class ConvW:
    """Packed-weight cache + packed-gradient accumulator for one nn.Conv2d parameter pair (OIHW weight, bias)."""

    def __init__(self, conv: torch.nn.Conv2d):
        self.conv = conv
        w = conv.weight
        if w.dim() == 2:                       # nn.Linear = 1x1 convolution over token rows (same OI memory layout)
            (self.Cout, self.Cin), self.R, self.S, self.pad, self.stride = w.shape, 1, 1, 0, 1
        else:
            self.Cout, self.Cin, self.R, self.S = w.shape
            self.pad = conv.padding[0] if isinstance(conv.padding, tuple) else int(conv.padding)
            st = getattr(conv, "stride", 1)
            self.stride = st[0] if isinstance(st, tuple) else int(st)
        self.T = self.R * self.S
        self.fwd_flat = (self.Cin % 32) != 0
        self.dgrad_flat = (self.Cout % 32) != 0
        # flat wgrad (GEMM N axis = taps*Cin, gathered): few input channels, or big kernels over an odd channel count
        self.wgrad_flat = self.Cin < 32 or (self.T >= 25 and self.Cin % 32 != 0 and self.Cin < 64)
        # direct VALU kernels for <= 4 output channels (forward + weight gradient) and for the data gradient of
        # convs with <= 4 input channels (= a few-output conv over dY)
        self.fewout = self.Cout <= 4 and self.Cin % 4 == 0 and self.R == self.S
        self.fewin = self.Cin <= 4 and self.Cout % 4 == 0 and self.R == self.S
        self._fo = self._fi = None
        self._ver_fo = self._ver_fi = None
        self._fwd = None
        self._dg = None
        self._ver_f = self._ver_d = None
        self._ktab_f = None
        self._ktab_d = None
        self.dw_acc: Optional[torch.Tensor] = None
        self.db_acc: Optional[torch.Tensor] = None

    # -- tables
    @staticmethod
    def _ktab(Cc: int, R: int, S: int, pad: int, device) -> torch.Tensor:
        kp = (R * S * Cc + 31) // 32 * 32
        buf = (C.c_int * kp)()
        hip.check(hip.lib().mrfa_build_ktab(buf, Cc, R, S, pad, 0), "mrfa_build_ktab")
        return torch.tensor(list(buf), dtype=torch.int32, device=device)

    def ktab_fwd(self):
        if self._ktab_f is None or self._ktab_f.device != self.conv.weight.device:
            self._ktab_f = self._ktab(self.Cin, self.R, self.S, self.pad, self.conv.weight.device)
        return self._ktab_f

    def ktab_dgrad(self):
        if self._ktab_d is None or self._ktab_d.device != self.conv.weight.device:
            self._ktab_d = self._ktab(self.Cout, self.R, self.S, self.R - 1 - self.pad, self.conv.weight.device)
        return self._ktab_d

    # -- packs (re-done whenever the parameter was modified in place, e.g. by the optimizer)
    def _key(self):
        w = self.conv.weight
        return (w._version, w.data_ptr(), CAPTURE_KEY)

    def fwd_pack(self, padded: bool = False) -> torch.Tensor:
        """padded: chunked layout with Cin zero-padded to a multiple of 32 even though Cin % 32 != 0 (zpad inputs)"""
        if padded != getattr(self, "_fwd_padded", False):
            self._fwd = None
            self._fwd_padded = padded
        if self._fwd is None or self._ver_f != self._key():
            w = self.conv.weight.detach()
            cop = (self.Cout + 127) // 128 * 128
            if padded:
                n, mode = self.T * cop * ((self.Cin + 31) // 32 * 32), 0
            elif self.fwd_flat:
                kp = (self.T * self.Cin + 31) // 32 * 32
                n, mode = cop * kp, 1
            else:
                n, mode = self.T * cop * self.Cin, 0
            if self._fwd is None or self._fwd.numel() != n or self._fwd.device != w.device:
                self._fwd = torch.zeros(n, dtype=torch.float32, device=w.device)      # zero pads: PackPlan never writes them
            hip.check(hip.lib().mrfa_pack_conv_weight(hip.stream_ptr(), w.contiguous().data_ptr(), self._fwd.data_ptr(),
                                                      self.Cout, self.Cin, self.R, self.S, mode), "pack(fwd)")
            self._ver_f = self._key()
        return self._fwd

    def dgrad_pack(self, padded: bool = False) -> torch.Tensor:
        if padded != getattr(self, "_dg_padded", False):
            self._dg = None
            self._dg_padded = padded
        if self._dg is None or self._ver_d != self._key():
            w = self.conv.weight.detach()
            cip = (self.Cin + 127) // 128 * 128
            if padded:
                n, mode = self.T * cip * ((self.Cout + 31) // 32 * 32), 2
            elif self.dgrad_flat:
                kp = (self.T * self.Cout + 31) // 32 * 32
                n, mode = cip * kp, 3
            else:
                n, mode = self.T * cip * self.Cout, 2
            if self._dg is None or self._dg.numel() != n or self._dg.device != w.device:
                self._dg = torch.zeros(n, dtype=torch.float32, device=w.device)
            hip.check(hip.lib().mrfa_pack_conv_weight(hip.stream_ptr(), w.contiguous().data_ptr(), self._dg.data_ptr(),
                                                      self.Cout, self.Cin, self.R, self.S, mode), "pack(dgrad)")
            self._ver_d = self._key()
        return self._dg

    def split_pack(self, which: str, padded: bool, rne: bool = False) -> tuple:
        """(buffer, elements per piece) of the weights pre-split into three bf16 pieces for the bf16x6 kernels (pack mode 8
        = forward layout, 9 = data-gradient layout); chunked layouts only"""
        fwd = which == "f"
        attr, ver = ("_fwd_s", "_ver_fs") if fwd else ("_dg_s", "_ver_ds")
        if rne:                              # plain bf16 mode: ONE plane rounded to nearest even (pack modes 14 / 15)
            attr, ver = ("_fwd_r", "_ver_fr") if fwd else ("_dg_r", "_ver_dr")
        if fwd:
            rows, cols = (self.Cout + 127) // 128 * 128, (self.Cin + 31) // 32 * 32
        else:
            rows, cols = (self.Cin + 127) // 128 * 128, (self.Cout + 31) // 32 * 32
        piece = self.T * rows * cols
        npl = 1 if rne else 3
        buf = getattr(self, attr, None)
        w = self.conv.weight.detach()
        if buf is None or buf.numel() != npl * piece or buf.device != w.device:
            buf = torch.zeros(npl * piece, dtype=torch.int16, device=w.device)
            setattr(self, attr, buf)
            setattr(self, ver, None)
        if getattr(self, ver, None) != self._key():
            d = hip.PackDesc()
            d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.contiguous().data_ptr(), self.Cout, self.Cin, self.R, self.S, 1
            d.dst[0], d.mode[0] = buf.data_ptr(), ((14 if fwd else 15) if rne else (8 if fwd else 9))
            hip.check(hip.lib().mrfa_pack_conv_weights_multi(hip.stream_ptr(), C.pointer(d), 1), "pack(split)")
            setattr(self, ver, self._key())
        return buf, (0 if rne else piece)

    def phase_pack(self, dgrad: bool = False) -> tuple:
        """(buffer, elements per piece) of the 16 phase-tap weights of nearest-x2 + this 3x3 conv (pack mode 12: UpBlock2d as four 2x2
        convolutions on the low-resolution input, util.py:172-176), pre-split into three bf16 pieces; dgrad: transposed (mode 13)"""
        assert self.R == 3 and self.S == 3
        attr, ver, mode = ("_dg_ph", "_ver_dph", 13) if dgrad else ("_fwd_ph", "_ver_ph", 12)
        if dgrad:
            assert not self.dgrad_flat
            piece = 16 * ((self.Cin + 127) // 128 * 128) * self.Cout
        else:
            assert not self.fwd_flat
            piece = 16 * ((self.Cout + 127) // 128 * 128) * self.Cin
        buf = getattr(self, attr, None)
        w = self.conv.weight.detach()
        if buf is None or buf.numel() != 3 * piece or buf.device != w.device:
            buf = torch.zeros(3 * piece, dtype=torch.int16, device=w.device)
            setattr(self, attr, buf)
            setattr(self, ver, None)
        if getattr(self, ver, None) != self._key():
            d = hip.PackDesc()
            d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.contiguous().data_ptr(), self.Cout, self.Cin, self.R, self.S, 1
            d.dst[0], d.mode[0] = buf.data_ptr(), mode
            hip.check(hip.lib().mrfa_pack_conv_weights_multi(hip.stream_ptr(), C.pointer(d), 1), "pack(phase)")
            setattr(self, ver, self._key())
        return buf, piece

    def _simple_pack(self, attr, ver_attr, mode):
        if getattr(self, attr) is None or getattr(self, ver_attr) != self._key():
            w = self.conv.weight.detach()
            buf = getattr(self, attr)
            if buf is None or buf.device != w.device:
                buf = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
                setattr(self, attr, buf)
            hip.check(hip.lib().mrfa_pack_conv_weight(hip.stream_ptr(), w.contiguous().data_ptr(), buf.data_ptr(), self.Cout, self.Cin,
                                                      self.R, self.S, mode), f"pack(mode {mode})")
            setattr(self, ver_attr, self._key())
        return getattr(self, attr)

    def fewout_pack(self) -> torch.Tensor:          # [Cout][tap][Cin]
        return self._simple_pack("_fo", "_ver_fo", 5)

    def fewin_dgrad_pack(self) -> torch.Tensor:     # [Cin][tap'][Cout]
        return self._simple_pack("_fi", "_ver_fi", 7)

    def grad_acc(self, pool: Optional[ZeroPool] = None):
        if self.dw_acc is None:
            dev = self.conv.weight.device
            z = pool.take if pool is not None else (lambda n: torch.zeros(n, dtype=torch.float32, device=dev))
            self._direct = _direct_ok(self.conv.weight, self.conv.bias)
            self.dw_acc = z(self.T * self.Cout * self.Cin)
            if self.conv.bias is None:
                self.db_acc = None
            else:                                  # direct mode: the bias gradient is atomically added where it belongs
                self.db_acc = self.conv.bias.grad if self._direct else z(self.Cout)
        return self.dw_acc, self.db_acc

    def take_grads(self):
        """-> (dW OIHW or None, dbias or None); clears the accumulators.  Direct mode: the packed [tap][Cout][Cin]
        accumulator is un-packed INTO weight.grad (+=) and (None, None) is returned."""
        if self.dw_acc is None:
            return None, None
        mode = 6 if self.fewout else 4
        if self._direct:
            hip.check(hip.lib().mrfa_pack_conv_weight(hip.stream_ptr(), self.dw_acc.data_ptr(), self.conv.weight.grad.data_ptr(), self.Cout,
                                                      self.Cin, self.R, self.S, mode), "unpack(wgrad, +=)")
            self.dw_acc = self.db_acc = None
            return None, None
        dw = torch.empty_like(self.conv.weight)
        hip.check(hip.lib().mrfa_pack_conv_weight(hip.stream_ptr(), self.dw_acc.data_ptr(), dw.data_ptr(), self.Cout, self.Cin,
                                                  self.R, self.S, mode | 16), "unpack(wgrad)")
        db = self.db_acc
        self.dw_acc = self.db_acc = None
        return dw, db


class PackPlan:
    """Every packed layout that exists so far (i.e. after one forward + backward) of every convolution of `model`,
    refreshed from the current weights by ONE batched launch per 48 convolutions (mrfa_pack_conv_weights_multi) instead
    of ~180 single-layout launches; run() also marks the per-convolution caches valid, so the engine's own
    fwd_pack()/dgrad_pack() calls of the same step find nothing to do."""

    def __init__(self, model: torch.nn.Module, only: Optional[torch.nn.Module] = None, exclude: Optional[torch.nn.Module] = None):
        """only / exclude: a sub-module of `model` whose convolutions are the plan / are left out of it (GraphedTrainStep packs the keypoint
        encoder's layouts first and everything else on a side stream beside the encoder's forward)"""
        skip = {id(m) for m in exclude.modules()} if exclude is not None else set()
        self.cws: List[ConvW] = [m._mrfa_convw for m in (only if only is not None else model).modules()
                                 if getattr(m, "_mrfa_convw", None) is not None and id(m) not in skip]
        descs = []
        self._keep = []
        for cw in self.cws:
            w = cw.conv.weight
            assert w.is_contiguous()
            dsts = []
            if cw._fwd is not None:
                dsts.append((cw._fwd, 0 if (getattr(cw, "_fwd_padded", False) or not cw.fwd_flat) else 1))
            if cw._dg is not None:
                dsts.append((cw._dg, 2 if (getattr(cw, "_dg_padded", False) or not cw.dgrad_flat) else 3))
            if getattr(cw, "_fwd_s", None) is not None:
                dsts.append((cw._fwd_s, 8))
            if getattr(cw, "_dg_s", None) is not None:
                dsts.append((cw._dg_s, 9))
            if getattr(cw, "_fwd_r", None) is not None:
                dsts.append((cw._fwd_r, 14))
            if getattr(cw, "_dg_r", None) is not None:
                dsts.append((cw._dg_r, 15))
            if getattr(cw, "_fwd_ph", None) is not None:
                dsts.append((cw._fwd_ph, 12))
            if getattr(cw, "_dg_ph", None) is not None:
                dsts.append((cw._dg_ph, 13))
            if cw._fo is not None:
                dsts.append((cw._fo, 5))
            if cw._fi is not None:
                dsts.append((cw._fi, 7))
            for i in range(0, len(dsts), 3):
                d = hip.PackDesc()
                d.src, d.Cout, d.Cin, d.R, d.S = w.data_ptr(), cw.Cout, cw.Cin, cw.R, cw.S
                part = dsts[i:i + 3]
                d.ndst = len(part)
                for k, (buf, mode) in enumerate(part):
                    d.dst[k], d.mode[k] = buf.data_ptr(), mode
                    self._keep.append(buf)
                descs.append(d)
        self.n = len(descs)
        self.table = (hip.PackDesc * max(self.n, 1))(*descs)
        self.ptrs = [(cw, cw.conv.weight.data_ptr()) + self._buffer_ids(cw) for cw in self.cws]

    _PLANES = ("_fwd", "_dg", "_fwd_s", "_dg_s", "_fwd_r", "_dg_r", "_fwd_ph", "_dg_ph", "_fo", "_fi")

    @classmethod
    def _buffer_ids(cls, cw) -> tuple:
        """identity of every packed layout the plan may write: a layout created (or re-allocated) after the plan was built is not in
        the plan's table, and stamping its version below would make the engine skip the pack it needs -- stale weights under replay"""
        return tuple(id(getattr(cw, a, None)) for a in cls._PLANES)

    def run(self):
        if self.n:
            hip.check(hip.lib().mrfa_pack_conv_weights_multi(hip.stream_ptr(), self.table, self.n), "pack_conv_weights_multi")
        for rec in self.ptrs:
            cw, wptr, ids = rec[0], rec[1], rec[2:]
            assert cw.conv.weight.data_ptr() == wptr and self._buffer_ids(cw) == ids, "PackPlan is stale: rebuild it"
            k = cw._key()
            cw._ver_f = cw._ver_d = cw._ver_fo = cw._ver_fi = cw._ver_fs = cw._ver_ds = cw._ver_ph = cw._ver_dph = cw._ver_fr = cw._ver_dr = k


def unpack_direct(cws: List["ConvW"], accs: Optional[List[torch.Tensor]] = None):
    """direct-gradient mode: weight.grad += un-packed accumulator for all given convolutions in one batched launch.  accs: the accumulators
    themselves (a deferring program detaches them from its ConvW objects when its backward ends, so that no later program shares them)"""
    if not cws:
        return
    table = (hip.UnpackDesc * len(cws))()
    for i, (d, cw) in enumerate(zip(table, cws)):
        d.src, d.dst = (accs[i] if accs is not None else cw.dw_acc).data_ptr(), cw.conv.weight.grad.data_ptr()
        d.Cout, d.Cin, d.T, d.fewout = cw.Cout, cw.Cin, cw.T, int(cw.fewout)
    hip.check(hip.lib().mrfa_unpack_wgrads_multi(hip.stream_ptr(), table, len(cws)), "unpack_wgrads_multi")
    if accs is None:
        for cw in cws:
            cw.dw_acc = cw.db_acc = None


def convw(conv: torch.nn.Conv2d) -> ConvW:
    cw = getattr(conv, "_mrfa_convw", None)
    if cw is None:
        cw = ConvW(conv)
        object.__setattr__(conv, "_mrfa_convw", cw)
    return cw


class BNGrad:
    """gradient accumulators for one BatchNorm2d (gamma, beta)."""

    def __init__(self, bn: torch.nn.BatchNorm2d):
        self.bn = bn
        self.dg: Optional[torch.Tensor] = None
        self.db: Optional[torch.Tensor] = None

    def acc(self, pool: Optional[ZeroPool] = None):
        if self.dg is None:
            self._direct = _direct_ok(self.bn.weight, self.bn.bias)
            if self._direct:                       # the kernels accumulate (+=) straight into the existing gradients
                self.dg, self.db = self.bn.weight.grad, self.bn.bias.grad
            elif pool is not None:
                self.dg, self.db = pool.take(self.bn.weight.numel()), pool.take(self.bn.bias.numel())
            else:
                self.dg = torch.zeros_like(self.bn.weight)
                self.db = torch.zeros_like(self.bn.bias)
        return self.dg, self.db

    def take(self):
        dg, db = self.dg, self.db
        self.dg = self.db = None
        return (None, None) if self._direct else (dg, db)


def bngrad(bn) -> BNGrad:
    g = getattr(bn, "_mrfa_bngrad", None)
    if g is None:
        g = BNGrad(bn)
        object.__setattr__(bn, "_mrfa_bngrad", g)
    return g


# ------------------------------------------------------------------------------------------------- torch "glue islands"
class IslandOut:
    """Output of a torch glue island: a detached contiguous tensor that can be re-entered into the engine as an NHWC
    View (zero copy), fed to another island, or returned to the caller; gradients from all three routes are summed."""

    def __init__(self, t: torch.Tensor):
        self.t = t
        self._view: Optional[View] = None
        self.ext_grad: Optional[torch.Tensor] = None

    def view(self) -> "View":
        if self._view is None:
            t = self.t
            assert t.dim() == 4
            N, H, W, C_ = t.shape
            self._view = View(Storage(t.view(N * H * W, C_)), N, H, W, C_)
        return self._view

    def add_grad(self, g: torch.Tensor):
        # no .clone(): nothing mutates g afterwards, and a contiguous clone is a device-to-device hipMemcpyAsync, which a
        # stream capture records as a MEMCPY NODE -- on ROCm 7.2 memcpy / memset nodes are not reliably ordered against
        # the kernel nodes around them when the graph is replayed (mrfa_amd/graph.py: "MEMCPY / MEMSET NODES")
        g = g.reshape(self.t.shape)
        self.ext_grad = g if self.ext_grad is None else self.ext_grad + g

    def total_grad(self) -> Optional[torch.Tensor]:
        g = self.ext_grad
        if self._view is not None and self._view.has_grad:
            vg = self._view.st.grad.view(self.t.shape)
            g = vg if g is None else g + vg
        return g


# ------------------------------------------------------------------------------------------------- the op context
class Ctx:
    # bench.py sets this to a list to collect (config, flops, start_event, end_event) per MFMA conv/GEMM launch
    profile: Optional[list] = None
    _wgrad_ws: dict = {}                               # stream -> scratch for the two-stage wgrad split reduction
    _side: Optional["torch.cuda.Stream"] = None        # second stream for the weight-gradient side chain (WGRAD_STREAM)
    debug_backward: Optional[list] = None             # list -> run_backward appends per-closure gradient fingerprints

    def __init__(self, device: torch.device, train: bool, record: bool):
        self.dev = device
        self.train = train
        self.record = record
        self.tape: List[Callable[[], None]] = []
        self.L = hip.lib()
        self.split = self.L.mrfa_get_mfma_mode() in (1, 2)   # bf16x6 / bf16x3 kernels: also hand over pre-split weights
        self.bf16 = self.L.mrfa_get_mfma_mode() == 3         # plain bf16 products: the patch-tiled kernels take ONE rounded weight plane
        self.in_backward = False
        self.groups = STAT_GROUPS if train else 1       # statistic groups of the batch (stat_groups); eval mode has no batch statistics
        self.wdefer = WGRAD_DEFER            # not None: weight-gradient launches of this program are collected (DeferredWgrads)
        self.touched_convs: List[ConvW] = []
        self.touched_bns: List[BNGrad] = []
        self.nbt = {}                    # BatchNorm module -> forward passes in train mode during this program
        self.ext_grads = {}              # id(tensor) -> grad for module inputs / parameters touched by islands
        self.storages: List[Storage] = []            # forward activations whose gradients live in one zero arena
        self._pools = {}                 # (stream, dtype) -> ZeroPool: a chunk is zero-filled on the stream that carves it up
        self._home = hip.stream_ptr() if device.type == "cuda" else 0
        self._fin_done = {}              # statistics buffer -> result of a BatchNorm finalize done inside the producing convolution's call
        self._synced = {}                # statistics buffer -> backward group of a SyncBatchNorm layer whose statistics sync_stats() exchanged already
        # set of forward-activation indices whose (large, lazily initialised) gradient buffer needed a zero fill on first touch the last time this program
        # ran (run_program keeps one per module and input shapes: the programs are static): run_backward() zero-fills them in one multi-tensor launch
        # instead of ~100 fill launches scattered over the backward chains (tools/trace_fills.py: 121 fills per step, ~10 us each on the critical path)
        self.zero_plan: Optional[set] = None
        self._rs = None                    # fused_resizes(): the copies / resizes recorded for one launch

    # -- plumbing
    @property
    def s(self):
        return hip.stream_ptr()

    def _pool(self, dtype, big: int, small: int) -> "ZeroPool":
        st = self.s
        p = self._pools.get((st, dtype))
        if p is None:
            p = self._pools[(st, dtype)] = ZeroPool(self.dev, dtype, big if st == self._home else small)
        return p

    @property
    def pool32(self) -> "ZeroPool":
        return self._pool(torch.float32, 32 << 20, 2 << 20)         # 128 MiB chunks (8 MiB on branch streams)

    @property
    def pool64(self) -> "ZeroPool":
        return self._pool(torch.float64, 1 << 21, 1 << 19)          # 16 MiB chunks (4 MiB on branch streams): [STATS_SLOTS][2C] per BatchNorm

    # -- parallel branches inside one program (HRNet's resolution branches): only while a hipGraph is being captured -- a
    # captured graph owns static memory, whereas eager launches would hand blocks of the caching allocator from one stream
    # to another without the allocator knowing
    _lanes: dict = {}

    def lanes(self, n: int, enabled: Optional[bool] = None):
        """n side streams for branches of this program, or [] when branches must run in line (not capturing / CPU / switched off)"""
        on = BRANCH_STREAMS if enabled is None else enabled
        if self.dev.type != "cuda" or not on or not torch.cuda.is_current_stream_capturing():
            return []
        key = (self.dev, self.s)
        have = Ctx._lanes.setdefault(key, [])
        while len(have) < n:
            have.append(torch.cuda.Stream(device=self.dev))
        return have[:n]

    def fork(self, streams):
        """the branch streams start after everything issued so far; in backward the home stream waits for them here"""
        if not streams:
            return
        home = torch.cuda.current_stream(self.dev)
        for st in streams:
            st.wait_stream(home)
        if self.record:
            def bwd():
                cur = torch.cuda.current_stream(self.dev)
                for st in streams:
                    cur.wait_stream(st)
            self.tape.append(bwd)

    def join(self, streams):
        """the home stream waits for the branches; in backward the branch streams start from here"""
        if not streams:
            return
        home = torch.cuda.current_stream(self.dev)
        for st in streams:
            home.wait_stream(st)
        if self.record:
            def bwd():
                cur = torch.cuda.current_stream(self.dev)
                for st in streams:
                    st.wait_stream(cur)
            self.tape.append(bwd)

    def branch(self, stream, fn):
        """run fn() with `stream` current (None: in line); the backward closures it records run on the same stream"""
        if stream is None:
            return fn()
        n0 = len(self.tape)
        with torch.cuda.stream(stream):
            out = fn()
        for i in range(n0, len(self.tape)):
            f = self.tape[i]

            def on_stream(f=f):
                with torch.cuda.stream(stream):
                    f()
            self.tape[i] = on_stream
        return out

    def _chk(self, rc, what):
        if rc:
            hip.check(rc, what)

    def mark(self, name: str):
        """measurement aid (engine.MARKS set): a device timestamp here in the forward pass and, on the tape, where the backward pass comes back to this point"""
        if MARKS is None:
            return
        mark("fwd " + name)
        if self.record:
            self.tape.append(lambda: mark("bwd " + name))

    def rec(self, fn):
        if self.record:
            self.tape.append(fn)

    def _launch_conv(self, p, what: str, alg_cin: Optional[int] = None):
        if FORCE_TILE:
            p.tile = FORCE_TILE
        prof = Ctx.profile
        if prof is None:
            self._chk(self.L.mrfa_conv2d_nhwc(self.s, C.byref(p)), what)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._chk(self.L.mrfa_conv2d_nhwc(self.s, C.byref(p)), what)
        e1.record()
        nb = max(p.nbatch, 1)
        flops = 2.0 * nb * p.N * p.Hout * p.Wout * p.Cout * (alg_cin or p.Cin) * p.R * p.S      # algorithmic (unpadded)
        prof.append((self.L.mrfa_conv2d_last_config(), flops, e0, e1,
                     f"{what} {p.Cin}->{p.Cout} {p.R}x{p.S} @{p.Hout}x{p.Wout} N={p.N} nb={nb} ups={p.ups}"))

    def new(self, N, H, W, C_, ld=None, zero=False, pad32=False) -> View:
        if pad32 and C_ % 32 != 0:
            ld = (C_ + 31) // 32 * 32
            v = View(Storage(torch.zeros((N * H * W, ld), dtype=torch.float32, device=self.dev)), N, H, W, C_, 0, True)
            if self.record and not self.in_backward:
                self._register(v.st)
            return v
        ld = _r4(C_) if ld is None else ld
        alloc = torch.zeros if zero else torch.empty
        v = View(Storage(alloc((N * H * W, ld), dtype=torch.float32, device=self.dev)), N, H, W, C_)
        if self.record and not self.in_backward:
            self._register(v.st)
        return v

    def _register(self, st: "Storage"):
        st.seq, st.plan = len(self.storages), self.zero_plan
        self.storages.append(st)

    def wrap_nhwc(self, t: torch.Tensor) -> View:
        """Zero-copy view of a contiguous (N,H,W,C) fp32 tensor."""
        assert t.dim() == 4 and t.is_contiguous() and t.dtype == torch.float32
        N, H, W, C_ = t.shape
        return View(Storage(t.view(N * H * W, C_)), N, H, W, C_)

    def from_nchw(self, t: torch.Tensor, out: Optional[View] = None) -> View:
        N, C_, H, W = t.shape
        t = t.contiguous().float()
        out = out or self.new(N, H, W, C_)
        self._chk(self.L.mrfa_nchw_to_nhwc(self.s, t.data_ptr(), out.ptr, out.ld, N, C_, H, W, 0), "nchw_to_nhwc")
        return out

    def to_nchw(self, v: View) -> torch.Tensor:
        t = torch.empty((v.N, v.C, v.H, v.W), dtype=torch.float32, device=self.dev)
        self._chk(self.L.mrfa_nhwc_to_nchw(self.s, v.ptr, v.ld, t.data_ptr(), v.N, v.C, v.H, v.W, 0), "nhwc_to_nchw")
        return t

    def grad_to_nchw(self, v: View) -> torch.Tensor:
        t = torch.empty((v.N, v.C, v.H, v.W), dtype=torch.float32, device=self.dev)
        self._chk(self.L.mrfa_nhwc_to_nchw(self.s, v.gptr, v.ld, t.data_ptr(), v.N, v.C, v.H, v.W, 0), "nhwc_to_nchw(grad)")
        return t

    def seed_grad_nchw(self, v: View, g: torch.Tensor):
        g = g.contiguous().float()
        self._chk(self.L.mrfa_nchw_to_nhwc(self.s, g.data_ptr(), v.gptr, v.ld, v.N, v.C, v.H, v.W, 1), "seed grad")

    def f32(self, n, zero=False):
        return (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=self.dev)

    def f64z(self, n):
        return self.pool64.take(n)

    @staticmethod
    def _claim(v: View) -> bool:
        """True: the caller is the first writer of v's gradient buffer and covers all of it -- it must then OVERWRITE (no zero fill
        happened, nothing may be read).  Views with padding columns or a channel offset never qualify."""
        st = v.st
        if st.grad is not None and st.fresh and v.coff == 0 and v.C == st.ld:
            st.fresh = False
            return True
        return False

    # -- convolution ------------------------------------------------------------------------------------------
    def conv(self, x: View, conv: torch.nn.Conv2d, out: Optional[View] = None, *, relu=False, ups=False, pre=None,
             stats: Optional[torch.Tensor] = None, res: Optional[View] = None, need_dx=True, use_bias=True, relu_in=False, fin=None) -> View:
        """y = conv(pre(ups(x))) (+bias)(+res)(ReLU).  pre = (scale, shift) tensors of a pre-activation BN+ReLU.
        fin: the BatchNorm that follows (with `stats`): finished inside the convolution call (mrfa_conv_params.fin_*, see _fin_params).
        relu_in: the caller states that x (all of its storage) holds ReLU outputs and that this conv is their only consumer: the ReLU
        backward of x's producers then rides in this conv's data gradient (mrfa_conv_params.mask) instead of a pass of its own."""
        cw = convw(conv)
        relu_in = bool(relu_in and RELU_IN and self.record and need_dx and pre is None and not ups)
        if relu_in:
            assert x.coff == 0 and x.C == x.st.ld, "relu_in: the view must cover its storage (every producer skips its ReLU backward)"
            x.st.bwd_masked = True
        assert x.C == cw.Cin, (x.C, cw.Cin)
        Hv, Wv = (x.H * 2, x.W * 2) if ups else (x.H, x.W)
        Ho, Wo = Hv + 2 * cw.pad - cw.R + 1, Wv + 2 * cw.pad - cw.S + 1
        assert out is None or (out.N, out.H, out.W, out.C) == (x.N, Ho, Wo, cw.Cout)
        bias = conv.bias if use_bias else None
        direct = (cw.fewout and not ups and pre is None and res is None and stats is None and not relu
                  and x.ld % 4 == 0 and x.coff % 4 == 0)
        if direct:
            out = out or self.new(x.N, Ho, Wo, cw.Cout)
            self._chk(self.L.mrfa_conv_fewout_fwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, cw.Cin, cw.fewout_pack().data_ptr(),
                                                  bias.data_ptr() if bias is not None else None, out.ptr, out.ld, cw.Cout, cw.R, cw.pad, 0),
                      "conv_fewout_fwd")
            if self.record:
                def bwd_direct():
                    if not out.has_grad:
                        return
                    dw, db = cw.grad_acc(self.pool32)
                    dyp = out.gptr

                    def launch():
                        self._chk(self.L.mrfa_conv_fewout_wgrad(hip.stream_ptr(), x.ptr, x.ld, x.N, x.H, x.W, cw.Cin, dyp, out.ld, cw.Cout, cw.R,
                                                                cw.pad, dw.data_ptr(), db.data_ptr() if (bias is not None and db is not None) else None),
                                  "conv_fewout_wgrad")
                    if self._defer_ok(cw):
                        self.wdefer.add(launch)
                    else:
                        launch()
                    if need_dx:
                        self._conv_dgrad(x, cw, out, ups, pre, relu_in)
                self.tape.append(bwd_direct)
                if cw not in self.touched_convs:
                    self.touched_convs.append(cw)
            return out
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = x.ptr, x.ld, x.H, x.W, int(ups), x.N, cw.Cin
        padded = cw.fwd_flat and x.zpad and x.coff % 4 == 0 and pre is None
        cop = (cw.Cout + 127) // 128 * 128
        if padded:
            cip32 = (cw.Cin + 31) // 32 * 32
            p.Cin = cip32
            p.w_ld, p.w_tap, p.kflat = cip32, cop * cip32, 0
            if self.split or self.bf16:
                ws, p.w_piece = cw.split_pack("f", True, rne=self.bf16)
                p.w_split = ws.data_ptr()
        elif cw.fwd_flat:
            kp = (cw.T * cw.Cin + 31) // 32 * 32
            p.w_ld, p.w_tap, p.kflat = kp, 0, cw.T * cw.Cin
            p.ktab = cw.ktab_fwd().data_ptr()
        else:
            p.w_ld, p.w_tap, p.kflat = cw.Cin, cop * cw.Cin, 0
            if self.split or self.bf16:
                ws, p.w_piece = cw.split_pack("f", False, rne=self.bf16)
                p.w_split = ws.data_ptr()
                if self.split and ups and cw.R == 3 and cw.S == 3 and cw.pad == 1 and PHASE_UPCONV:
                    # UpBlock2d: the library may run nearest-x2 + 3x3 as four 2x2 phase convolutions (16 / 36 of the MACs, csrc/conv_halo.hip)
                    wph, p.w_phase_piece = cw.phase_pack()
                    p.w_phase = wph.data_ptr()
        p.w_rows = cop
        # the fp32 weight layout only where the kernel this call runs reads it (_fp32_weights below): a placeholder until the block is complete
        p.w = p.w_split if p.w_split else cw.fwd_pack(padded).data_ptr()
        p.Cout, p.Hout, p.Wout = cw.Cout, Ho, Wo
        p.R, p.S, p.pad = cw.R, cw.S, cw.pad
        if pre is not None:
            p.in_scale, p.in_shift, p.in_relu = pre[0].data_ptr(), pre[1].data_ptr(), 1
            if self.train and self.groups > 1:        # (v9: the prologue vectors are [groups][Cin]; prologue_ok() asked the library)
                p.groups = self.groups
        p.bias = bias.data_ptr() if bias is not None else None
        p.relu = int(relu)
        if res is not None:
            p.res, p.ldr = res.ptr, res.ld
        p.alpha, p.nbatch = 1.0, 1
        late_stats = False
        if stats is not None:
            p.stats, p.groups = stats.data_ptr(), self.groups
        out = self._conv_out(p, out, x.N, Ho, Wo, cw.Cout)
        if stats is not None:
            if self.groups > 1 and not self.L.mrfa_conv2d_groups_supported(C.byref(p)):
                assert pre is None, "a prologue with statistic groups where the library has none: ask prologue_ok() first"
                p.stats, p.groups, late_stats = None, 0, True       # (a tile would straddle two groups: one statistics pass per group behind the launch)
            elif fin is not None:
                self._fin_params(p, fin, stats, out.rows)
        self._fp32_weights(p, lambda: cw.fwd_pack(padded))
        self._launch_conv(p, "conv2d", cw.Cin)
        if late_stats:
            self._bn_stats_into(out, stats)

        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                if relu and not out.st.bwd_masked:
                    self._chk(self.L.mrfa_act_bwd(self.s, out.ptr, out.ld, out.gptr, out.ld, out.rows, out.C, 1, out.gptr, out.ld, 0),
                              "relu_bwd")
                if res is not None:
                    self._chk(self.L.mrfa_copy_view(self.s, out.gptr, out.ld, out.rows, out.C, res.gptr, res.ld, 1.0, 1), "res_bwd")
                if conv.weight.requires_grad:              # frozen weights (the VGG19 of the perceptual loss): data gradient only
                    self._conv_wgrad(x, cw, out, ups, pre, bias is not None)
                if need_dx:
                    self._conv_dgrad(x, cw, out, ups, pre, relu_in)
            self.tape.append(bwd)
            if cw not in self.touched_convs:
                self.touched_convs.append(cw)
        return out

    def _fp32_weights(self, p, pack):
        """p.w of a complete parameter block whose `w` is still the placeholder (= w_split): the fp32 layout `pack()` only if the kernel this call runs reads
        it (mrfa_conv2d_reads_fp32_weights, v9).  The patch-tiled, lean and row-tiled split-operand kernels read the pre-split planes only, so for most of
        the decoder's ~100 M parameters the fp32 forward / data-gradient layouts are never built -- and never refreshed by the per-step PackPlan (8 of its
        28 bytes per parameter; the refresh runs beside the keypoint encoder's forward and slowed it by what it moved)"""
        ph = p.w_split or p.w_phase
        if not ph or p.w != ph:
            return                                    # (no pre-split planes: conv() / _conv_dgrad() put the fp32 layout there already)
        if not (LAZY_FP32_PACKS and not self.L.mrfa_conv2d_reads_fp32_weights(C.byref(p))):
            p.w = pack().data_ptr()

    def _conv_out(self, p, out: Optional[View], N, Ho, Wo, Cout) -> View:
        """the output of a forward convolution whose parameter block is complete but for y.  Launches that split K (the low-resolution levels: too few output
        tiles for 256 CUs) finish inside the launch (mrfa_conv_params.sk_ticket, v8): their tickets and -- when the output is ours to allocate -- the output
        itself come from the zero-filled arena, so that neither the y = bias pass before nor the affine / residual / ReLU / statistics pass behind such a launch
        exists (107 launches per training step on the serial low-resolution chains of the hourglasses and the generator's bottleneck)."""
        rows = N * Ho * Wo
        if FUSED_SPLITK and rows <= 32768:
            p.sk_ticket = p.x                         # (placeholder for the query, which only asks whether tickets will be there: ADVICE r5 -- the ticket
            if self.L.mrfa_conv2d_split_k(C.byref(p)) > 1:      # block is carved from the zero arena only for launches that do split)
                p.sk_ticket = self.pool32.take(((rows + 31) // 32) * ((Cout + 31) // 32)).data_ptr()
                if out is None:
                    ld = _r4(Cout)
                    out = View(Storage(self.pool32.take(rows * ld).view(rows, ld)), N, Ho, Wo, Cout)
                    if self.record and not self.in_backward:
                        self._register(out.st)
                    p.y_zero = 1
            else:
                p.sk_ticket = None
        out = out or self.new(N, Ho, Wo, Cout)
        p.y, p.ldy = out.ptr, out.ld
        return out

    def _defer_ok(self, cw: ConvW) -> bool:
        """this weight gradient may go to the deferred side chain: a collection is active and the gradient is accumulated straight
        into .grad (cw._direct, set by grad_acc): nothing hands it to autograd before the join"""
        return self.wdefer is not None and cw._direct

    def _conv_wgrad(self, x: View, cw: ConvW, out: View, ups, pre, has_bias):
        dw, db = cw.grad_acc(self.pool32)
        q = hip.WgradParams()
        q.x, q.ldx, q.Hin, q.Win, q.ups, q.N, q.Cin = x.ptr, x.ld, x.H, x.W, int(ups), x.N, cw.Cin
        if pre is not None:
            q.in_scale, q.in_shift, q.in_relu = pre[0].data_ptr(), pre[1].data_ptr(), 1
            q.groups = pre[2].get("groups", 1)
        q.dy, q.ldy, q.Cout, q.Hout, q.Wout = out.gptr, out.ld, cw.Cout, out.H, out.W
        q.R, q.S, q.pad = cw.R, cw.S, cw.pad
        q.dw = dw.data_ptr()
        q.dbias = db.data_ptr() if (has_bias and db is not None) else None
        q.alpha, q.nbatch, q.ksplit = 1.0, 1, 0
        if cw.wgrad_flat:
            q.ktab, q.kflat = cw.ktab_fwd().data_ptr(), cw.T * cw.Cin
        ws = Ctx._wgrad_ws.get((self.dev, self.s))      # per stream: concurrent passes must not share the scratch
        if ws is None:
            ws = Ctx._wgrad_ws[(self.dev, self.s)] = torch.empty(16 << 20, dtype=torch.float32, device=self.dev)      # 64 MiB scratch
        q.ws, q.ws_bytes = ws.data_ptr(), ws.numel() * 4
        prof = Ctx.profile
        if prof is None and self._defer_ok(cw):
            keep = (x, out, pre, dw, db)                   # alive until DeferredWgrads.join()

            def launch():
                st = hip.stream_ptr()
                w2 = Ctx._wgrad_ws.get((self.dev, st))
                if w2 is None:
                    w2 = Ctx._wgrad_ws[(self.dev, st)] = torch.empty(16 << 20, dtype=torch.float32, device=self.dev)
                q.ws, q.ws_bytes = w2.data_ptr(), w2.numel() * 4
                assert keep[0].st.data is not None
                self._chk(self.L.mrfa_conv2d_wgrad_nhwc(st, C.byref(q)), "wgrad(deferred)")
            launch.params = q                              # DeferredWgrads batches such launches (mrfa_conv2d_wgrad_multi)
            launch.keep = keep
            self.wdefer.add(launch)
        elif prof is None and WGRAD_STREAM:
            main = torch.cuda.current_stream(self.dev)
            if Ctx._side is None or Ctx._side.device != self.dev:
                Ctx._side = torch.cuda.Stream(device=self.dev)
            side = Ctx._side
            side.wait_stream(main)                    # dY, the zeroed accumulators and everything before are ordered first
            with torch.cuda.stream(side):
                self._chk(self.L.mrfa_conv2d_wgrad_nhwc(hip.stream_ptr(), C.byref(q)), "wgrad")
            self.side_used = True
        elif prof is None:
            self._chk(self.L.mrfa_conv2d_wgrad_nhwc(self.s, C.byref(q)), "wgrad")
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._chk(self.L.mrfa_conv2d_wgrad_nhwc(self.s, C.byref(q)), "wgrad")
            e1.record()
            flops = 2.0 * q.N * q.Hout * q.Wout * q.Cout * q.Cin * q.R * q.S
            prof.append((-1, flops, e0, e1, f"wgrad {q.Cin}->{q.Cout} {q.R}x{q.S} @{q.Hout}x{q.Wout} N={q.N} ups={q.ups} flat={int(q.kflat > 0)} ldx={q.ldx} ldy={q.ldy} xal={x.ptr % 16} dyal={out.gptr % 16} bias={int(bool(q.dbias))} pre={int(pre is not None)}"))

    def _relu_mask_pass(self, x: View):
        """x.grad *= (x > 0): the ReLU backward of x's producers as a pass of its own (where no kernel fuses it)"""
        self._chk(self.L.mrfa_act_bwd(self.s, x.ptr, x.ld, x.gptr, x.ld, x.rows, x.C, 1, x.gptr, x.ld, 0), "relu_bwd(consumer)")

    def _conv_dgrad(self, x: View, cw: ConvW, out: View, ups, pre, relu_in=False):
        """x.grad += conv_transpose(out.grad); with ups the hi-res gradient is sum-pooled; with pre it is pushed
        through the pre-activation BN+ReLU by the caller-registered closure (see prebn).  relu_in (see conv): this is the only
        consumer of the ReLU outputs in x, whose mask (x > 0) is applied here."""
        Hv, Wv = (x.H * 2, x.W * 2) if ups else (x.H, x.W)
        direct = (not ups) and pre is None
        assert direct or not relu_in
        first = direct and self._claim(x)             # first writer of x.grad covering all of it: overwrite, no zero fill needed
        if direct and cw.fewin and out.ld % 4 == 0 and out.coff % 4 == 0:
            # few input channels: the data gradient is a few-output conv over dY
            self._chk(self.L.mrfa_conv_fewout_fwd(self.s, out.gptr, out.ld, out.N, out.H, out.W, cw.Cout, cw.fewin_dgrad_pack().data_ptr(),
                                                  None, x.gptr, x.ld, cw.Cin, cw.R, cw.R - 1 - cw.pad, 0 if first else 1), "conv_fewout(dgrad)")
            if relu_in:
                self._relu_mask_pass(x)
            return
        if (direct and cw.fewout and x.coff % 4 == 0 and
                self.L.mrfa_conv_fewout_dgrad_supported(cw.Cin, cw.Cout, cw.R, cw.pad, x.W, x.ld)):
            # 3x3 layer with one or two output channels: channel-lane kernel (csrc/conv_fewout3.hip) instead of a K = 9 Cout MFMA GEMM
            self._chk(self.L.mrfa_conv_fewout_dgrad(self.s, out.gptr, out.ld, out.N, out.H, out.W, cw.Cout, cw.fewout_pack().data_ptr(),
                                                    x.gptr, x.ld, cw.Cin, cw.R, cw.pad, 0 if first else 1,
                                                    x.ptr if relu_in else None, x.ld), "conv_fewout_dgrad")
            return
        if (ups and pre is None and self.split and PHASE_UPCONV and cw.R == 3 and cw.S == 3 and cw.pad == 1 and not cw.dgrad_flat
                and x.coff % 4 == 0):
            # UpBlock2d: data gradient in phase form (four transposed 2x2 convolutions of the phase images of dY, csrc/conv_halo.hip MODE 2)
            # straight into x.grad -- instead of the 3x3 data gradient on the 2H x 2W grid + the 2x2 sum-pooling pass
            q = hip.ConvParams()
            q.x, q.ldx, q.Hin, q.Win, q.ups, q.N, q.Cin = out.gptr, out.ld, out.H, out.W, 2, out.N, cw.Cout
            cipd = (cw.Cin + 127) // 128 * 128
            wph, q.w_phase_piece = cw.phase_pack(dgrad=True)
            q.w_phase = wph.data_ptr()
            q.w = q.w_phase                                    # (placeholder: the phase kernel reads w_phase only; the ABI's specification reads w: _fp32_weights)
            q.w_ld, q.w_tap, q.kflat, q.w_rows = cw.Cout, cipd * cw.Cout, 0, cipd
            q.Cout, q.Hout, q.Wout = cw.Cin, x.H, x.W
            q.R, q.S, q.pad = cw.R, cw.S, cw.R - 1 - cw.pad
            q.alpha, q.nbatch = 1.0, 1
            if self.L.mrfa_conv2d_phase_dgrad_supported(C.byref(q)):
                first_ph = self._claim(x)
                q.y, q.ldy = x.gptr, x.ld
                q.accumulate = 0 if first_ph else 1
                self._fp32_weights(q, lambda: cw.dgrad_pack(False))
                self._launch_conv(q, "dgrad(phase)", cw.Cout)
                return
        tgt = x if direct else self.new(x.N, Hv, Wv, cw.Cin)
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = out.gptr, out.ld, out.H, out.W, 0, out.N, cw.Cout
        co32 = (cw.Cout + 31) // 32 * 32
        # Cout % 32 != 0 but the dY view sits in a wider (zero-initialised, finite) gradient buffer: read it as co32
        # channels against zero-padded weights instead of taking the scalar-gather flat path
        padded = cw.dgrad_flat and out.coff % 4 == 0 and out.ld % 4 == 0 and out.coff + co32 <= out.ld
        cip = (cw.Cin + 127) // 128 * 128
        if padded:
            p.Cin = co32
            p.w_ld, p.w_tap, p.kflat = co32, cip * co32, 0
            if self.split or self.bf16:
                ws, p.w_piece = cw.split_pack("d", True, rne=self.bf16)
                p.w_split = ws.data_ptr()
        elif cw.dgrad_flat:
            kp = (cw.T * cw.Cout + 31) // 32 * 32
            p.w_ld, p.w_tap, p.kflat = kp, 0, cw.T * cw.Cout
            p.ktab = cw.ktab_dgrad().data_ptr()
        else:
            p.w_ld, p.w_tap, p.kflat = cw.Cout, cip * cw.Cout, 0
            if self.split or self.bf16:
                ws, p.w_piece = cw.split_pack("d", False, rne=self.bf16)
                p.w_split = ws.data_ptr()
        p.w_rows = cip
        p.w = p.w_split if p.w_split else cw.dgrad_pack(padded).data_ptr()       # (placeholder: _fp32_weights below)
        p.y, p.ldy = (tgt.gptr if direct else tgt.ptr), tgt.ld
        p.Cout, p.Hout, p.Wout = cw.Cin, Hv, Wv
        p.R, p.S, p.pad = cw.R, cw.S, cw.R - 1 - cw.pad
        p.alpha, p.nbatch = 1.0, 1
        p.accumulate = 1 if (direct and not first) else 0
        fused = False
        if relu_in:
            p.mask, p.ldm = x.ptr, x.ld
            fused = bool(self.L.mrfa_conv2d_mask_supported(C.byref(p)))
            if not fused:
                p.mask, p.ldm = None, 0
        hint = getattr(x.st, "bn_hint", None) if (direct and BN_BWD_IN_DGRAD) else None
        if pre is not None and not ups and BN_BWD_IN_DGRAD:
            hint = pre[2].get("hint")                  # (prebn(): this launch is the only writer of d(relu(bn(x))), the fresh buffer `tgt`)
        if hint is not None and hint["red"] is None and not relu_in and x.coff == 0 and x.C == x.st.ld and not padded and not cw.dgrad_flat:
            # x is the output of a BatchNorm + activation whose ONLY consumer is this convolution (bn_act(out_sole=True)): this launch is the only writer
            # of its gradient, so the first phase of that BatchNorm's backward -- the per-channel sums of du and du * xhat -- rides in its epilogue
            # (mrfa_conv_params.bst_*) where the library has it; _bn_bwd then runs phase 2 only.  One launch less per layer on the encoder's backward chains.
            bx = hint["x"]
            red = self.f64z(hint["groups"] * hip.STATS_SLOTS * 2 * x.C + 2)
            p.stats, p.groups = red.data_ptr(), hint["groups"]
            p.bst_x, p.bst_ldx, p.bst_relu = bx.ptr, bx.ld, int(hint["relu"])
            p.bst_scale, p.bst_shift = hint["scale"].data_ptr(), hint["shift"].data_ptr()
            p.bst_mean, p.bst_invstd = hint["mean"].data_ptr(), hint["invstd"].data_ptr()
            if self.L.mrfa_conv2d_bwdstats_supported(C.byref(p)) and self.L.mrfa_conv2d_groups_supported(C.byref(p)):
                hint["red"] = red
            else:
                p.stats = p.bst_x = p.bst_scale = p.bst_shift = p.bst_mean = p.bst_invstd = None
                p.groups = 0
        self._fp32_weights(p, lambda: cw.dgrad_pack(padded))
        self._launch_conv(p, "dgrad", cw.Cout)
        if relu_in and not fused:
            self._relu_mask_pass(x)
        if direct:
            return
        cur = tgt            # holds d(pre(ups(x))) as DATA
        if pre is not None:
            # gradient through relu(scale*x+shift) is handled by the BN closure: it reads pre[2] (stash)
            pre[2]["dz"] = cur
            pre[2]["ups"] = ups
            return
        # ups only
        self._chk(self.L.mrfa_sumpool2_acc(self.s, cur.ptr, cur.ld, x.N, x.H, x.W, x.C, x.gptr, x.ld, 1.0), "sumpool2")

    # -- batch norm -------------------------------------------------------------------------------------------
    @staticmethod
    def _sync_world(bn) -> int:
        """>1 when `bn` is a SyncBatchNorm (torch.nn.SyncBatchNorm.convert_sync_batchnorm, reference train.py:43) in an
        initialised process group: batch statistics are then reduced over all ranks (RCCL all-reduce of 2C doubles)."""
        if isinstance(bn, torch.nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_world_size()
        return 1

    @staticmethod
    def _sync_collective(world: int) -> bool:
        """issue the statistics exchange?  Always with > 1 rank; MRFA_SYNCBN_FORCE_COLLECTIVE=1 also issues it in a one-rank group (a sum over
        one rank: the identity) so that the captured-collective schedule can be exercised on a single-GPU box"""
        return world > 1 or (SYNCBN_FORCE and torch.distributed.is_available() and torch.distributed.is_initialized())

    def _bn_finalize(self, bn, stats, count):
        Cn = bn.num_features
        done = self._fin_done.pop(stats.data_ptr(), None) if stats is not None else None
        if done is not None:                              # finished inside the producing convolution's call (_fin_params)
            assert done[0] is bn and done[1] == count, "fused BatchNorm finalize: another layer / row count than the convolution was told"
            scale, shift, mean, invstd = done[2:]
            self.nbt[bn] = self.nbt.get(bn, 0) + self.groups
            return scale, shift, mean, invstd
        train = self.train
        G = self.groups if (train and stats is not None) else 1      # statistic groups: [G][C] results, `count` = rows of the whole batch
        scale, shift, mean, invstd = self.f32(G * Cn), self.f32(G * Cn), self.f32(G * Cn), self.f32(G * Cn)
        if train and stats is not None:
            world = self._sync_world(bn)
            if stats.data_ptr() in self._synced:          # exchanged by sync_stats() together with the layers beside it
                count = count * world
            elif self._sync_collective(world) and isinstance(bn, torch.nn.SyncBatchNorm):
                # sum / sum-of-squares over every rank's pixels.  Small layers: ALL slots (of all statistic groups: one collective for the source and the
                # driving pass) are all-reduced in place (<= 128 KB: the collective is latency-bound
                # there, and nothing but the collective is launched -- round 3 summed, zeroed and copied: three glue launches per layer and direction,
                # ~1 200 per step with the MTIA prior); wide layers: the slots are summed locally first (one launch) so that the message is 2C doubles
                nsl = hip.STATS_SLOTS * 2 * Cn
                if G * nsl * 8 <= SYNCBN_DIRECT_BYTES:
                    _syncbn_all_reduce(stats[:G * nsl])
                else:
                    stats = self._allreduce_slot_sums(stats, G, Cn)
                count = count * world
        if G > 1:
            self._chk(self.L.mrfa_bn_finalize_groups(self.s, stats.data_ptr(), count // G, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                     bn.running_mean.data_ptr(), bn.running_var.data_ptr(), BN_MOMENTUM, BN_EPS, Cn, G,
                                                     scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()), "bn_finalize_groups")
        else:
            self._chk(self.L.mrfa_bn_finalize(self.s, stats.data_ptr() if stats is not None else None, count, bn.weight.data_ptr(),
                                              bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                              BN_MOMENTUM, BN_EPS, Cn, int(train), scale.data_ptr(), shift.data_ptr(),
                                              mean.data_ptr(), invstd.data_ptr()), "bn_finalize")
        if train:
            self.nbt[bn] = self.nbt.get(bn, 0) + G     # num_batches_tracked += 1 per statistic group, batched in flush_forward()
        return scale, shift, mean, invstd

    def syncbn_lockstep(self, bn) -> bool:
        """should the caller walk independent SyncBatchNorm layers side by side (all convolutions, sync_stats(), all BatchNorm applications)?"""
        return bool(SYNCBN_LOCKSTEP and self.train and isinstance(bn, torch.nn.SyncBatchNorm) and self._sync_collective(self._sync_world(bn)))

    def sync_stats(self, items):
        """items = [(bn, stats)] of SyncBatchNorm layers that do not depend on each other, statistics accumulated (conv_bn_raw), not yet applied: ONE
        statistics collective for all of them now, and ONE for the sums of their backward passes -- the bn_act() backward closures of the group run their
        first phase as they come and the last one exchanges all the sums and runs every second phase (the residual gradient is a first-phase output, the
        data gradient of a member is only read by its convolution's closure, which the caller recorded before any bn_act of the group)."""
        if not items or not self.syncbn_lockstep(items[0][0]):
            return
        grp = {"left": 0, "pending": [], "red": {}} if self.record else None
        blocks = []
        for bn, st in items:
            nsl = self.groups * hip.STATS_SLOTS * 2 * bn.num_features
            if st is None or nsl * 8 > SYNCBN_DIRECT_BYTES:
                continue                                   # (wide layers sum their slots first: on their own, _bn_finalize)
            blocks.append(st)
            self._synced[st.data_ptr()] = grp
            if grp is not None:
                grp["left"] += 1
                grp["red"][st.data_ptr()] = self.f64z(nsl + 2)      # the members' backward sums: carved now, back to back (their closures do not run in a row)
        _syncbn_all_reduce_many(blocks)

    def _bn_group_done(self, grp):
        """one member of a sync_stats() group has run (or skipped) the first phase of its backward; the last one finishes all of them"""
        grp["left"] -= 1
        if grp["left"] > 0 or not grp["pending"]:
            return
        _syncbn_all_reduce_many([red for _, red in grp["pending"]])
        for q, _ in grp["pending"]:
            self._chk(self.L.mrfa_bn_act_bwd(self.s, C.byref(q)), "bn_act_bwd(2)")
        grp["pending"] = []

    def _allreduce_slot_sums(self, slots: torch.Tensor, G: int, Cn: int) -> torch.Tensor:
        """wide layers under SyncBatchNorm: the [G][STATS_SLOTS][2C] blocks are summed over the slots locally, the [G][2C] sums are all-reduced (a message of
        G x 2C doubles instead of 32 times that) and returned as slot 0 of a fresh zeroed [G][STATS_SLOTS][2C] block"""
        S = hip.STATS_SLOTS
        local = torch.sum(slots[:G * S * 2 * Cn].view(G, S, 2 * Cn), 1)
        _syncbn_all_reduce(local)
        out = self.f64z(G * S * 2 * Cn)
        out.view(G, S, 2 * Cn)[:, 0].copy_(local)
        return out

    def flush_forward(self):
        """end of a program's forward: one multi-tensor launch for all BatchNorm batch counters instead of one each"""
        by_count = {}
        for bn, k in self.nbt.items():
            by_count.setdefault(k, []).append(bn.num_batches_tracked)
        for k, ts in by_count.items():
            torch._foreach_add_(ts, k)
        self.nbt = {}

    def bn_stats_buf(self, bn):
        """[groups][STATS_SLOTS][2C] zeroed doubles (+ FIN_WORDS zeroed words behind them: the ticket counters of a finalize fused into the producing launch)"""
        return self.f64z(self.groups * hip.STATS_SLOTS * 2 * bn.num_features + hip.FIN_WORDS // 2) if self.train else None

    def _bn_stats_into(self, x: View, stats: torch.Tensor):
        """per-channel sum / sum of squares of x into `stats` by a pass of its own (one per statistic group: a group is a run of consecutive rows)"""
        G = self.groups if self.train else 1
        rows = x.rows // G
        for g in range(G):
            self._chk(self.L.mrfa_bn_stats(self.s, x.ptr + 4 * g * rows * x.ld, x.ld, rows, x.C, stats.data_ptr() + 8 * g * hip.STATS_SLOTS * 2 * x.C),
                      "bn_stats")

    @staticmethod
    def fin(bn):
        """the `fin=` argument of conv() for a convolution whose statistics `bn` consumes (None: the finalize stays a launch of its own)"""
        return bn if BN_FIN_FUSED else None

    def _fin_params(self, p, bn, stats, count: int):
        """BatchNorm finalize inside the convolution launch that accumulates `stats` (mrfa_conv_params.fin_*): the small-problem kernel's last
        workgroup does it, every other kernel is followed by the finalize launch inside the call -- bn_act() then finds the result here
        instead of launching mrfa_bn_finalize (366 launches per training step on the keypoint encoder's forward chains)."""
        if not self.train or stats is None:
            return
        if isinstance(bn, torch.nn.SyncBatchNorm) and self._sync_collective(self._sync_world(bn)):
            return                                        # the statistics are exchanged between the convolution and the finalize
        Cn, G = bn.num_features, self.groups
        scale, shift, mean, invstd = self.f32(G * Cn), self.f32(G * Cn), self.f32(G * Cn), self.f32(G * Cn)
        p.fin_gamma, p.fin_beta = bn.weight.data_ptr(), bn.bias.data_ptr()
        p.fin_rmean, p.fin_rvar = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        p.fin_momentum, p.fin_eps, p.fin_count = BN_MOMENTUM, BN_EPS, count // G
        p.fin_scale, p.fin_shift, p.fin_mean, p.fin_invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
        p.fin_counter = stats.data_ptr() + 8 * G * hip.STATS_SLOTS * 2 * Cn
        self._fin_done[stats.data_ptr()] = (bn, count, scale, shift, mean, invstd)

    def bn_act(self, x: View, bn, stats, *, relu=True, pool=False, blend=None, out: Optional[View] = None,
               sole_consumer: bool = False, res: Optional[View] = None, out_sole: bool = False) -> View:
        """out = [blend_a*occ +] act(bn(x)) [*(1-occ)], optional 2x2 avg-pool.  x = raw conv output, stats = its
        epilogue-accumulated sums (train) or None (eval)."""
        scale, shift, mean, invstd = self._bn_finalize(bn, stats, x.rows)
        Ho, Wo = (x.H // 2, x.W // 2) if pool else (x.H, x.W)
        out = out or self.new(x.N, Ho, Wo, x.C)
        p = hip.BnActParams()
        p.x, p.ldx, p.N, p.H, p.W, p.C = x.ptr, x.ld, x.N, x.H, x.W, x.C
        p.scale, p.shift, p.relu, p.pool = scale.data_ptr(), shift.data_ptr(), int(relu), int(pool)
        p.groups = self.groups if (self.train and stats is not None) else 1
        if blend is not None:
            a, occ = blend
            p.blend_a, p.lda, p.occ, p.ldo = a.ptr, a.ld, occ.ptr, occ.ld
        if res is not None:                           # y = act(bn(x) + res): HRNet residual blocks
            assert (res.N, res.H, res.W, res.C) == (x.N, x.H, x.W, x.C) and not pool and blend is None
            p.res, p.ldr = res.ptr, res.ld
        p.y, p.ldy = out.ptr, out.ld
        self._chk(self.L.mrfa_bn_act_fwd(self.s, C.byref(p)), "bn_act_fwd")
        if self.record:
            train = self.train
            if sole_consumer and x.coff == 0 and x.ld == x.C and x.st.grad is None and not self.in_backward:
                x.st.grad_noinit = True               # this BN's backward is the only writer of x.grad and covers all of it

            hint = None
            if (out_sole and train and not pool and blend is None and res is None and out.coff == 0 and out.C == out.st.ld
                    and not (isinstance(bn, torch.nn.SyncBatchNorm) and self._sync_collective(self._sync_world(bn)))):
                # the caller states that ONE convolution consumes `out`: its data gradient may carry this BatchNorm's first backward phase (_conv_dgrad)
                hint = out.st.bn_hint = {"x": x, "scale": scale, "shift": shift, "mean": mean, "invstd": invstd, "relu": relu, "red": None,
                                         "groups": p.groups}

            groups = p.groups
            grp = self._synced.pop(stats.data_ptr(), None) if stats is not None else None       # sync_stats(): the backward sums travel together too
            if grp is not None:
                grp = (grp, grp["red"].pop(stats.data_ptr()))

            def bwd():
                if not out.has_grad:
                    if grp is not None:
                        self._bn_group_done(grp[0])
                    return
                self._bn_bwd(x, bn, scale, shift, mean, invstd, relu, pool, out.gptr, out.ld, blend, train, x, res, hint=hint, groups=groups, group=grp)
            self.tape.append(bwd)
        elif stats is not None:
            self._synced.pop(stats.data_ptr(), None)
        return out

    def _bn_bwd(self, x, bn, scale, shift, mean, invstd, relu, pool, dy_ptr, dy_ld, blend, train, dx_view, res=None, hint=None, groups=1, group=None):
        bg = bngrad(bn)
        if bg not in self.touched_bns:
            self.touched_bns.append(bg)
        dg, db = bg.acc(self.pool32)
        nred = groups * hip.STATS_SLOTS * 2 * x.C          # (statistic groups: [groups][STATS_SLOTS][2C])
        pre_red = hint["red"] if hint is not None else None          # the sums of phase 1, already accumulated by the data gradient that wrote dy
        group, group_red = group if group is not None else (None, None)
        # slotted like the statistics buffers (MRFA_STATS_SLOTS) + the fused launch's barrier word
        red = pre_red if pre_red is not None else (group_red if group_red is not None else self.f64z(nred + 2))
        q = hip.BnBwdParams()
        q.x, q.ldx, q.N, q.H, q.W, q.C = x.ptr, x.ld, x.N, x.H, x.W, x.C
        q.scale, q.shift, q.relu, q.pool = scale.data_ptr(), shift.data_ptr(), int(relu), int(pool)
        q.mean, q.invstd, q.gamma = mean.data_ptr(), invstd.data_ptr(), bn.weight.data_ptr()
        q.dy, q.lddy = dy_ptr, dy_ld
        if blend is not None:
            a, occ = blend
            q.blend_a, q.lda, q.occ, q.ldo = a.ptr, a.ld, occ.ptr, occ.ld
            q.dblend_a, q.ldda, q.docc, q.lddo = a.gptr, a.ld, occ.gptr, occ.ld
        if res is not None:
            q.res, q.ldr, q.dres, q.lddr = res.ptr, res.ld, res.gptr, res.ld
        q.red = red.data_ptr()
        q.dx_overwrite = int(self._claim(dx_view))         # before .gptr, which would zero-fill a fresh buffer
        q.dx, q.lddx = dx_view.gptr, dx_view.ld
        q.dgamma, q.dbeta = dg.data_ptr(), db.data_ptr()
        q.train, q.groups = int(train), groups
        world = self._sync_world(bn) if train else 1
        synced = train and self._sync_collective(world) and isinstance(bn, torch.nn.SyncBatchNorm)
        if pre_red is None:
            q.phase = 1
            self._chk(self.L.mrfa_bn_act_bwd(self.s, C.byref(q)), "bn_act_bwd(1)")
        else:
            q.red_all = 1                                   # the convolution's epilogue spread its sums over all slots
        if synced:
            # SyncBN backward: the batch means of du and du*xhat are global; gamma/beta gradients stay local sums: taken from the local slots by one
            # launch, then the slots are all-reduced in place and phase 2 divides by world x the local row count (mrfa_bnbwd_params.red_world).
            # (Round 3: sum, two adds, the collective and a division = five launches per layer.)
            Cn = x.C
            self._chk(self.L.mrfa_bn_param_grad_groups(self.s, red.data_ptr(), Cn, groups, dg.data_ptr(), db.data_ptr()), "bn_param_grad")
            q.red_world = world
            q.dgamma = q.dbeta = None
            q.phase = 2
            if group is not None:                           # sync_stats(): the sums of the group's members are exchanged together (_bn_group_done)
                group["pending"].append((q, red))
                self._bn_group_done(group)
                return
            if nred * 8 <= SYNCBN_DIRECT_BYTES:
                _syncbn_all_reduce(red[:nred])
            else:
                q.red = self._allreduce_slot_sums(red, groups, Cn).data_ptr()
        elif group is not None:
            self._bn_group_done(group)
        q.phase = 2
        self._chk(self.L.mrfa_bn_act_bwd(self.s, C.byref(q)), "bn_act_bwd(2)")

    def prebn(self, x: View, bn, stats=None):
        """Pre-activation BN+ReLU folded into the next conv's prologue (ResBlock2d / ChannelBlock2d).  Returns the `pre`
        triple for conv(); must be called BEFORE that conv so the tape order is right (its closure runs AFTER the conv's)."""
        assert self.groups == 1 or stats is not None, "pre-activation BatchNorm with statistic groups needs the producing convolution's grouped statistics"
        if self.train and stats is None:
            stats = self.f64z(hip.STATS_SLOTS * 2 * x.C)
            self._chk(self.L.mrfa_bn_stats(self.s, x.ptr, x.ld, x.rows, x.C, stats.data_ptr()), "bn_stats")
        scale, shift, mean, invstd = self._bn_finalize(bn, stats, x.rows)
        groups = self.groups if (self.train and stats is not None) else 1
        stash = {"groups": groups}
        if self.record:
            train = self.train
            if train and not (isinstance(bn, torch.nn.SyncBatchNorm) and self._sync_collective(self._sync_world(bn))):
                # the consuming convolution's data gradient writes d(relu(bn(x))) into a buffer of its own: the first phase of this BatchNorm's backward may
                # ride in its epilogue (_conv_dgrad), as for bn_act(out_sole=True)
                stash["hint"] = {"x": x, "scale": scale, "shift": shift, "mean": mean, "invstd": invstd, "relu": True, "red": None, "groups": groups}

            def bwd():
                dz = stash.get("dz")
                if dz is None:
                    return
                assert not stash.get("ups", False)
                self._bn_bwd(x, bn, scale, shift, mean, invstd, True, False, dz.ptr, dz.ld, None, train, x, hint=stash.get("hint"), groups=groups)
            self.tape.append(bwd)
        return (scale, shift, stash)

    def prologue_ok(self, x: View, conv) -> bool:
        """may the BatchNorm-apply + ReLU between two convolutions of a residual block ride in the second one's prologue (forward, data gradient and weight
        gradient read the first one's raw output; the bn_act launch and its tensor disappear)?  Training only; with statistic groups only where the library
        keeps one prologue vector pair per group (conv_lean.hip / wgrad_lean.hip: the keypoint encoder's 3x3 layers)"""
        if not (self.train and PROLOGUE_FUSION and (self.split or self.bf16)):
            return False
        cw = convw(conv)
        if cw.stride != 1 or cw.fwd_flat or cw.dgrad_flat or cw.wgrad_flat or x.coff != 0 or x.C != x.st.ld:
            return False
        if self.groups <= 1:
            return True
        cop = (cw.Cout + 127) // 128 * 128
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.N, p.Cin = x.ptr, x.ld, x.H, x.W, x.N, cw.Cin
        p.w = p.w_split = p.in_scale = p.in_shift = p.y = x.ptr           # (placeholders of the right alignment for the query)
        p.w_ld, p.w_tap, p.w_rows, p.w_piece = cw.Cin, cop * cw.Cin, cop, (0 if self.bf16 else cw.T * cop * cw.Cin)
        p.in_relu, p.ldy, p.Cout, p.Hout, p.Wout = 1, _r4(cw.Cout), cw.Cout, x.H + 2 * cw.pad - cw.R + 1, x.W + 2 * cw.pad - cw.S + 1
        p.R, p.S, p.pad, p.alpha, p.nbatch, p.groups = cw.R, cw.S, cw.pad, 1.0, 1, self.groups
        q = hip.WgradParams()
        q.x, q.ldx, q.Hin, q.Win, q.N, q.Cin = x.ptr, x.ld, x.H, x.W, x.N, cw.Cin
        q.in_scale = q.in_shift = q.dy = q.dw = x.ptr
        q.in_relu, q.ldy, q.Cout, q.Hout, q.Wout, q.R, q.S, q.pad = 1, _r4(cw.Cout), cw.Cout, p.Hout, p.Wout, cw.R, cw.S, cw.pad
        q.alpha, q.nbatch, q.groups = 1.0, 1, self.groups
        return bool(self.L.mrfa_conv2d_groups_supported(C.byref(p)) and self.L.mrfa_conv2d_wgrad_groups_supported(C.byref(q)))

    # -- samplers ---------------------------------------------------------------------------------------------
    def grid_sample(self, inp: View, grid: View, mode: int, out: Optional[View] = None, in_rep: int = 1, need_din=True,
                    need_dgrid=True) -> View:
        n_out = grid.N
        assert grid.C == 2 and n_out == inp.N * in_rep
        out = out or self.new(n_out, grid.H, grid.W, inp.C)
        bs = inp.H * inp.W * inp.ld
        self._chk(self.L.mrfa_grid_sample_fwd(self.s, inp.ptr, inp.ld, bs, in_rep, inp.H, inp.W, inp.C, grid.ptr, grid.ld, n_out,
                                              grid.H, grid.W, out.ptr, out.ld, mode), "grid_sample_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_grid_sample_bwd(self.s, inp.ptr, inp.ld, bs, in_rep, inp.H, inp.W, inp.C, grid.ptr, grid.ld, n_out,
                                                      grid.H, grid.W, out.gptr, out.ld, mode,
                                                      inp.gptr if need_din else None, inp.ld, bs,
                                                      grid.gptr if need_dgrid else None, grid.ld), "grid_sample_bwd")
            self.tape.append(bwd)
        return out

    def resize(self, x: View, Ho: int, Wo: int, mul: float = 1.0, out: Optional[View] = None, acc: bool = False) -> View:
        out = out or self.new(x.N, Ho, Wo, x.C, pad32=x.zpad)
        if self._rs is not None:
            return self._rs_record(x, out, mul, acc)
        self._chk(self.L.mrfa_resize_bilinear_fwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, x.C, out.ptr, out.ld, Ho, Wo, mul, int(acc)),
                  "resize_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_resize_bilinear_bwd(self.s, out.gptr, out.ld, x.N, x.H, x.W, x.C, x.gptr, x.ld, Ho, Wo, mul),
                          "resize_bwd")
            self.tape.append(bwd)
        return out

    # -- many copies / resizes in one launch (mrfa_resize_sum_multi, v8) ------------------------------------------------------------------------------
    @contextlib.contextmanager
    def fused_resizes(self):
        """the copy() / resize() calls issued inside are INDEPENDENT of each other except for accumulation chains into one output (out = resize(a);
        resize(b, out=out, acc=True); ...): they are recorded and issued as ONE launch when the block ends (one thread per output element evaluates its chain
        in call order: the arithmetic of the separate launches), and their backward passes as ONE launch grouped by input (down-sampling terms keep
        their scatter launches).  RaftFlow's running-flow updates and its flow / occlusion re-composition between two refinement levels (raft.py:258-262,
        276-295) are ~16 launches over 1- and 2-channel maps per level on a chain where nothing else runs."""
        if not FUSED_RESIZES or self._rs is not None:
            yield
            return
        self._rs = []
        try:
            yield
        finally:
            ops, self._rs = self._rs, None
        if ops:
            self._rs_flush(ops)

    @staticmethod
    def _rs_key(v: View):
        return (id(v.st), v.coff, v.C)

    def _rs_record(self, x: View, out: View, mul: float, acc: bool, nograd: bool = False) -> View:
        assert (x.N, x.C) == (out.N, out.C)
        if any(self._rs_key(x) == self._rs_key(o) for _, o, _, _, _ in self._rs):
            ops, self._rs = self._rs, []                      # reads what an earlier call of the block writes: that part goes first
            self._rs_flush(ops)
        self._rs.append((x, out, float(mul), bool(acc), bool(nograd)))
        return out

    def _rs_launch(self, recs, bwd: bool):
        """recs: [(dst view, is gradient, overwrite, [(src view, is gradient, mul)])] -> launches of <= 4 terms per record (a longer chain continues in the next launch)"""
        T = hip.RESIZE_SUM_TERMS
        while recs:
            descs, rest = [], []
            for dst, dgrad, ow, terms in recs:
                d = hip.ResizeSumDesc()
                d.dst, d.ldd, d.N, d.Hd, d.Wd, d.C = (dst.gptr if dgrad else dst.ptr), dst.ld, dst.N, dst.H, dst.W, dst.C
                d.nterm, d.overwrite = min(len(terms), T), int(ow)
                for k, (src, sgrad, mul) in enumerate(terms[:T]):
                    d.term[k].src, d.term[k].lds, d.term[k].Hs, d.term[k].Ws, d.term[k].mul = (src.gptr if sgrad else src.ptr), src.ld, src.H, src.W, mul
                descs.append(d)
                if len(terms) > T:
                    rest.append((dst, dgrad, False, terms[T:]))
            table = (hip.ResizeSumDesc * len(descs))(*descs)
            fn = self.L.mrfa_resize_sum_multi_bwd if bwd else self.L.mrfa_resize_sum_multi
            self._chk(fn(self.s, table, len(descs)), "resize_sum_multi" + ("_bwd" if bwd else ""))
            recs = rest

    def _rs_flush(self, ops):
        groups = {}
        for x, out, mul, acc, _ in ops:                       # by output, in call order
            g = groups.setdefault(self._rs_key(out), [out, not acc, []])
            g[2].append((x, False, mul))
        self._rs_launch([(out, False, ow, terms) for out, ow, terms in groups.values()], bwd=False)
        if not self.record:
            return

        def bwd():
            by_in = {}
            for x, out, mul, acc, nograd in reversed(ops):    # the order the separate backward closures ran in
                if nograd or not out.has_grad:
                    continue
                if out.H >= x.H and out.W >= x.W:
                    by_in.setdefault(self._rs_key(x), [x, []])[1].append((out, True, mul))
                else:                                         # down-sampling: the scatter form, a launch of its own
                    self._chk(self.L.mrfa_resize_bilinear_bwd(self.s, out.gptr, out.ld, x.N, x.H, x.W, x.C, x.gptr, x.ld, out.H, out.W, mul), "resize_bwd")
            if by_in:
                self._rs_launch([(x, True, False, terms) for x, terms in by_in.values()], bwd=True)
        self.tape.append(bwd)

    def corr_lookup(self, vol0: torch.Tensor, vol1: torch.Tensor, dvols, Hs: int, Ws: int, coords: View, radius: int = 3,
                    out: Optional[View] = None) -> View:
        """vol0 (Q,Hs*Ws), vol1 (Q,Hs/2*Ws/2) torch tensors; dvols: callable returning (dvol0, dvol1) grad tensors or None."""
        Q = coords.rows
        nwin = (2 * radius + 1) ** 2
        out = out or self.new(coords.N, coords.H, coords.W, 2 * nwin, pad32=True)
        self._chk(self.L.mrfa_corr_lookup_fwd(self.s, vol0.data_ptr(), vol1.data_ptr(), Hs, Ws, coords.ptr, coords.ld, Q, radius,
                                              out.ptr, out.ld), "corr_lookup_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                d0, d1 = dvols() if dvols is not None else (None, None)
                self._chk(self.L.mrfa_corr_lookup_bwd(self.s, vol0.data_ptr(), vol1.data_ptr(), Hs, Ws, coords.ptr, coords.ld, Q, radius,
                                                      out.gptr, out.ld, d0.data_ptr() if d0 is not None else None,
                                                      d1.data_ptr() if d1 is not None else None, coords.gptr, coords.ld),
                          "corr_lookup_bwd")
            self.tape.append(bwd)
        return out

    # -- prior-motion stage: K14-K17 (csrc/prior.hip) -----------------------------------------------------------------
    def _ext_acc(self, x: torch.Tensor) -> torch.Tensor:
        """zero-initialised fp32 accumulator for the gradient of a module input / parameter that a kernel adds into (delivered
        by the program's in_grads / _ProgramFn.backward exactly like the gradients torch islands compute)"""
        k = id(x)
        g = self.ext_grads.get(k)
        if g is None:
            g = self.ext_grads[k] = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
        elif not g.is_contiguous():
            g = self.ext_grads[k] = g.contiguous()
        return g

    @staticmethod
    def _cf(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        return None if t is None else t.detach().contiguous().float()

    def kp_gaussian(self, kp: torch.Tensor, variance: float, out: View, pos: Optional[torch.Tensor] = None) -> View:
        """out[b,y,x,k] = exp(-|grid(y,x) - kp[b,k]|^2 / (2 variance)) [+ pos[0,k,y,x]]   (util.py:59-87, raft.py:177-178)"""
        kc, pc = self._cf(kp), self._cf(pos)
        B, K = kc.shape[0], kc.shape[1]
        assert out.C == K and out.N == B
        if pc is not None and tuple(pc.shape[-3:]) != (K, out.H, out.W):
            # the kernel indexes pos[(k*H + y)*W + x] with the OUTPUT's H, W (and the backward adds into dpos the same way); the reference
            # raises a broadcast error here (raft.py:177-178 with a `size` that does not match the input resolution)
            raise RuntimeError(f"kp_gaussian: pos_embedding {tuple(pc.shape)} does not match the heat-map view (.., {K}, {out.H}, {out.W})")
        self._chk(self.L.mrfa_kp_gaussian_fwd(self.s, kc.data_ptr(), pc.data_ptr() if pc is not None else None, B, K, out.H, out.W,
                                              float(variance), out.ptr, out.ld), "kp_gaussian_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                dk = self._ext_acc(kp)
                dp = self._ext_acc(pos) if pos is not None else None
                self._chk(self.L.mrfa_kp_gaussian_bwd(self.s, kc.data_ptr(), B, K, out.H, out.W, float(variance), out.gptr, out.ld,
                                                      dk.data_ptr(), dp.data_ptr() if dp is not None else None), "kp_gaussian_bwd")
            self.tape.append(bwd)
        return out

    def prior_motion(self, src: View, kd, ks, jd, js, bg, variance: float):
        """DenseMotionNetwork's heat-map differences, sparse motions and the K+1 warps of the 1/4-scale source in one launch
        (dense_motion.py:36-85, 117-119) -> (motions View (B*K1,h,w,2), hourglass input View (B,h,w,K1*(C+1)), sparse_deformed IslandOut)"""
        B, H, W, Cc = src.N, src.H, src.W, src.C
        K = kd.shape[1]
        K1 = K + 1
        t = [self._cf(x) for x in (kd, ks, jd, js, bg)]
        motions = self.new(B * K1, H, W, 2)
        inp = self.new(B, H, W, K1 * (Cc + 1), pad32=True)
        sparse = IslandOut(torch.empty((B, K1, Cc, H, W), dtype=torch.float32, device=self.dev))
        p = hip.PriorParams()
        p.kd, p.ks = t[0].data_ptr(), t[1].data_ptr()
        p.jd, p.js = (t[2].data_ptr(), t[3].data_ptr()) if t[2] is not None else (None, None)
        p.bg = t[4].data_ptr() if t[4] is not None else None
        p.src, p.lds, p.B, p.K, p.H, p.W, p.C, p.inv_var = src.ptr, src.ld, B, K, H, W, Cc, 1.0 / float(variance)
        p.motions, p.ldm, p.inp, p.ldi, p.sparse = motions.ptr, motions.ld, inp.ptr, inp.ld, sparse.t.data_ptr()
        self._chk(self.L.mrfa_prior_motion_fwd(self.s, C.byref(p)), "prior_motion_fwd")
        if self.record:
            keep = (src, t)            # `p` holds raw pointers only: the source image and the input copies must outlive the forward

            def bwd():
                ds = sparse.total_grad()
                if not (inp.has_grad or motions.has_grad or ds is not None):
                    return
                assert keep[0].st.data is not None
                q = hip.PriorParams()
                C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
                q.dinp, q.lddi = inp.gptr, inp.ld                      # (zero-filled on first touch when no consumer wrote it)
                q.dmotions = motions.gptr if motions.has_grad else None
                dsc = ds.contiguous().float() if ds is not None else None
                q.dsparse = dsc.data_ptr() if dsc is not None else None
                q.dkd, q.dks = self._ext_acc(kd).data_ptr(), self._ext_acc(ks).data_ptr()
                if jd is not None:
                    q.djd, q.djs = self._ext_acc(jd).data_ptr(), self._ext_acc(js).data_ptr()
                if bg is not None:
                    q.dbg = self._ext_acc(bg).data_ptr()
                self._chk(self.L.mrfa_prior_motion_bwd(self.s, C.byref(q)), "prior_motion_bwd")
            self.tape.append(bwd)
        return motions, inp, sparse

    def softmax_combine(self, logit: View, motions: View):
        """mask = softmax_k(logit), deformation = sum_k mask_k motion_k (dense_motion.py:129-136) -> IslandOuts deformation (B,h,w,2),
        mask (B,K1,h,w), logit_mask (B,K1,h,w)"""
        B, H, W, K1 = logit.N, logit.H, logit.W, logit.C
        assert motions.N == B * K1 and motions.C == 2
        mk = lambda *shp: IslandOut(torch.empty(shp, dtype=torch.float32, device=self.dev))
        deform, mask, lg = mk(B, H, W, 2), mk(B, K1, H, W), mk(B, K1, H, W)
        self._chk(self.L.mrfa_softmax_combine_fwd(self.s, logit.ptr, logit.ld, motions.ptr, motions.ld, B, H, W, K1, deform.t.data_ptr(),
                                                  mask.t.data_ptr(), lg.t.data_ptr()), "softmax_combine_fwd")
        if self.record:
            def bwd():
                g = [self._cf(o.total_grad()) for o in (deform, mask, lg)]
                if all(x is None for x in g):
                    return
                ptr = lambda x: x.data_ptr() if x is not None else None
                self._chk(self.L.mrfa_softmax_combine_bwd(self.s, motions.ptr, motions.ld, B, H, W, K1, mask.t.data_ptr(), ptr(g[0]), ptr(g[1]),
                                                          ptr(g[2]), logit.gptr, logit.ld, motions.gptr if g[0] is not None else None),
                          "softmax_combine_bwd")
            self.tape.append(bwd)
        return deform, mask, lg

    def kp_head(self, logits: View, jm: Optional[View], temperature: float):
        """spatial softmax at temperature T + soft-argmax (+ heat-map-weighted Jacobian pooling), kp_detector.py:90-120 ->
        IslandOuts kp (B,K,2) [, jacobian (B,K,2,2)]"""
        B, H, W, K = logits.N, logits.H, logits.W, logits.C
        kp = IslandOut(torch.empty((B, K, 2), dtype=torch.float32, device=self.dev))
        jac = IslandOut(torch.empty((B, K, 2, 2), dtype=torch.float32, device=self.dev)) if jm is not None else None
        stat = torch.empty((B, K, 2), dtype=torch.float32, device=self.dev)
        self._chk(self.L.mrfa_kp_head_fwd(self.s, logits.ptr, logits.ld, jm.ptr if jm is not None else None, jm.ld if jm is not None else 0,
                                          B, H, W, K, float(temperature), kp.t.data_ptr(), jac.t.data_ptr() if jac is not None else None,
                                          stat.data_ptr()), "kp_head_fwd")
        if self.record:
            def bwd():
                dk = self._cf(kp.total_grad())
                dj = self._cf(jac.total_grad()) if jac is not None else None
                if dk is None and dj is None:
                    return
                self._chk(self.L.mrfa_kp_head_bwd(self.s, logits.ptr, logits.ld, jm.ptr if jm is not None else None,
                                                  jm.ld if jm is not None else 0, B, H, W, K, float(temperature), kp.t.data_ptr(),
                                                  jac.t.data_ptr() if jac is not None else None, stat.data_ptr(),
                                                  dk.data_ptr() if dk is not None else None, dj.data_ptr() if dj is not None else None,
                                                  logits.gptr, logits.ld, jm.gptr if (jm is not None and dj is not None) else None,
                                                  jm.ld if jm is not None else 0), "kp_head_bwd")
            self.tape.append(bwd)
        return kp, jac

    # -- elementwise ------------------------------------------------------------------------------------------
    def copy(self, x: View, out: Optional[View] = None, mul: float = 1.0, acc: bool = False, nograd: bool = False) -> View:
        """out (=|+=) mul * x.  nograd: x is a constant (a coordinate grid): nothing goes on the tape for it"""
        out = out or self.new(x.N, x.H, x.W, x.C)
        if self._rs is not None:
            return self._rs_record(x, out, mul, acc, nograd)
        self._chk(self.L.mrfa_copy_view(self.s, x.ptr, x.ld, x.rows, x.C, out.ptr, out.ld, mul, int(acc)), "copy_view")
        if self.record and not nograd:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_copy_view(self.s, out.gptr, out.ld, out.rows, out.C, x.gptr, x.ld, mul, 1), "copy_view_bwd")
            self.tape.append(bwd)
        return out

    def add_const(self, const: View, x: View):
        """x += const in place, const carries no gradient (d/dx is the identity: nothing goes on the tape)"""
        self._chk(self.L.mrfa_copy_view(self.s, const.ptr, const.ld, const.rows, const.C, x.ptr, x.ld, 1.0, 1), "copy_view(+=const)")

    def act(self, x: View, kind: int, out: Optional[View] = None) -> View:
        """kind 1 relu, 2 sigmoid"""
        out = out or self.new(x.N, x.H, x.W, x.C)
        self._chk(self.L.mrfa_bias_act(self.s, x.ptr, x.ld, x.rows, x.C, None, kind, out.ptr, out.ld, None), "act")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_act_bwd(self.s, out.ptr, out.ld, out.gptr, out.ld, out.rows, out.C, kind, x.gptr, x.ld, 1), "act_bwd")
            self.tape.append(bwd)
        return out

    def blend(self, a: View, b: Optional[View], occ: View, out: Optional[View] = None) -> View:
        """out = a*occ + b*(1-occ)   (b None -> a*occ)"""
        out = out or self.new(a.N, a.H, a.W, a.C)
        self._chk(self.L.mrfa_blend_fwd(self.s, a.ptr, a.ld, b.ptr if b else None, b.ld if b else 0, occ.ptr, occ.ld, a.rows, a.C,
                                        out.ptr, out.ld), "blend_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_blend_bwd(self.s, a.ptr, a.ld, b.ptr if b else None, b.ld if b else 0, occ.ptr, occ.ld,
                                                out.gptr, out.ld, a.rows, a.C, a.gptr, a.ld, b.gptr if b else None, b.ld if b else 0,
                                                occ.gptr, occ.ld), "blend_bwd")
            self.tape.append(bwd)
        return out

    def avgpool2(self, x: View, out: Optional[View] = None) -> View:
        out = out or self.new(x.N, x.H // 2, x.W // 2, x.C)
        self._chk(self.L.mrfa_avgpool2_fwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, x.C, out.ptr, out.ld), "avgpool2")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_unpool2_acc(self.s, out.gptr, out.ld, x.N, out.H, out.W, x.C, x.gptr, x.ld, 0.25), "avgpool2_bwd")
            self.tape.append(bwd)
        return out

    def antialias_down(self, img_nchw: torch.Tensor, kern: torch.Tensor, stride: int, out: Optional[View] = None,
                       dimg: Optional[torch.Tensor] = None) -> View:
        """AntiAliasInterpolation2d on an NCHW image -> NHWC view.  dimg: NCHW buffer that receives (+=) the image gradient
        (the ImagePyramide of the GENERATED image); None = the input is data."""
        N, C_, H, W = img_nchw.shape
        assert kern.device == img_nchw.device, f"anti-alias kernel on {kern.device}, image on {img_nchw.device}: move the module"
        img_nchw = img_nchw.contiguous().float()
        k = kern.shape[-1]
        kern2d = kern[0, 0].contiguous().float()
        out = out or self.new(N, H // stride, W // stride, C_)
        self._chk(self.L.mrfa_antialias_down(self.s, img_nchw.data_ptr(), N, C_, H, W, kern2d.data_ptr(), k, stride, out.ptr, out.ld),
                  "antialias_down")
        if self.record and dimg is not None:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_antialias_down_bwd(self.s, out.gptr, out.ld, N, C_, H, W, kern2d.data_ptr(), k, stride, dimg.data_ptr()),
                          "antialias_down_bwd")
            self.tape.append(bwd)
        return out

    # -- MTIA prior (TokenPose_B): include/mrfa_hip.h K21 ------------------------------------------------------
    def bn_stats(self, x: View, bn):
        """batch statistics of x by a separate pass (train mode; None in eval): for raw conv outputs whose epilogue could
        not accumulate them (the stride-2 convolutions, which are sub-sampled after the conv)"""
        if not self.train:
            return None
        stats = self.f64z(self.groups * hip.STATS_SLOTS * 2 * bn.num_features)
        self._bn_stats_into(x, stats)
        return stats

    def subsample(self, x: View, stride: int = 2, out: Optional[View] = None) -> View:
        """out[n,y,x] = x[n,y*stride,x*stride]"""
        out = out or self.new(x.N, x.H // stride, x.W // stride, x.C)
        self._chk(self.L.mrfa_subsample_fwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, x.C, stride, out.ptr, out.ld), "subsample_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_subsample_bwd(self.s, out.gptr, out.ld, x.N, x.H, x.W, x.C, stride, x.gptr, x.ld), "subsample_bwd")
            self.tape.append(bwd)
        return out

    def conv_bn_raw(self, x: View, conv, bn, need_dx=True, pre=None):
        """raw = conv(pre(x)) for a conv of stride 1 or 2 that a BatchNorm follows -> (raw, batch statistics of raw); pre: see prebn() / prologue_ok()"""
        cw = convw(conv)
        fin = bn if BN_FIN_FUSED else None
        if cw.stride == 1:
            st = self.bn_stats_buf(bn)
            return self.conv(x, conv, stats=st, need_dx=need_dx, fin=fin, pre=pre), st
        assert pre is None
        if cw.stride == 2 and conv.bias is None:
            got = self._conv_strided(x, conv, cw, self.bn_stats_buf(bn), need_dx, fin=fin)
            if got is not None:
                return got
        raw = self.subsample(self.conv(x, conv, need_dx=need_dx), cw.stride)
        return raw, self.bn_stats(raw, bn)

    def _conv_strided(self, x: View, conv, cw: ConvW, stats, need_dx, fin=None):
        """stride-2 convolution as ONE launch with a strided gather and the BatchNorm statistics in its epilogue (hr_base.py:241,253,302,305,365)
        instead of the stride-1 convolution + sub-sampling + statistics passes: a quarter of the MACs forward and in the weight gradient.  The
        data gradient stays the stride-1 one over the zero-stuffed dY.  None: the library has no strided kernel for this shape."""
        Ho, Wo = (x.H + 2 * cw.pad - cw.R) // 2 + 1, (x.W + 2 * cw.pad - cw.S) // 2 + 1
        cop = (cw.Cout + 127) // 128 * 128
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = x.ptr, x.ld, x.H, x.W, 0, x.N, cw.Cin
        if cw.fwd_flat:                                # few input channels (the 3 -> 64 stem, hr_base.py:302): the fp32 tile kernel's flat-K gather, strided
            p.w_ld, p.w_tap, p.kflat, p.w_rows = (cw.T * cw.Cin + 31) // 32 * 32, 0, cw.T * cw.Cin, cop
            p.ktab = cw.ktab_fwd().data_ptr()
        else:
            p.w_ld, p.w_tap, p.kflat, p.w_rows = cw.Cin, cop * cw.Cin, 0, cop
        p.Cout, p.Hout, p.Wout = cw.Cout, Ho, Wo
        p.R, p.S, p.pad, p.stride = cw.R, cw.S, cw.pad, 2
        p.alpha, p.nbatch, p.splitk = 1.0, 1, 1
        p.y, p.ldy = x.ptr, _r4(cw.Cout)              # (placeholders of the right alignment for the query)
        p.w = x.ptr
        if not self.L.mrfa_conv2d_stride_supported(C.byref(p)):
            return None
        out = self.new(x.N, Ho, Wo, cw.Cout)
        p.w = cw.fwd_pack(False).data_ptr()
        p.y, p.ldy = out.ptr, out.ld
        late_stats = False
        if stats is not None:
            p.stats, p.groups = stats.data_ptr(), self.groups
            if self.groups > 1 and not self.L.mrfa_conv2d_groups_supported(C.byref(p)):
                p.stats, p.groups, late_stats = None, 0, True
            elif fin is not None:
                self._fin_params(p, fin, stats, out.rows)
        self._launch_conv(p, "conv2d(stride 2)", cw.Cin)
        if late_stats:
            self._bn_stats_into(out, stats)
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                if conv.weight.requires_grad:
                    dw, db = cw.grad_acc(self.pool32)
                    q = hip.WgradParams()
                    q.x, q.ldx, q.Hin, q.Win, q.ups, q.N, q.Cin = x.ptr, x.ld, x.H, x.W, 0, x.N, cw.Cin
                    q.dy, q.ldy, q.Cout, q.Hout, q.Wout = out.gptr, out.ld, cw.Cout, Ho, Wo
                    q.R, q.S, q.pad, q.stride = cw.R, cw.S, cw.pad, 2
                    q.dw, q.alpha, q.nbatch, q.ksplit = dw.data_ptr(), 1.0, 1, 0
                    if self.L.mrfa_conv2d_wgrad_stride_supported(C.byref(q)):
                        self._chk(self.L.mrfa_conv2d_wgrad_nhwc(self.s, C.byref(q)), "wgrad(stride 2)")
                        full = None
                    else:
                        full = self._zero_stuffed(out, x.H, x.W)
                        self._conv_wgrad(x, cw, full, False, None, False)
                else:
                    full = None
                if need_dx:
                    full = full or self._zero_stuffed(out, x.H, x.W)
                    self._conv_dgrad(x, cw, full, False, None)
            self.tape.append(bwd)
            if cw not in self.touched_convs:
                self.touched_convs.append(cw)
        return out, stats

    def _zero_stuffed(self, out: View, H: int, W: int) -> View:
        """(backward) a stride-1-sized view whose GRADIENT is out.grad at the even pixels and zero elsewhere: dY of the equivalent
        stride-1 convolution + sub-sampling"""
        full = self.new(out.N, H, W, out.C)
        full.st.grad = torch.zeros_like(full.st.data)
        self._chk(self.L.mrfa_subsample_bwd(self.s, out.gptr, out.ld, out.N, H, W, out.C, 2, full.st.grad.data_ptr(), full.ld), "subsample_bwd")
        return full

    def ups_add(self, lo: View, base: View, factor: int = 1, relu: bool = False, out: Optional[View] = None) -> View:
        """out = act(base + nearest_upsample(lo, factor))"""
        assert (base.N, base.H, base.W, base.C) == (lo.N, lo.H * factor, lo.W * factor, lo.C)
        out = out or self.new(base.N, base.H, base.W, base.C)
        self._chk(self.L.mrfa_upsample_add_act_fwd(self.s, lo.ptr, lo.ld, lo.N, lo.H, lo.W, lo.C, factor, base.ptr, base.ld, int(relu),
                                                   out.ptr, out.ld), "upsample_add_act_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_upsample_add_act_bwd(self.s, out.ptr, out.ld, out.gptr, out.ld, lo.N, lo.H, lo.W, lo.C, factor, int(relu),
                                                           lo.gptr, lo.ld, base.gptr, base.ld), "upsample_add_act_bwd")
            self.tape.append(bwd)
        return out

    def layernorm(self, x: View, ln: torch.nn.LayerNorm, out: Optional[View] = None) -> View:
        out = out or self.new(x.N, x.H, x.W, x.C)
        mean, rstd = self.f32(x.rows), self.f32(x.rows)
        self._chk(self.L.mrfa_layernorm_fwd(self.s, x.ptr, x.ld, x.rows, x.C, ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps),
                                            out.ptr, out.ld, mean.data_ptr(), rstd.data_ptr()), "layernorm_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                bg = bngrad(ln)
                if bg not in self.touched_bns:
                    self.touched_bns.append(bg)
                dg, db = bg.acc(self.pool32)
                scr = self.pool32.take(hip.LN_SLOTS * 2 * x.C + 1)          # slotted partials of the parameter gradients + the ticket word
                self._chk(self.L.mrfa_layernorm_bwd(self.s, x.ptr, x.ld, out.gptr, out.ld, x.rows, x.C, ln.weight.data_ptr(), mean.data_ptr(),
                                                    rstd.data_ptr(), x.gptr, x.ld, dg.data_ptr(), db.data_ptr(), scr.data_ptr()), "layernorm_bwd")
            self.tape.append(bwd)
        return out

    def gelu(self, x: View, out: Optional[View] = None) -> View:
        out = out or self.new(x.N, x.H, x.W, x.C)
        self._chk(self.L.mrfa_gelu_fwd(self.s, x.ptr, x.ld, x.rows, x.C, out.ptr, out.ld), "gelu_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_gelu_bwd(self.s, x.ptr, x.ld, out.gptr, out.ld, x.rows, x.C, x.gptr, x.ld), "gelu_bwd")
            self.tape.append(bwd)
        return out

    def attention(self, qkv: View, heads: int, scale: float) -> View:
        """qkv: (B,1,n,3*heads*d) token rows [q | k | v] -> (B,1,n,heads*d)"""
        B, n = qkv.N, qkv.H * qkv.W
        inner = qkv.C // 3
        d = inner // heads
        out = self.new(B, qkv.H, qkv.W, inner)
        lse = self.f32(B * heads * n)
        self._chk(self.L.mrfa_attention_fwd(self.s, qkv.ptr, qkv.ld, B, n, heads, d, scale, out.ptr, out.ld, lse.data_ptr()), "attention_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                delta = self.f32(B * heads * n)
                self._chk(self.L.mrfa_attention_bwd(self.s, qkv.ptr, qkv.ld, out.ptr, out.ld, out.gptr, out.ld, lse.data_ptr(), delta.data_ptr(),
                                                    B, n, heads, d, scale, qkv.gptr, qkv.ld), "attention_bwd")
            self.tape.append(bwd)
        return out

    # -- training losses: include/mrfa_hip.h K22 ---------------------------------------------------------------
    def maxpool2(self, x: View, out: Optional[View] = None) -> View:
        out = out or self.new(x.N, x.H // 2, x.W // 2, x.C)
        self._chk(self.L.mrfa_maxpool2_fwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, x.C, out.ptr, out.ld), "maxpool2_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_maxpool2_bwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, x.C, out.gptr, out.ld, x.gptr, x.ld), "maxpool2_bwd")
            self.tape.append(bwd)
        return out

    def maxpool3s2(self, x: View, out: Optional[View] = None) -> View:
        """nn.MaxPool2d(3, stride=2, padding=1)"""
        out = out or self.new(x.N, (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1, x.C)
        self._chk(self.L.mrfa_maxpool3s2_fwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, x.C, out.ptr, out.ld), "maxpool3s2_fwd")
        if self.record:
            def bwd():
                if not out.has_grad:
                    return
                self._chk(self.L.mrfa_maxpool3s2_bwd(self.s, x.ptr, x.ld, x.N, x.H, x.W, x.C, out.gptr, out.ld, x.gptr, x.ld), "maxpool3s2_bwd")
            self.tape.append(bwd)
        return out

    def l1_diff(self, x: View, y: View, acc: torch.Tensor, coef: float, gscale: Optional[torch.Tensor] = None):
        """acc (fp64 device scalar) += coef * sum|x - y|; backward dx += gscale * coef * sign(x - y) (y is a constant: the
        detached VGG features of the real image, model.py:226)"""
        assert (x.N, x.H, x.W, x.C) == (y.N, y.H, y.W, y.C)
        self._chk(self.L.mrfa_l1_diff_fwd(self.s, x.ptr, x.ld, y.ptr, y.ld, x.rows, x.C, float(coef), acc.data_ptr()), "l1_diff_fwd")
        if self.record:
            def bwd():
                self._chk(self.L.mrfa_l1_diff_bwd(self.s, x.ptr, x.ld, y.ptr, y.ld, x.rows, x.C, gscale.data_ptr() if gscale is not None else None,
                                                  coef, x.gptr, x.ld), "l1_diff_bwd")
            self.tape.append(bwd)

    # -- GEMMs for the correlation volume -----------------------------------------------------------------------
    def gemm_nt(self, a_ptr, lda, b_ptr, ldb, c_ptr, ldc, M, Nn, K, alpha, nbatch, a_bs, b_bs, c_bs, accumulate=False):
        """C[b] (=|+=) alpha * A[b] (M x K, k contiguous) @ B[b]^T (Nn x K, k contiguous)"""
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = a_ptr, lda, 1, M, 0, 1, K
        p.w, p.w_ld, p.w_tap, p.w_rows = b_ptr, ldb, 0, Nn
        p.y, p.ldy, p.Cout, p.Hout, p.Wout = c_ptr, ldc, Nn, 1, M
        p.R, p.S, p.pad = 1, 1, 0
        p.alpha, p.accumulate = alpha, int(accumulate)
        p.nbatch, p.x_bs, p.w_bs, p.y_bs = nbatch, a_bs, b_bs, c_bs
        # accumulating launches with few row tiles and a long K (dq of the pooled correlation levels: 64 ... 1 024 query rows x 4 096 keys per sample) let
        # the library split K over workgroups (atomics onto the existing gradient: no init / epilogue pass): 16-128 workgroups looping 128 k-steps each took
        # 190 us per launch for 1 GFLOP (5.6 TF/s, tools/profile_step.py)
        p.splitk = 0 if accumulate else 1
        self._launch_conv(p, "gemm_nt")

    def gemm_tn_acc(self, a_ptr, lda, b_ptr, ldb, c_ptr, M, Nn, K, alpha, nbatch, a_bs, b_bs, c_bs):
        """C[b][m][n] += alpha * sum_k A[b][k][m] * B[b][k][n]   (C dense M x Nn, atomics)"""
        q = hip.WgradParams()
        q.x, q.ldx, q.Hin, q.Win, q.ups, q.N, q.Cin = b_ptr, ldb, 1, K, 0, 1, Nn
        q.dy, q.ldy, q.Cout, q.Hout, q.Wout = a_ptr, lda, M, 1, K
        q.R, q.S, q.pad = 1, 1, 0
        q.dw, q.alpha = c_ptr, alpha
        q.nbatch, q.x_bs, q.dy_bs, q.dw_bs = nbatch, b_bs, a_bs, c_bs
        q.ksplit = 0
        self._chk(self.L.mrfa_conv2d_wgrad_nhwc(self.s, C.byref(q)), "gemm_tn")

    # -- torch glue islands (tiny tensors only: keypoint maths, soft-argmax, flow bookkeeping) ------------------
    def island(self, fn, ins) -> List[IslandOut]:
        """outs = fn(*ins) with torch device ops; ins may be Views (seen as strided (N,H,W,C) tensors), IslandOuts or
        plain tensors (module inputs / parameters).  Backward = torch.autograd.grad over the recorded sub-graph."""
        leafs = []
        for x in ins:
            if isinstance(x, View):
                t = x.tensor().detach()
            elif isinstance(x, IslandOut):
                t = x.t.detach()
            else:
                t = x.detach()
            if self.record and t.is_floating_point():
                t.requires_grad_(True)
            leafs.append(t)
        if self.record:
            with torch.enable_grad():
                graph_outs = [o.contiguous() for o in fn(*leafs)]
        else:
            with torch.no_grad():
                graph_outs = [o.contiguous() for o in fn(*leafs)]
        outs = [IslandOut(o.detach()) for o in graph_outs]
        if self.record:
            def bwd():
                sel = [(o, io.total_grad()) for o, io in zip(graph_outs, outs)]
                sel = [(o, g) for o, g in sel if g is not None and o.requires_grad]
                if not sel:
                    return
                req = [(x, l) for x, l in zip(ins, leafs) if l.requires_grad]
                grads = torch.autograd.grad([o for o, _ in sel], [l for _, l in req], [g for _, g in sel], allow_unused=True)
                for (x, _), g in zip(req, grads):
                    if g is None:
                        continue
                    if isinstance(x, View):
                        x.grad_tensor().add_(g)
                    elif isinstance(x, IslandOut):
                        x.add_grad(g)
                    else:
                        k = id(x)
                        self.ext_grads[k] = g if k not in self.ext_grads else self.ext_grads[k] + g
            self.tape.append(bwd)
        return outs

    # -- backward driver --------------------------------------------------------------------------------------
    def run_backward(self):
        # one zero arena for the gradients of every forward activation (single memset instead of ~500 fills)
        self.in_backward = True
        planned = self.zero_plan or ()
        for st in self.storages:
            # (buffers the plan knows to need a zero fill -- they took the lazy one the last time this program ran -- join the arena below: ONE fill at HBM
            # speed for them and the small buffers together; as a multi-tensor zero of their own they were 7 launches / 0.26 ms at the head of RaftFlow's backward)
            if st.grad is None and (st.grad_noinit or st.data.numel() >= FRESH_MIN_ELEMS) and st.seq not in planned:
                st.grad = torch.empty_like(st.data)
                if FRESH_NAN:
                    st.grad.fill_(float("nan"))
                st.fresh = True
        todo = [st for st in self.storages if st.grad is None]
        total = sum(st.data.numel() for st in todo)
        if total:
            arena = torch.zeros(total, dtype=torch.float32, device=self.dev)
            off = 0
            for st in todo:
                n = st.data.numel()
                st.grad = arena[off:off + n].view(st.rows, st.ld)
                off += n
        self.storages = []
        dbg = Ctx.debug_backward
        if dbg is None:
            for fn in reversed(self.tape):
                fn()
        else:
            # determinism debugging (Ctx.debug_backward = a list): fingerprint of every activation gradient after each closure
            rec = []
            for i, fn in enumerate(reversed(self.tape)):
                fn()
                fv = dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in (fn.__closure__ or ()))))
                desc = fn.__qualname__.replace("Ctx.", "").replace(".<locals>", "")
                o, cw = fv.get("out"), fv.get("cw")
                if isinstance(o, View):
                    desc += f" out=({o.N},{o.H},{o.W},{o.C})"
                if cw is not None and hasattr(cw, "Cin"):
                    desc += f" conv {cw.Cin}->{cw.Cout} {cw.R}x{cw.S}"
                fp = float(arena.abs().sum(dtype=torch.float64)) if total else 0.0
                rec.append((i, desc, fp))
            dbg.append(rec)
        if getattr(self, "side_used", False):         # join the weight-gradient side chain before anyone reads dW
            torch.cuda.current_stream(self.dev).wait_stream(Ctx._side)
            self.side_used = False
        self.tape = []


# ------------------------------------------------------------------------------------------------- autograd bridge
class _ProgramFn(torch.autograd.Function):
    """Runs `program(ctx, *inputs) -> (outputs, seeders, input_grad_fns)` as one autograd node.

    outputs: tuple of torch tensors returned to the caller; seeders[i](grad) pushes d(outputs[i]) into the engine;
    input_grad_fns[j]() returns the gradient of inputs[j] (or None) after the tape ran."""

    @staticmethod
    def forward(actx, program, module, n_in, want_grad, *args):
        inputs, params = args[:n_in], args[n_in:]
        dev = inputs[0].device
        # want_grad: grad mode of the CALLER (inside Function.forward it is always off, and needs_input_grad reports the
        # parameters' requires_grad even under torch.no_grad()): no tape, no retained activations for inference
        need = want_grad and any(actx.needs_input_grad[4:])
        ectx = Ctx(dev, train=module.training, record=need)
        if need and dev.type == "cuda":
            plans = module.__dict__.setdefault("_mrfa_zero_plans", {})
            ectx.zero_plan = plans.setdefault((getattr(program, "__name__", repr(program)), module.training, STAT_GROUPS, tuple(tuple(t.shape) for t in inputs)), set())
        outs, seeders, in_grad_fns = program(ectx, *inputs)
        ectx.flush_forward()
        actx.ectx, actx.seeders, actx.in_grad_fns = ectx, seeders, in_grad_fns
        actx.params, actx.n_in = params, n_in
        actx.mname = type(module).__name__
        # return ALIASES: autograd stamps grad_fn (= this node) on the returned tensor objects, and the seeders held
        # by this node reference the program's own output tensors -- handing those out would close a reference
        # cycle through C++ (node -> seeders -> tensor -> grad_fn -> node) that no garbage collector can break
        return tuple(o.detach() for o in outs)

    @staticmethod
    def backward(actx, *gouts):
        ectx = actx.ectx
        if _PENDING_DEFERRED and any(d is not ectx.wdefer for d in _PENDING_DEFERRED):
            # the backward pass has left the programs deferring into another collection: their weight gradients start now, beside this
            # program's backward (this program's own collection, if it has one, waits for HotPath.join())
            for d in [d for d in _PENDING_DEFERRED if d is not ectx.wdefer]:
                d.flush(ectx.dev)
                _PENDING_DEFERRED.remove(d)
        mark("backward " + actx.mname + ": start")
        for seed, g in zip(actx.seeders, gouts):
            if g is not None and seed is not None:
                seed(g)
        ectx.run_backward()
        mark("backward " + actx.mname + ": tape done")
        pgrads = {}
        direct_cws = [cw for cw in ectx.touched_convs if cw.dw_acc is not None and cw._direct]
        skip = ()
        if ectx.wdefer is not None and direct_cws:
            # after the deferred launches, on their stream.  The accumulators leave their ConvW objects NOW: the next program that touches these
            # convolutions (the other encoder pass, deferring or not) gets accumulators of its own -- shared ones would be un-packed (and dropped)
            # by whichever program finishes first while deferred launches still add into them
            accs = [cw.dw_acc for cw in direct_cws]
            for cw in direct_cws:
                cw.dw_acc = cw.db_acc = None
            ectx.wdefer.add(lambda: unpack_direct(direct_cws, accs), final=True)
            if ectx.wdefer not in _PENDING_DEFERRED and not ectx.wdefer.manual:
                _PENDING_DEFERRED.append(ectx.wdefer)
            skip = {id(cw) for cw in direct_cws}
        else:
            unpack_direct(direct_cws)
        for cw in ectx.touched_convs:
            if id(cw) in skip:
                continue
            dw, db = cw.take_grads()
            if dw is not None:
                pgrads[id(cw.conv.weight)] = dw
                if cw.conv.bias is not None and db is not None:
                    pgrads[id(cw.conv.bias)] = db
        for bg in ectx.touched_bns:
            dg, db = bg.take()
            if dg is not None:
                pgrads[id(bg.bn.weight)] = dg
                pgrads[id(bg.bn.bias)] = db
        for p in actx.params:                      # parameters touched by torch glue islands
            g = ectx.ext_grads.get(id(p))
            if g is not None:
                if _direct_ok(p):
                    p.grad.add_(g)
                else:
                    pgrads[id(p)] = g
        in_grads = [fn() if (fn is not None and need) else None
                    for fn, need in zip(actx.in_grad_fns, actx.needs_input_grad[4:4 + actx.n_in])]
        out = [None, None, None, None] + in_grads + [pgrads.get(id(p)) for p in actx.params]
        actx.ectx = actx.seeders = actx.in_grad_fns = actx.params = None
        return tuple(out)


def run_program(module: torch.nn.Module, program, inputs: Sequence[torch.Tensor]):
    params = [p for p in module.parameters()]
    return _ProgramFn.apply(program, module, len(inputs), torch.is_grad_enabled(), *inputs, *params)
