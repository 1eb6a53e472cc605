"""Minimal data-parallel training harness for the hot path (the reference's train.py:21-25,39-48,58-70 restated):
Adam(lr 2e-4, betas (0.5, 0.999)) over encoder/decoder/dense_motion, inf-norm gradient clipping on encoder and
dense_motion, DistributedDataParallel over RCCL for gradient sync only (no model sharding).

The reference's real losses (VGG19 perceptual pyramid, equivariance) need torchvision weights and are out of scope
(SURVEY.md 8f-2); the step here uses the surrogate L1 loss mean|gen - driving| that SURVEY.md 8(d) defines for the
fwd+bwd metric."""
from __future__ import annotations

import contextlib
import math
import os

import torch
from torch import nn

from .modules import DenseMotionNetwork, KPDetector, RaftFlow
from .modules.util import AntiAliasInterpolation2d

VOX1 = dict(
    fomm_kp_detector=dict(block_expansion=32, num_kp=10, num_channels=3, max_features=1024, num_blocks=5, temperature=0.1,
                          scale_factor=0.25, estimate_jacobian=True, estimate_occlusion=False),
    dense_motion=dict(block_expansion=64, max_features=1024, num_blocks=5, scale_factor=0.25, num_kp=10, num_channels=3,
                      estimate_occlusion_map=True),
    raft_flow=dict(prior_only=False, num_kp=10, dim=256, size=256,
                   generator=dict(num_channels=3, block_expansion=64, max_features=512, num_up_blocks=5),
                   driving_encoder=dict(in_features=10, block_expansion=32, max_features=512, num_blocks=5),
                   source_encoder=dict(in_features=13, block_expansion=32, max_features=512, num_blocks=5)),
    # config/vox1.yaml:117-186 (the same block in celebvhq.yaml): the MTIA prior, TokenPose_B
    mtia_kp_detector=dict(MODEL=dict(
        ESTIMATE_JACOBIAN=True, DATA_PREPROCESS=False, FIX_IMG2MOTION_ATTENTION=False, TRANSFORMER_DEPTH=12, TRANSFORMER_HEADS=8, DIM=192,
        INIT_WEIGHTS=False, NAME="pose_tokenpose_b", NUM_JOINTS=10, PRETRAINED="", PATCH_SIZE=[4, 4], IMAGE_SIZE=[256, 256],
        HEATMAP_SIZE=[64, 64], TAG_PER_JOINT=True, HIDDEN_HEATMAP_DIM=-1, MULTI_TRANSFORMER_DEPTH=[12, 12],
        MULTI_TRANSFORMER_HEADS=[16, 16], MULTI_DIM=[48, 48], NUM_BRANCHES=1, BASE_CHANNEL=32, TRANSFORMER_MLP_RATIO=3,
        POS_EMBEDDING_TYPE="sine-full", TEMPERATURE=0.1, TARGET_TYPE="gaussian", INIT=True, SIGMA=2,
        EXTRA=dict(PRETRAINED_LAYERS=["conv1", "bn1", "conv2", "bn2", "layer1", "transition1", "stage2", "transition2", "stage3"],
                   FINAL_CONV_KERNEL=1,
                   STAGE2=dict(NUM_MODULES=1, NUM_BRANCHES=2, BLOCK="BASIC", NUM_BLOCKS=[4, 4], NUM_CHANNELS=[32, 64], FUSE_METHOD="SUM"),
                   STAGE3=dict(NUM_MODULES=4, NUM_BRANCHES=3, BLOCK="BASIC", NUM_BLOCKS=[4, 4, 4], NUM_CHANNELS=[32, 64, 128],
                               FUSE_METHOD="SUM")))),
    train_params=dict(lr=2.0e-4, clip=10.0, prior_model="mtia"),
)


class HotPath(nn.Module):
    """encoder -> dense_motion -> decoder wiring of MRFA.forward (modules/model.py:185-210), attribute names as in the
    reference so checkpoints and train.py's parameter groups line up."""

    def __init__(self, cfg=VOX1, prior: str = "fomm", background: bool = False):
        """prior: 'fomm' = KPDetector (hourglass + soft-argmax), 'mtia' = TokenPose_B (HRNet + token transformer; the
        `prior_model` both reference YAMLs select, model.py:170-172); background: BGMotionPredictor feeding bg_param to the dense
        motion network (celebvhq.yaml `bg_start: 0`, model.py:174-176,189-192)"""
        super().__init__()
        self.prior = prior
        self.bg_predictor = None
        if background:
            from .modules.bg_motion_predictor import BGMotionPredictor
            self.bg_predictor = BGMotionPredictor()
        if prior == "fomm":
            self.encoder = KPDetector(**cfg["fomm_kp_detector"])
        elif prior == "mtia":
            import copy
            from .modules.transformer import get_pose_net
            from .modules.util import convert_dict_to_attrit_dict
            self.encoder = get_pose_net(convert_dict_to_attrit_dict(copy.deepcopy(cfg["mtia_kp_detector"])), is_train=True)
        else:
            raise NotImplementedError(f"prior_model={prior!r}: 'fomm' and 'mtia' are built (TPSM is out of scope, SURVEY.md section 8)")
        self.dense_motion = DenseMotionNetwork(**cfg["dense_motion"])
        self.decoder = RaftFlow(**cfg["raft_flow"])
        self.down = AntiAliasInterpolation2d(3, 0.25)
        # training: the encoder calls of a step -- encoder(source), encoder(driving) (and encoder(transformed driving) under the reference objective;
        # model.py:185-186,234) -- as ONE program over a batch of len(frames) x B samples whose BatchNorm layers keep their batch statistics per call
        # ("statistic groups", include/mrfa_hip.h v7 / engine.stat_groups): the results of the separate calls -- per-call statistics, running buffers
        # updated once per call in call order, num_batches_tracked += number of calls -- from half / a third of the launches (2 483 instead of 3 573 per
        # step), on one stream.  Rounds 1-4 ran the calls as separate programs on concurrent streams; measured against that on one box (round 5, B = 8):
        # surrogate step 84.7-85.6 vs 83.8-84.2 ms, reference objective (three calls) 120.6 vs 124.8 ms, SyncBatchNorm (one stream by necessity) 95.1 vs
        # 104.0 ms.  Default for the MTIA prior (TokenPose_B: ~1 000 launches per call and direction); KPDetector's ~150-launch calls stay separate.
        self.batched_encoder = prior == "mtia"
        # training with direct parameter gradients: the weight-gradient kernels of dense motion + decoder (~100 launches, 22 ms, each
        # filling the chip) are collected during their backward and issued on a side stream when the backward reaches the keypoint
        # encoder, whose small kernels leave most of the GPU idle (engine.DeferredWgrads).  join() orders them before the optimizer.
        # mrfa_amd.graph.GraphedTrainStep switches it on for the MTIA prior.
        self.defer_decoder_wgrads = False
        from .engine import DeferredWgrads
        object.__setattr__(self, "_wdefer", DeferredWgrads())
        object.__setattr__(self, "_wdefer_dev", None)
        # the keypoint encoder's own weight gradients (~200 small launches inside its backward chain): collected and dealt onto
        # MRFA_ENC_WGRAD_FANOUT (default 4; 0 / 1 = in line) side streams after the chain: the backward chain loses a fifth of
        # its launches, the independent launches then run four abreast as multi-problem launches (round 3, 5 alternating runs of 20 steps on one box:
        # 85.9 -> 84.2 ms).  Same conditions as defer_decoder_wgrads.
        self.defer_encoder_wgrads = True       # (GraphedTrainStep switches it off with SyncBatchNorm: one stream, one enqueue order per rank)
        object.__setattr__(self, "_wdefer_enc", DeferredWgrads(fanout=int(os.environ.get("MRFA_ENC_WGRAD_FANOUT", "4") or 0), manual=True,
                                                                   batch=True))
        # test / diagnostics hook: a dict here makes forward() keep the keypoint tensors (`kp_s`, `jac_s`, `kp_d`, `jac_d`) and copy the
        # gradients that arrive at them into static buffers (`dkp_s`, ...) with a kernel (no memcpy node), so that d loss / d keypoints of
        # a hipGraph-REPLAYED step can be read (tests/test_headline_gpu.py).  None (default): nothing is recorded
        object.__setattr__(self, "probe", None)
        # GraphedTrainStep: the stream on which the packed weight layouts of everything but the keypoint encoder are being refreshed
        object.__setattr__(self, "_pack_stream", None)

    def encode_many(self, frames):
        """[encoder(f) for f in frames] (reference model.py:185-186 and the third pass of :234): in training with `batched_encoder`, one program over
        the concatenated batch with one BatchNorm statistic group per frame (engine.stat_groups)"""
        from . import engine
        first = frames[0]
        enc_defer = None
        if (self.defer_decoder_wgrads and self.defer_encoder_wgrads and self._wdefer_enc.fanout > 1 and self.training and torch.is_grad_enabled()
                and first.is_cuda and len(frames) >= 2):
            enc_defer = self._wdefer_enc
            enc_defer.reset()
        with engine.defer_wgrads(enc_defer):
            if not (self.batched_encoder and self.training and len(frames) > 1 and all(f.shape == first.shape for f in frames)):
                return [self.encoder(f) for f in frames]
            with engine.stat_groups(len(frames)):
                kp = self.encoder(torch.cat(list(frames), dim=0))
            b = first.shape[0]
            parts = {k: v.view(len(frames), b, *v.shape[1:]).unbind(0) for k, v in kp.items()}
            return [{k: parts[k][i] for k in kp} for i in range(len(frames))]

    def encode_pair(self, source, driving):
        """(kp_source, kp_driving) = (encoder(source), encoder(driving)), reference model.py:185-186"""
        if not self.training:
            return encode_pair_eval(self.encoder, source, driving)
        kp_s, kp_d = self.encode_many([source, driving])
        return kp_s, kp_d

    def await_packs(self):
        """before the first launch that reads a packed weight layout outside the keypoint encoder (GraphedTrainStep refreshes those on a side
        stream beside the encoder's forward)"""
        st = self._pack_stream
        if st is not None:
            torch.cuda.current_stream(st.device).wait_stream(st)
            object.__setattr__(self, "_pack_stream", None)

    def join(self):
        """after backward(): the side streams' kernels (the deferred weight gradients) are ordered before whatever the caller issues next"""
        from . import engine
        engine.mark("backward: main stream done")
        if self._wdefer_dev is not None:
            self._wdefer.join(self._wdefer_dev)
            engine.mark("joined dec wgrads")
            self._wdefer_enc.join(self._wdefer_dev)
            engine.mark("joined enc wgrads")

    def decode(self, source, kp_s, kp_d, bg_param=None):
        """dense motion + refinement + generator for given keypoints (model.py:188-210)"""
        from . import engine
        self.await_packs()
        defer = self._wdefer if (self.defer_decoder_wgrads and self.training and torch.is_grad_enabled()) else None
        if defer is not None:
            defer.reset()                  # nothing of an earlier (possibly aborted) step may reach this step's gradients
            object.__setattr__(self, "_wdefer_dev", source.device)
        with engine.defer_wgrads(defer):
            img_down = self.down(source)
            dm = self.dense_motion(source, kp_d, kp_s, bg_param=bg_param)
            gen, warp_img, occ = self.decoder(kp_s["kp"], kp_d["kp"], dm, img=img_down, img_full=source)
        return gen

    def _record_probe(self, kp_s, kp_d):
        pr = self.probe
        for tag, kp in (("s", kp_s), ("d", kp_d)):
            for key, short in (("kp", "kp"), ("jacobian", "jac")):
                t = kp.get(key)
                if t is None:
                    continue
                pr[f"{short}_{tag}"] = t.detach()
                if t.requires_grad:
                    def hook(g, name=f"d{short}_{tag}"):
                        buf = pr.get(name)
                        if buf is None or buf.shape != g.shape:
                            pr[name] = buf = torch.empty_like(g)
                        torch.mul(g, 1.0, out=buf)
                    t.register_hook(hook)

    def forward(self, source, driving):
        from . import engine
        engine.mark("forward: start")
        kp_s, kp_d = self.encode_pair(source, driving)
        engine.mark("forward: keypoints")
        self.await_packs()
        if self.probe is not None:
            self._record_probe(kp_s, kp_d)
        bg_param = self.bg_predictor(source, driving) if self.bg_predictor is not None else None
        gen = self.decode(source, kp_s, kp_d, bg_param)
        engine.mark("forward: generated")
        return gen


def encode_pair_eval(encoder, source, driving):
    """eval mode: BatchNorm uses its running statistics, every sample is independent, so the source and the driving frames go
    through the keypoint encoder as ONE batch of 2B (half the launches, twice the rows per launch)"""
    b = source.shape[0]
    kp = encoder(torch.cat([source, driving], dim=0))
    return {k: v[:b] for k, v in kp.items()}, {k: v[b:] for k, v in kp.items()}


def l1_loss(gen: torch.Tensor, driving: torch.Tensor) -> torch.Tensor:
    """mean |gen - driving| (the surrogate loss of SURVEY.md 8(d)) as two single-workgroup-per-output reductions.
    torch's .mean() over ~1.5 M elements is a multi-workgroup reduction that zeroes its semaphore with hipMemsetAsync;
    captured into a hipGraph that becomes a memset node, which ROCm 7.2 does not order reliably against kernel nodes
    on replay (the loss then reads 0.0).  Same value up to fp32 summation order."""
    d = (gen - driving).abs()
    n = d.numel()
    for w in (1024, 512, 256, 128):
        if n % w == 0 and n // w <= 4096:
            return d.view(-1, w).sum(dim=1).sum() / n
    return d.mean()


def reference_loss(model: HotPath, full_loss, source, driving) -> torch.Tensor:
    """The reference's generator objective (train.py:60-63: sum of the .mean() of every entry of MRFA.forward's loss_values,
    model.py:219-246): VGG19 perceptual pyramid + equivariance + equivariance-Jacobian; `full_loss` = mrfa_amd.losses.GeneratorFullLoss."""
    transform = transformed_kp = None
    if full_loss.loss_weights['equivariance'] != 0:
        # the third encoder pass (model.py:232-234) depends on the driving frame only: issued together with the first two
        from .losses import Transform
        transform = Transform(driving.shape[0], device=driving.device, **full_loss.train_params['transform_params'])
        kp_s, kp_d, transformed_kp = model.encode_many([source, driving, transform.transform_frame(driving)])
    else:
        kp_s, kp_d = model.encode_pair(source, driving)
    model.await_packs()
    bg = model.bg_predictor(source, driving) if model.bg_predictor is not None else None
    gen = model.decode(source, kp_s, kp_d, bg)
    bg_rev = model.bg_predictor(driving, source) if bg is not None else None
    values = full_loss(model.encoder, driving, gen, kp_d, transform=transform, transformed_kp=transformed_kp, bg_param=bg,
                       bg_param_reverse=bg_rev)
    return sum(v.mean() for v in values.values())


def make_optimizer(model: HotPath, lr=2.0e-4, capturable=False, fused=False, clip=10.0):
    """The reference's optimizer (train.py:21): Adam(lr, betas=(0.5, 0.999)) over three parameter groups.

    fused=False  torch.optim.Adam (capturable=True keeps its step counters on the device for hipGraph capture);
    fused=True   mrfa_amd.optim.FlatAdam: parameters, gradients and moments re-homed into flat HBM buffers, inf-norm
                 clipping of the encoder / dense_motion groups (train.py:65-67) + Adam as 6 HIP launches."""
    m = model.module if hasattr(model, "module") else model
    bg = getattr(m, "bg_predictor", None)           # train.py:23-25,68-72: its own Adam in the reference (same lr / betas), clipped too
    if fused:
        from .optim import FlatAdam
        groups = [{"params": list(m.encoder.parameters()), "clip": clip}, {"params": list(m.decoder.parameters())},
                  {"params": list(m.dense_motion.parameters()), "clip": clip}]
        if bg is not None:
            groups.append({"params": list(bg.parameters()), "clip": clip})
        return FlatAdam(groups, lr=lr, betas=(0.5, 0.999))
    groups = [{"params": m.encoder.parameters()}, {"params": m.decoder.parameters()}, {"params": m.dense_motion.parameters()}]
    if bg is not None:
        groups.append({"params": bg.parameters()})
    return torch.optim.Adam(groups, lr=lr, betas=(0.5, 0.999), capturable=capturable)


def train_step(model, optimizer, source, driving, clip=10.0, loss_fn=None):
    """one fwd + bwd + clip + Adam step; returns the (device) loss tensor, detached: a loss that still references its
    autograd graph would keep the parameters' AccumulateGrad nodes -- and the stream they were created on -- alive,
    which breaks a later hipGraph capture on another stream"""
    from . import engine
    optimizer.zero_grad(set_to_none=True)
    wrapped = hasattr(model, "module")                 # DistributedDataParallel needs autograd's gradient hooks
    fused = getattr(optimizer, "fused_clip", False)
    with (engine.direct_param_grads() if (fused and not wrapped) else contextlib.nullcontext()):
        if loss_fn is None:
            gen = model(source, driving)
            loss = l1_loss(gen, driving)
        else:
            loss = loss_fn(model, source, driving)
        loss.backward()
    m = model.module if wrapped else model
    if hasattr(m, "join"):
        m.join()
    if clip and not fused:
        nn.utils.clip_grad_norm_(m.encoder.parameters(), max_norm=clip, norm_type=math.inf)
        nn.utils.clip_grad_norm_(m.dense_motion.parameters(), max_norm=clip, norm_type=math.inf)
        if getattr(m, "bg_predictor", None) is not None:
            nn.utils.clip_grad_norm_(m.bg_predictor.parameters(), max_norm=clip, norm_type=math.inf)
    optimizer.step()
    return loss.detach()


class SplitBackward:
    """The step's forward + backward cut at the keypoint encoder's outputs, so that the data-parallel gradient exchange can start
    while the encoder's backward still runs (the reference gets the same overlap from DistributedDataParallel's buckets,
    train.py:45-48): `head()` = encoder forward, dense motion + decoder forward, loss, backward down to d(loss)/d(keypoints) --
    after it the decoder / dense-motion / background gradients (408 of the 457 MB of the MTIA configuration) are final;
    `tail()` = the encoder's backward (both passes; ~30 ms of the 113 ms step) from those keypoint gradients.  Same arithmetic as
    one backward() call: autograd would run exactly these nodes in this order."""

    def __init__(self, model, loss_fn=None):
        assert self.supported(model, loss_fn)
        self.model = model
        self.cut = self.det = None

    @staticmethod
    def supported(model, loss_fn=None) -> bool:
        """the surrogate-loss step of a HotPath; the reference objective (third encoder pass, equivariance terms on the keypoints
        themselves) keeps the single backward and the un-overlapped exchange"""
        return loss_fn is None and all(hasattr(model, a) for a in ("encode_pair", "decode", "encoder"))

    def head(self, source, driving):
        m = self.model
        kp_s, kp_d = m.encode_pair(source, driving)
        self.cut, self.det = [], []

        def leaves(kp):
            out = {}
            for k, v in kp.items():
                if torch.is_tensor(v) and v.requires_grad:
                    w = v.detach().requires_grad_(True)
                    self.cut.append(v)
                    self.det.append(w)
                    out[k] = w
                else:
                    out[k] = v
            return out
        kp_s, kp_d = leaves(kp_s), leaves(kp_d)
        bg = m.bg_predictor(source, driving) if getattr(m, "bg_predictor", None) is not None else None
        gen = m.decode(source, kp_s, kp_d, bg)
        loss = l1_loss(gen, driving)
        loss.backward()
        return loss, gen

    def tail(self):
        pairs = [(c, d.grad) for c, d in zip(self.cut, self.det) if d.grad is not None]
        self.cut = self.det = None
        if pairs:
            torch.autograd.backward([c for c, _ in pairs], [g for _, g in pairs])
        if hasattr(self.model, "join"):
            self.model.join()


def exchange_ranges(grads, model):
    """(head, tail) element ranges of the flat gradient buffer `grads` (mrfa_amd.graph.FlatGradients): tail = the keypoint encoder's
    parameters, head = everything else (final after SplitBackward.head())"""
    enc = {id(p) for p in model.encoder.parameters()}
    return (grads.ranges_of([p for p in grads.params if id(p) not in enc]), grads.ranges_of([p for p in grads.params if id(p) in enc]))


def train_step_overlapped(model, optimizer, source, driving, world: int = 1):
    """train_step's data-parallel form WITHOUT DistributedDataParallel, as mrfa_amd.graph.GraphedTrainStep schedules it (this is the
    same schedule launched eagerly; it runs on CPU / gloo through the ABI emulator in tests/):
        head -> async all-reduce of the non-encoder gradient ranges -> tail (encoder backward) -> all-reduce of the encoder range
        -> wait -> 1/world + clip + Adam (FlatAdam).
    `model` is the bare HotPath, `optimizer` a FlatAdam (make_optimizer(fused=True))."""
    from . import engine
    assert getattr(optimizer, "fused_clip", False) and not hasattr(model, "module")
    grads = optimizer.grads
    if not grads.bound():
        grads.bind()
    grads.flat.zero_()
    head_r, tail_r = exchange_ranges(grads, model)
    sb = SplitBackward(model)
    with engine.direct_param_grads():
        loss, _ = sb.head(source, driving)
        handles = grads.all_reduce(head_r, async_op=True) if world > 1 else []
        sb.tail()
        handles += grads.all_reduce(tail_r, async_op=True) if world > 1 else []
    for h in handles:
        h.wait()
    optimizer.grad_scale = 1.0 / world
    optimizer.step()
    return loss.detach()


def sync_bn_buffers(model) -> int:
    """Data-parallel runs WITHOUT SyncBatchNorm (the hipGraph step: per-rank batch statistics, `broadcast_buffers=False`) let every
    rank's BatchNorm running_mean / running_var drift apart: each is an exponential average over that rank's own batches.  Before a
    checkpoint is written they are averaged over the ranks -- ONE all-reduce of all floating-point buffers flattened together -- so that
    the file does not depend on which rank wrote it (for equal per-rank batch sizes the mean of the per-rank running means IS the
    running mean over the global batches; the running variances average the within-rank variances, which is what
    DistributedDataParallel(broadcast_buffers=True) without SyncBatchNorm would keep from rank 0 only).  A COLLECTIVE: every rank must
    call it (it is never issued implicitly -- save_checkpoint is rank-local unless asked).  No-op without a process group / with one
    rank; with SyncBatchNorm (reference train.py:43) the buffers are already identical and callers skip it.  Returns the element count."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    m = model.module if hasattr(model, "module") else model
    bufs = [b for n, b in m.named_buffers() if b.is_floating_point() and n.endswith(("running_mean", "running_var"))]
    if not bufs:
        return 0
    flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
    dist.all_reduce(flat)
    flat.mul_(1.0 / dist.get_world_size())
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    return off


def save_checkpoint(path: str, model, optimizer, epoch: int, collective: bool = False):
    """The reference's checkpoint file (logger.py:50-58, train.py:94): {'model': state_dict with DDP's 'module.' prefix,
    'optimizer': optimizer.state_dict(), 'epoch': int} -- readable by the reference's Logger.load_cpk and by load_checkpoint.
    `model` = mrfa_amd.modules.MRFA gives the reference MRFA's exact key set (`pyramid.*`, `vgg.*`, `encoder.*`, ...: pinned by
    tests/golden/state_dict_manifest.json['MRFA']); a HotPath writes the networks only (no loss modules: the reference's strict
    loader then reports the missing `pyramid.*` / `vgg.*` keys, as it would for any file without them).

    RANK-LOCAL by default, like the reference (train.py:89-94 saves under `if local_rank == 0`): the caller decides which rank writes, no
    collective is issued here, the live BatchNorm buffers are not touched.  Data-parallel runs with per-rank batch statistics that want
    a rank-independent file either call `sync_bn_buffers(model)` on EVERY rank first, or pass collective=True on every rank: the
    buffers are then averaged over the ranks (one all-reduce; skipped when the model holds SyncBatchNorm layers, whose buffers are
    already identical) and only rank 0 writes."""
    m = model.module if hasattr(model, "module") else model
    if collective:
        import torch.distributed as dist
        if not any(isinstance(x, nn.SyncBatchNorm) for x in m.modules()):
            sync_bn_buffers(m)
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
    sd = {"module." + k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    torch.save({"model": sd, "optimizer": optimizer.state_dict(), "epoch": int(epoch)}, path)


# keys a checkpoint may hold that a model without the loss modules has no slot for (reference model.py:154-157), and vice versa
_LOSS_PREFIXES = ("pyramid.", "vgg.")


def load_checkpoint(path: str, model, optimizer=None, strict: bool = True) -> int:
    """Logger.load_cpk (logger.py:60-66) for either side's files: accepts keys with or without the 'module.' prefix and an
    optimizer state written by torch.optim.Adam or by FlatAdam; returns the epoch to resume from.

    strict=True (the reference's behaviour): every key of the model must be in the file and every key of the file must have a slot in
    the model, otherwise RuntimeError naming the keys -- with ONE documented exception in each direction: the loss modules'
    `pyramid.*` / `vgg.*` entries are ignored when the model has no such modules (a HotPath loading a full MRFA checkpoint), and are
    reported in the warning log, never silently, when the file lacks them (a HotPath checkpoint loaded into an MRFA: the VGG19 then
    keeps whatever weights it had).  strict=False: train.py:27-32's fine-tuning load (missing / unexpected keys are returned by
    torch and logged)."""
    import logging
    log = logging.getLogger("mrfa_amd")
    cpk = torch.load(path, map_location="cpu")
    m = model.module if hasattr(model, "module") else model
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in cpk["model"].items()}
    own = m.state_dict()
    has_loss_modules = any(k.startswith(_LOSS_PREFIXES) for k in own)
    if not has_loss_modules:
        dropped = sorted(k for k in sd if k.startswith(_LOSS_PREFIXES))
        if dropped:
            log.warning("load_checkpoint: %d loss-module entries of %s have no slot in %s and are not loaded (%s ...)", len(dropped), path,
                        type(m).__name__, dropped[0])
        sd = {k: v for k, v in sd.items() if not k.startswith(_LOSS_PREFIXES)}
    missing = sorted(k for k in own if k not in sd)
    unexpected = sorted(k for k in sd if k not in own)
    loss_missing = [k for k in missing if k.startswith(_LOSS_PREFIXES)]
    if loss_missing and len(loss_missing) == len(missing) and not unexpected:
        log.warning("load_checkpoint: %s holds no `pyramid.*` / `vgg.*` entries (%d keys): the perceptual-loss modules keep their current "
                    "weights", path, len(loss_missing))
        m.load_state_dict(sd, strict=False)
    elif strict and (missing or unexpected):
        raise RuntimeError(f"load_checkpoint({path}): missing keys {missing[:8]}{' ...' if len(missing) > 8 else ''} ({len(missing)}), "
                           f"unexpected keys {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''} ({len(unexpected)})")
    else:
        res = m.load_state_dict(sd, strict=False)
        if res.missing_keys or res.unexpected_keys:
            log.warning("load_checkpoint(strict=False): %d missing, %d unexpected keys", len(res.missing_keys), len(res.unexpected_keys))
    if optimizer is not None and "optimizer" in cpk:
        optimizer.load_state_dict(cpk["optimizer"])
    return int(cpk.get("epoch", 0))
