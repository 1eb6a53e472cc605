"""FlatAdam: the optimizer step of the training path (reference train.py:21 torch.optim.Adam(betas=(0.5, 0.999)),
train.py:65-67 clip_grad_norm_(.., norm_type=inf), train.py:70 optimizer.step()) on flat HBM buffers.

Every parameter, gradient and Adam moment of the model is a 16-byte aligned slice of ONE fp32 buffer each (parameter
groups are contiguous segments), so that
  * zero_grad is one memset and the data-parallel gradient exchange one all-reduce (mrfa_amd.graph.FlatGradients),
  * clip + Adam for 311 tensors / 116 M parameters is 6 launches of include/mrfa_hip.h K20 (mrfa_adam_prepare,
    mrfa_grad_absmax, mrfa_adam_flat) instead of ~1 500 ATen launches, and every scalar of the update (step count,
    bias corrections, learning rate, clip coefficient) lives in device memory: the step is hipGraph-capturable.
The Optimizer surface is torch's: param_groups (an LR scheduler edits group['lr']), state[p] = {step, exp_avg,
exp_avg_sq} (views of the flat buffers), state_dict() / load_state_dict() interchangeable with torch.optim.Adam.
"""
from __future__ import annotations

from typing import List

import torch

from . import hip
from .graph import FlatGradients, _increment_version

_STATE = 8          # MRFA_ADAM_STATE_FLOATS


class FlatAdam(torch.optim.Optimizer):
    fused_clip = True           # train_step: the inf-norm clipping of groups with a 'clip' entry happens inside step()

    def __init__(self, params, lr=2.0e-4, betas=(0.5, 0.999), eps=1e-8):
        """params: parameter groups as for torch.optim.Adam; a group may carry 'clip': max inf-norm of its gradients."""
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, clip=None, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=True, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        for g in self.param_groups:
            assert g["weight_decay"] == 0 and not g["amsgrad"] and not g["maximize"], "FlatAdam: plain Adam only (the reference's setting)"
            assert tuple(g["betas"]) == tuple(betas) and g["eps"] == eps, "FlatAdam: betas / eps are shared by all groups"
        self.grad_scale = 1.0                    # 1 / world_size of the data-parallel mean, folded into the update
        self._flatten()

    # -- layout
    def _flatten(self):
        # frozen parameters (TokenPose_B's sine position code is an nn.Parameter(requires_grad=False)) stay in param_groups, as
        # they do in the reference's torch.optim.Adam, so that state_dict() indices are interchangeable; they get no slot
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        assert all(p.dtype == torch.float32 for p in ps)
        dev = ps[0].device
        self.grads = FlatGradients(ps)
        self.flat_g = self.grads.flat
        n = self.grads.total
        self.flat_w = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.dev_state = torch.zeros((len(self.param_groups), _STATE), dtype=torch.float32, device=dev)
        self.segments: List[tuple] = []
        off = 0
        for gi, g in enumerate(self.param_groups):
            begin = off
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                k = p.numel()
                w = self.flat_w[off:off + k].view_as(p)
                w.copy_(p.data)
                p.data = w                                            # the parameter now lives in the flat buffer
                self.state[p] = {"step": self.dev_state[gi, 0], "exp_avg": self.flat_m[off:off + k].view_as(p),
                                 "exp_avg_sq": self.flat_v[off:off + k].view_as(p)}
                off += (k + 3) // 4 * 4
            self.segments.append((begin, off))
        self.grads.bind()
        for g in self.param_groups:                  # what a torch.optim.Adam resuming from state_dict() should do
            g["capturable"] = dev.type == "cuda"
        self._lrs = None
        self.sync_lr()

    def sync_lr(self):
        """host -> device copy of the groups' learning rates when a scheduler changed them (call outside graph capture)"""
        lrs = [float(g["lr"]) for g in self.param_groups]
        if lrs != self._lrs:
            self.dev_state[:, 3] = torch.tensor(lrs, dtype=torch.float32).to(self.dev_state.device)
            self._lrs = lrs

    # -- torch.optim.Optimizer surface
    def zero_grad(self, set_to_none: bool = True):
        """one memset; the .grad views stay bound (set_to_none would unbind them and is ignored)"""
        if not self.grads.bound():
            self.grads.bind()
        self.flat_g.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        assert self.grads.bound(), "FlatAdam: a .grad was replaced (zero_grad(set_to_none=True) of another owner?)"
        L, s = hip.lib(), hip.stream_ptr()
        b1, b2 = self.defaults["betas"]
        eps = self.defaults["eps"]
        st = self.dev_state.data_ptr()
        hip.check(L.mrfa_adam_prepare(s, st, len(self.param_groups), b1, b2), "adam_prepare")
        for gi, (g, (b, e)) in enumerate(zip(self.param_groups, self.segments)):
            if g.get("clip"):
                hip.check(L.mrfa_grad_absmax(s, self.flat_g.data_ptr() + 4 * b, e - b, st + 4 * _STATE * gi, 0), "grad_absmax")
        for gi, (g, (b, e)) in enumerate(zip(self.param_groups, self.segments)):
            clip = g.get("clip")
            hip.check(L.mrfa_adam_flat(s, self.flat_w.data_ptr() + 4 * b, self.flat_g.data_ptr() + 4 * b, self.flat_m.data_ptr() + 4 * b,
                                       self.flat_v.data_ptr() + 4 * b, e - b, st + 4 * _STATE * gi, b1, b2, eps, float(self.grad_scale),
                                       0 if clip else -1, float(clip or 0.0)), "adam_flat")
        _increment_version(self.grads.params)

    def state_dict(self):
        sd = super().state_dict()
        # copies, detached from the flat buffers (the inner dicts torch returns ARE self.state's: build new ones)
        sd["state"] = {pid: {k: v.clone() for k, v in st.items()} for pid, st in sd["state"].items()}
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """accepts a torch.optim.Adam (or FlatAdam) state_dict: moments are copied INTO the flat buffers"""
        groups = state_dict["param_groups"]
        assert len(groups) == len(self.param_groups) and all(len(a["params"]) == len(b["params"]) for a, b in zip(groups, self.param_groups))
        for gi, (saved, mine) in enumerate(zip(groups, self.param_groups)):
            mine["lr"] = saved["lr"]
            steps = []
            for pid, p in zip(saved["params"], mine["params"]):
                st = state_dict["state"].get(pid)
                if st is None:
                    continue
                self.state[p]["exp_avg"].copy_(st["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                steps.append(float(st["step"]))
            if steps:
                assert max(steps) == min(steps), "FlatAdam keeps one step count per parameter group"
                self.dev_state[gi, 0] = steps[0]
        self._lrs = None
        self.sync_lr()


def clip_coefficients(opt: FlatAdam) -> List[float]:
    """(diagnostics) the clip coefficient each group used in the last step"""
    out = []
    for gi, g in enumerate(opt.param_groups):
        clip = g.get("clip")
        total = float(opt.dev_state[gi, 4]) * opt.grad_scale
        out.append(min(1.0, clip / (total + 1e-6)) if clip else 1.0)
    return out


__all__ = ["FlatAdam", "clip_coefficients"]
