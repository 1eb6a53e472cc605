"""The reference's generator training losses on the HIP engine.
reference: modules/model.py -- Transform 26-76, Vgg19 79-121, ImagePyramide 123-141, MRFA.forward loss wiring 219-254.

  perceptual    sum over scales {1, .5, .25, .125} and the 5 VGG19 slices of  w_i * mean|vgg_i(pyr(gen)) - vgg_i(pyr(driving))|
  equivariance  |kp_d - T(kp(T-warped driving))| and the Jacobian variant, T a random affine + thin-plate-spline warp

The perceptual pyramid is ONE engine program: AntiAliasInterpolation2d (K18, with its gradient K22) -> per-channel
normalisation -> VGG19 (16 3x3 convolutions on K1 with bias + ReLU in the epilogue, 2x2 max-pooling K22) -> mean|x - y| per
slice reduced by one kernel each (K22).  The real image's branch runs without a tape; the VGG weights are frozen, so the
backward is the data-gradient chain only.  Pretrained VGG19 weights (torchvision, downloaded by the reference) do not exist
offline: `Vgg19(state_dict=...)` takes the `state_dict()` of the reference's Vgg19 module (same parameter names); without it
the weights are a deterministic random initialisation and only the ARITHMETIC of the loss is comparable (tests use the same
weights on both sides).  Transform and the equivariance terms act on (B,10,2) tensors and one image warp: torch device ops.
The background term (model.py:248-253) uses mrfa_amd.modules.BGMotionPredictor (resnet18 on the engine).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import grad

from .engine import Ctx, View, run_program
from .modules.util import AntiAliasInterpolation2d, make_coordinate_grid

# torchvision.models.vgg19().features: index -> layer ('M' = MaxPool2d(2, 2)); convs are 3x3 pad 1, each followed by ReLU
_VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
_SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))              # model.py:91-100


def _vgg19_features() -> List[nn.Module]:
    layers, cin = [], 3
    for v in _VGG19_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return layers


class Vgg19(nn.Module):
    """relu1_1 / relu2_1 / relu3_1 / relu4_1 / relu5_1 of VGG19 on ImageNet-normalised input.  reference: model.py:79-121
    (same state_dict: slice<k>.<torchvision feature index>.{weight,bias}, mean, std)."""

    def __init__(self, requires_grad=False, state_dict: Optional[dict] = None):
        super().__init__()
        feats = _vgg19_features()
        for k, (lo, hi) in enumerate(_SLICES):
            sl = nn.Sequential()
            for i in range(lo, hi):
                sl.add_module(str(i), feats[i])
            setattr(self, f"slice{k + 1}", sl)
        self.mean = nn.Parameter(torch.tensor([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1), requires_grad=False)
        self.std = nn.Parameter(torch.tensor([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1), requires_grad=False)
        if state_dict is not None:
            self.load_state_dict(state_dict)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def _norm_consts(self):
        """(1/std per channel as Python floats, -mean/std as a device tensor), cached: reading them is a host round trip that
        must not happen inside a hipGraph capture"""
        key = (self.mean._version, self.std._version, self.mean.device)
        if getattr(self, "_nc_key", None) != key:
            inv = [1.0 / v for v in self.std.detach().reshape(3).tolist()]
            shift = (-self.mean.detach() / self.std.detach()).reshape(3).contiguous()
            object.__setattr__(self, "_nc", (inv, shift))
            object.__setattr__(self, "_nc_key", key)
        return self._nc

    def run(self, e: Ctx, x: View) -> List[View]:
        """x: (B,H,W,3) un-normalised image view -> the five slice outputs"""
        inv_std, shift = self._norm_consts()
        xn = e.new(x.N, x.H, x.W, 3)
        for c in range(3):                                              # (X - mean) / std   (model.py:110)
            e.copy(x.slice(c, c + 1), out=xn.slice(c, c + 1), mul=inv_std[c])
        e._chk(e.L.mrfa_bias_act(e.s, xn.ptr, xn.ld, xn.rows, 3, shift.data_ptr(), 0, xn.ptr, xn.ld, None), "vgg normalise")
        outs, y = [], xn
        for k in range(5):
            mods = list(getattr(self, f"slice{k + 1}"))
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.Conv2d):                            # conv + the ReLU that follows it, one launch
                    y = e.conv(y, m, relu=True)
                    i += 2
                else:
                    y = e.maxpool2(y)
                    i += 1
            outs.append(y)
        return outs

    def forward(self, X):
        def program(e: Ctx, xin):
            outs = self.run(e, e.from_nchw(xin))
            return tuple(e.to_nchw(o) for o in outs), tuple((lambda g, o=o: e.seed_grad_nchw(o, g)) for o in outs), (None,)
        return list(run_program(self, program, [X]))


class ImagePyramide(nn.Module):
    """{'prediction_<scale>': AntiAliasInterpolation2d(scale)(x)}.  reference: model.py:123-141"""

    def __init__(self, scales, num_channels=3):
        super().__init__()
        self.downs = nn.ModuleDict({str(s).replace('.', '-'): AntiAliasInterpolation2d(num_channels, s) for s in scales})

    def forward(self, x):
        return {'prediction_' + k.replace('-', '.'): m(x) for k, m in self.downs.items()}


class PerceptualLoss(nn.Module):
    """sum_scales sum_i w_i * mean|vgg_i(pyr_s(gen)) - vgg_i(pyr_s(real))|   (model.py:219-229) as one engine program"""

    def __init__(self, scales, weights, vgg: Optional[Vgg19] = None, num_channels=3):
        super().__init__()
        self.scales = list(scales)
        self.weights = list(weights)
        self.pyramid = ImagePyramide(self.scales, num_channels)
        self.vgg = vgg if vgg is not None else Vgg19()

    def _program(self, e: Ctx, gen: torch.Tensor, real: torch.Tensor):
        dev = gen.device
        gen = gen.contiguous().float()
        real = real.contiguous().float()
        dgen = torch.zeros_like(gen) if e.record else None
        ey = Ctx(dev, train=False, record=False)                        # the real image's branch: constants, no tape
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        gscale = torch.ones(1, dtype=torch.float32, device=dev)
        for si, s in enumerate(self.scales):
            down = self.pyramid.downs[str(s).replace('.', '-')]
            if s == 1:
                xv, yv = e.from_nchw(gen), ey.from_nchw(real)
                if e.record:
                    e.tape.append(lambda xv=xv: e._chk(e.L.mrfa_nhwc_to_nchw(e.s, xv.gptr, xv.ld, dgen.data_ptr(), xv.N, xv.C, xv.H, xv.W, 1),
                                                       "d(gen) scale 1") if xv.has_grad else None)
            else:
                stride = int(round(1.0 / s))
                xv = e.antialias_down(gen, down.weight, stride, dimg=dgen)
                yv = ey.antialias_down(real, down.weight, stride)
            fy = self.vgg.run(ey, yv)
            fx = self.vgg.run(e, xv)
            for i, w in enumerate(self.weights):
                if w == 0:
                    continue
                e.l1_diff(fx[i], fy[i], acc, float(w) / float(fx[i].rows * fx[i].C), gscale)
        loss = acc[0].float()

        def seed(g):
            torch.mul(g.reshape(1).float(), 1.0, out=gscale)
        return (loss,), (seed,), ((lambda: dgen), None)

    def forward(self, generated: torch.Tensor, real: torch.Tensor) -> torch.Tensor:
        return run_program(self, self._program, [generated, real])[0]


class Transform:
    """Random affine + thin-plate-spline warp of the equivariance constraint.  reference: model.py:26-76 (same attributes:
    theta (B,2,3), control_points (1,P*P,2), control_params (B,1,P*P)); `generator` makes the draw reproducible in tests."""

    def __init__(self, bs, generator: Optional[torch.Generator] = None, device=None, **kwargs):
        # drawn on `device` (no host tensors: a hipGraph capture cannot copy from pageable host memory; the device generator is
        # graph-safe), same distributions as the reference's CPU draws
        # (torch.normal(mean, std_tensor) validates std with a host round trip, which a capture forbids: randn * sigma instead)
        noise = torch.randn([bs, 2, 3], device=device, generator=generator) * kwargs['sigma_affine']
        self.theta = noise + torch.eye(2, 3, device=device).view(1, 2, 3)
        self.bs = bs
        self.tps = ('sigma_tps' in kwargs) and ('points_tps' in kwargs)
        if self.tps:
            p = kwargs['points_tps']
            self.control_points = make_coordinate_grid((p, p), noise).reshape(1, p * p, 2)
            self.control_params = torch.randn([bs, 1, p ** 2], device=device, generator=generator) * kwargs['sigma_tps']

    def transform_frame(self, frame):
        h, w = frame.shape[2:]
        grid = make_coordinate_grid((h, w), frame).reshape(1, h * w, 2)
        grid = self.warp_coordinates(grid).view(self.bs, h, w, 2).contiguous().float()
        # F.grid_sample(frame, grid, padding_mode="reflection") (model.py:48; align_corners=False) as one launch of the library
        from . import hip
        frame = frame.contiguous().float()
        out = torch.empty((self.bs, frame.shape[1], h, w), dtype=torch.float32, device=frame.device)
        hip.check(hip.lib().mrfa_warp_frame_reflect(hip.stream_ptr(), frame.data_ptr(), self.bs, frame.shape[1], h, w, grid.data_ptr(), h, w,
                                                    out.data_ptr()), "warp_frame_reflect")
        return out

    def warp_coordinates(self, coordinates):
        theta = self.theta.to(coordinates).unsqueeze(1)
        out = (torch.matmul(theta[:, :, :, :2], coordinates.unsqueeze(-1)) + theta[:, :, :, 2:]).squeeze(-1)
        if self.tps:
            cp = self.control_points.to(coordinates)
            d = torch.abs(coordinates.view(coordinates.shape[0], -1, 1, 2) - cp.view(1, 1, -1, 2)).sum(-1)      # L1 distances
            r = (d ** 2) * torch.log(d + 1e-6) * self.control_params.to(coordinates)
            out = out + r.sum(dim=2).view(self.bs, coordinates.shape[1], 1)
        return out

    def jacobian(self, coordinates):
        new = self.warp_coordinates(coordinates)
        gx = grad(new[..., 0].sum(), coordinates, create_graph=True)
        gy = grad(new[..., 1].sum(), coordinates, create_graph=True)
        return torch.cat([gx[0].unsqueeze(-2), gy[0].unsqueeze(-2)], dim=-2)


def _inv2x2(m):
    a, b, c, d = m[..., 0, 0], m[..., 0, 1], m[..., 1, 0], m[..., 1, 1]
    det = a * d - b * c
    return torch.stack([torch.stack([d, -b], dim=-1), torch.stack([-c, a], dim=-1)], dim=-2) / det[..., None, None]


class GeneratorFullLoss(nn.Module):
    """loss_values of MRFA.forward(is_train=True), model.py:219-246 (without the background term): 'perceptual', 'equivariance',
    'equivariance_jacobian'.  `encoder` is the model's keypoint encoder (third pass on the warped driving frame, model.py:234)."""

    def __init__(self, train_params: dict, vgg: Optional[Vgg19] = None):
        super().__init__()
        self.train_params = train_params
        self.scales = train_params['scales']
        self.loss_weights = train_params['loss_weights']
        self.perceptual = PerceptualLoss(self.scales, self.loss_weights['perceptual'], vgg) if sum(self.loss_weights['perceptual']) != 0 else None

    def forward(self, encoder: nn.Module, driving: torch.Tensor, generated: torch.Tensor, kp_driving: Dict[str, torch.Tensor],
                transform: Optional[Transform] = None, bg_param: Optional[torch.Tensor] = None,
                bg_param_reverse: Optional[torch.Tensor] = None, transformed_kp: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        """transformed_kp: the encoder's output on transform.transform_frame(driving) when the caller already ran that pass"""
        out = {}
        if self.perceptual is not None:
            out['perceptual'] = self.perceptual(generated, driving)
        w = self.loss_weights
        if w['equivariance'] != 0:
            if transform is None:
                transform = Transform(driving.shape[0], device=driving.device, **self.train_params['transform_params'])
            if transformed_kp is None:
                transformed_kp = encoder(transform.transform_frame(driving))
            value = torch.abs(kp_driving['kp'] - transform.warp_coordinates(transformed_kp['kp'])).mean()
            out['equivariance'] = w['equivariance'] * value
            if w['equivariance_jacobian'] != 0:
                jt = torch.matmul(transform.jacobian(transformed_kp['kp']), transformed_kp['jacobian'])
                value = torch.matmul(_inv2x2(kp_driving['jacobian']), jt)
                eye = torch.eye(2, device=value.device, dtype=value.dtype).view(1, 1, 2, 2)
                # model.py:245 keeps the un-reduced |I - J| tensor; train.py:62 then takes .mean() of every loss value
                out['equivariance_jacobian'] = w['equivariance_jacobian'] * torch.abs(eye - value)
        if bg_param is not None:                                           # model.py:248-253: forward o reverse background motion = identity
            value = torch.matmul(bg_param, bg_param_reverse)
            eye = torch.eye(3, device=value.device, dtype=value.dtype).view(1, 3, 3)
            out['bg'] = 10 * torch.abs(eye - value).mean()
        return out
