"""KPDetector (FOMM prior).  reference: modules/kp_detector.py:17-133."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ..engine import Ctx, run_program
from .util import AntiAliasInterpolation2d, DownBlock2d, Hourglass, make_coordinate_grid


class KPDetector(nn.Module):
    """Keypoints (B,K,2) in [-1,1] + heat-map-weighted 2x2 Jacobians.  Same kwargs / state_dict as the reference
    (kp_detector.py:22-53): predictor.{encoder,decoder}.*, kp, jacobian, kp_occlusion.{0..3}.{conv,norm}.* / kp_occlusion.4, down.weight.
    estimate_occlusion (kp_detector.py:41-48,124-128; off in both reference YAMLs): four DownBlock2d + a 4x4 / stride-4 convolution + sigmoid over the
    hourglass features -> out['kp_occlusion'] (B, K, h / 64, w / 64) -- on the 4 x 4 grid that 256 x 256 inputs leave (the reference's geometry) the strided convolution
    is ONE output position, i.e. the plain 4x4 launch; other input sizes raise."""

    def __init__(self, block_expansion=32, num_kp=15, num_channels=3, max_features=1024, num_blocks=5, temperature=0.1,
                 scale_factor=0.25, estimate_jacobian=False, estimate_occlusion=False):
        super().__init__()
        self.predictor = Hourglass(block_expansion, in_features=num_channels, max_features=max_features, num_blocks=num_blocks)
        self.kp = nn.Conv2d(self.predictor.out_filters, num_kp, kernel_size=(7, 7), padding=0)
        self.num_kp = num_kp
        self.estimate_jacobian = estimate_jacobian
        if estimate_jacobian:
            self.num_jacobian_maps = 1
            self.jacobian = nn.Conv2d(self.predictor.out_filters, 4 * self.num_jacobian_maps, kernel_size=(7, 7), padding=0)
            self.jacobian.weight.data.zero_()
            self.jacobian.bias.data.copy_(torch.tensor([1, 0, 0, 1] * self.num_jacobian_maps, dtype=torch.float))
        self.estimate_occlusion = estimate_occlusion
        if estimate_occlusion:
            be = block_expansion
            self.kp_occlusion = nn.Sequential(DownBlock2d(self.predictor.out_filters, be, kernel_size=3, padding=1),
                                              DownBlock2d(be, be * 2, kernel_size=3, padding=1), DownBlock2d(be * 2, be * 3, kernel_size=3, padding=1),
                                              DownBlock2d(be * 3, be * 4, kernel_size=3, padding=1),
                                              nn.Conv2d(be * 4, num_kp, kernel_size=(4, 4), padding=0, stride=4))
        self.temperature = temperature
        self.scale_factor = scale_factor
        if self.scale_factor != 1:
            self.down = AntiAliasInterpolation2d(num_channels, self.scale_factor)

    def _program(self, e: Ctx, x: torch.Tensor):
        xs = self.down.run(e, x) if self.scale_factor != 1 else e.from_nchw(x)
        fmap = self.predictor.run(e, xs, need_dx=False)
        logits = e.conv(fmap, self.kp)                                     # (B,58,58,K) NHWC
        jm = e.conv(fmap, self.jacobian) if self.estimate_jacobian else None       # (B,58,58,4)
        # K14 (csrc/prior.hip): spatial softmax at temperature T + soft-argmax + heat-map-weighted Jacobian pooling, kp_detector.py:90-120
        kp, jac = e.kp_head(logits, jm, self.temperature)
        outs = [kp] + ([jac] if jac is not None else [])
        ts, seeders = [o.t for o in outs], [o.add_grad for o in outs]
        if self.estimate_occlusion:
            occ = fmap
            for blk in list(self.kp_occlusion)[:4]:
                occ = blk.run(e, occ)
            if (occ.H, occ.W) != (4, 4):
                raise NotImplementedError(f"kp_occlusion: the 4x4 / stride-4 head is built for the 4 x 4 grid that 256 x 256 inputs at scale_factor 0.25 "
                                          f"leave behind the four DownBlock2d (got {occ.H} x {occ.W})")
            raw = e.conv(occ, self.kp_occlusion[4])                        # 4x4 kernel over the 4x4 grid, no padding: the one output of the stride-4 layer
            sig = e.act(raw, 2)
            ts.append(e.to_nchw(sig))
            seeders.append(lambda g, v=sig: e.seed_grad_nchw(v, g))
        return tuple(ts), tuple(seeders), (None,)

    def forward(self, x):
        outs = run_program(self, self._program, [x])
        out = {'kp': outs[0]}
        if self.estimate_jacobian:
            out['jacobian'] = outs[1]
        if self.estimate_occlusion:
            out['kp_occlusion'] = outs[-1]
        return out


class TPSKPDetector(nn.Module):
    """Name kept importable for `from .kp_detector import KPDetector, TPSKPDetector` (reference modules/model.py:19).  The TPSM prior
    (kp_detector.py:136-158: resnet18 + thin-plate splines) is in no BASELINE config and out of scope."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("TPSKPDetector (prior_model: tpsm) is out of scope of the MI355X hot path: use prior_model mtia or fomm")
