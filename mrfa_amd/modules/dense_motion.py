"""DenseMotionNetwork.  reference: modules/dense_motion.py:8-146."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ..engine import Ctx, run_program
from .util import AntiAliasInterpolation2d, Hourglass, kp2gaussian, make_coordinate_grid


class DenseMotionNetwork(nn.Module):
    """Dense prior motion from sparse keypoint motions.  Same kwargs / state_dict / output dict keys as the reference
    (`sparse_deformed, logit_mask, mask, deformation, occlusion`; occlusion are LOGITS, dense_motion.py:142-143)."""

    def __init__(self, block_expansion, num_blocks, max_features, num_kp, num_channels, estimate_occlusion_map=True,
                 scale_factor=1, kp_variance=0.01):
        super().__init__()
        infeatures = num_kp + 1
        self.infeatures = infeatures
        self.hourglass = Hourglass(block_expansion=block_expansion, in_features=infeatures * (num_channels + 1),
                                   max_features=max_features, num_blocks=num_blocks)
        self.mask = nn.Conv2d(self.hourglass.out_filters, infeatures, kernel_size=(7, 7), padding=(3, 3))
        self.occlusion = nn.Conv2d(self.hourglass.out_filters, 1, kernel_size=(7, 7), padding=(3, 3)) if estimate_occlusion_map else None
        self.num_kp = num_kp
        self.scale_factor = scale_factor
        self.kp_variance = kp_variance
        if self.scale_factor != 1:
            self.down = AntiAliasInterpolation2d(num_channels, self.scale_factor)

    def _program(self, e: Ctx, source_image, kd, ks, jd=None, js=None, bg=None):
        src = self.down.run(e, source_image) if self.scale_factor != 1 else e.from_nchw(source_image)   # (B,h,w,3)
        b, h, w, c = src.N, src.H, src.W, src.C
        k1 = self.num_kp + 1
        var = self.kp_variance
        # K15 + K16 + K10 in one launch (csrc/prior.hip): heat-map differences (dense_motion.py:36-46), the K+1 sparse motions with the
        # closed-form 2x2 inverse and the optional background affine (:48-76), the K+1 warps of the 1/4-scale source (:78-85), written
        # straight into the interleaved hourglass input [heat_k | warp_k(3)] (:117-119)
        ins = [kd, ks] + ([jd, js] if jd is not None else []) + ([bg] if bg is not None else [])
        motions, inp, sparse_deformed = e.prior_motion(src, kd, ks, jd, js, bg, var)
        pred = self.hourglass.run(e, inp)
        logit = e.conv(pred, self.mask)                                                              # (B,h,w,K1)
        occ = e.conv(pred, self.occlusion) if self.occlusion is not None else None
        # K14 + K17: softmax over the K+1 motions, mask-weighted deformation, NCHW exports of mask / logits (dense_motion.py:129-136)
        deformation, mask, logit_nchw = e.softmax_combine(logit, motions)
        outs = [sparse_deformed, logit_nchw, mask, deformation]
        if occ is not None:
            (occ_out,) = e.island(lambda o: [o.permute(0, 3, 1, 2)], [occ])                          # (B,h,w,1) -> (B,1,h,w): a view
            outs.append(occ_out)
        in_grads = [None] + [(lambda t=t: e.ext_grads.get(id(t))) for t in ins]
        return tuple(o.t for o in outs), tuple(o.add_grad for o in outs), tuple(in_grads)

    def forward(self, source_image, kp_driving, kp_source, bg_param=None, dropout_flag=False, dropout_p=0):
        if dropout_flag:
            raise NotImplementedError("dropout_softmax belongs to the TPSM prior (out of scope)")
        ins = [source_image, kp_driving['kp'], kp_source['kp']]
        has_jac = 'jacobian' in kp_driving
        if has_jac:
            ins += [kp_driving['jacobian'], kp_source['jacobian']]
        if bg_param is not None:
            ins.append(bg_param)
        prog = self._program
        if bg_param is not None and not has_jac:
            prog = lambda e, s_, kd_, ks_, bg_: self._program(e, s_, kd_, ks_, None, None, bg_)
        outs = run_program(self, prog, ins)
        out_dict = {'sparse_deformed': outs[0], 'logit_mask': outs[1], 'mask': outs[2], 'deformation': outs[3]}
        if self.occlusion is not None:
            out_dict['occlusion'] = outs[4]
        return out_dict


class TPSDenseMotionNetwork(nn.Module):
    """Name kept importable for `from .dense_motion import DenseMotionNetwork, TPSDenseMotionNetwork` (reference modules/model.py:20);
    the TPSM prior's dense-motion network (dense_motion.py:150-312) is out of scope."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("TPSDenseMotionNetwork (prior_model: tpsm) is out of scope of the MI355X hot path")
