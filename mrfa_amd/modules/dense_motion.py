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
        has_jac = jd is not None

        has_bg = bg is not None

        def prep(kd_, ks_, *rest):
            jd_, js_ = (rest[0], rest[1]) if has_jac else (None, None)
            bg_ = rest[-1] if has_bg else None
            # heat-map differences (dense_motion.py:36-46) and sparse motions (:48-76), NHWC-friendly layouts
            heat = kp2gaussian(kd_, (h, w), var) - kp2gaussian(ks_, (h, w), var)                   # (B,K,h,w)
            heat = torch.cat([torch.zeros_like(heat[:, :1]), heat], dim=1)                         # (B,K1,h,w)
            ident = make_coordinate_grid((h, w), kd_).view(1, 1, h, w, 2)
            z = ident - kd_.view(b, -1, 1, 1, 2)
            if jd_ is not None:
                # closed-form 2x2 inverse (reference: torch.inverse, dense_motion.py:54): no solver library call, so the
                # island stays free of host synchronisation and can be captured in a hipGraph
                a, b_, c_, d = jd_[..., 0, 0], jd_[..., 0, 1], jd_[..., 1, 0], jd_[..., 1, 1]
                det = a * d - b_ * c_
                inv = torch.stack([torch.stack([d, -b_], dim=-1), torch.stack([-c_, a], dim=-1)], dim=-2) / det[..., None, None]
                jac = torch.matmul(js_, inv)
                z = torch.einsum("bkij,bkhwj->bkhwi", jac, z)
            d2s = z + ks_.view(b, -1, 1, 1, 2)
            bg_grid = ident.expand(b, 1, h, w, 2)
            if bg_ is not None:                                                                    # dense_motion.py:69-73
                hom = torch.cat([bg_grid, torch.ones_like(bg_grid[..., :1])], dim=-1)
                hom = torch.matmul(bg_.view(b, 1, 1, 1, 3, 3), hom.unsqueeze(-1)).squeeze(-1)
                bg_grid = hom[..., :2] / hom[..., 2:3]
            motions = torch.cat([bg_grid, d2s], dim=1)                                             # (B,K1,h,w,2)
            return [heat, motions.reshape(b * k1, h, w, 2)]
        ins = [kd, ks] + ([jd, js] if has_jac else []) + ([bg] if has_bg else [])
        heat, motions = e.island(prep, ins)
        deformed = e.grid_sample(src, motions.view(), 0, in_rep=k1, need_din=False)               # (B*K1,h,w,3)

        def assemble(heat_, deformed_):
            d = deformed_.reshape(b, k1, h, w, c).permute(0, 2, 3, 1, 4)                            # (B,h,w,K1,c)
            return [torch.cat([heat_.permute(0, 2, 3, 1).unsqueeze(-1), d], dim=-1).reshape(b, h, w, k1 * (c + 1))]
        (inp,) = e.island(assemble, [heat, deformed])
        pred = self.hourglass.run(e, inp.view())
        logit = e.conv(pred, self.mask)                                                              # (B,h,w,K1)
        occ = e.conv(pred, self.occlusion) if self.occlusion is not None else None

        def combine(logit_, motions_):
            mask = F.softmax(logit_, dim=-1)                                                         # (B,h,w,K1)
            m = motions_.reshape(b, k1, h, w, 2).permute(0, 2, 3, 1, 4)                              # (B,h,w,K1,2)
            deformation = (m * mask.unsqueeze(-1)).sum(dim=3)                                        # (B,h,w,2)
            return [deformation, mask.permute(0, 3, 1, 2), logit_.permute(0, 3, 1, 2)]
        deformation, mask, logit_nchw = e.island(combine, [logit, motions])

        def export_deformed(d_):
            return [d_.reshape(b, k1, h, w, c).permute(0, 1, 4, 2, 3)]
        (sparse_deformed,) = e.island(export_deformed, [deformed])
        outs = [sparse_deformed, logit_nchw, mask, deformation]
        if occ is not None:
            (occ_out,) = e.island(lambda o: [o.permute(0, 3, 1, 2)], [occ])
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
