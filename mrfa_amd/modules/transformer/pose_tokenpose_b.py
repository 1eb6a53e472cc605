"""TokenPose_B, the MTIA prior's keypoint / Jacobian encoder (the default `prior_model: mtia` of vox1.yaml:67 and
celebvhq.yaml).  reference: modules/transformer/pose_tokenpose_b.py:16-61.  One engine program per forward: HRNet stem
and token transformer share the tape, so the whole encoder is a single autograd node like KPDetector."""
from __future__ import annotations

import torch
from torch import nn

from ...engine import Ctx, run_program
from .hr_base import HRNET_base
from .tokenpose_base import TokenPose_TB_base


class TokenPose_B(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.cfg = cfg
        m = cfg.MODEL
        self.pre_feature = HRNET_base(cfg, **kwargs)
        self.transformer = TokenPose_TB_base(
            feature_size=[m.IMAGE_SIZE[1] // 4, m.IMAGE_SIZE[0] // 4], patch_size=[m.PATCH_SIZE[1], m.PATCH_SIZE[0]],
            num_keypoints=m.NUM_JOINTS, dim=m.DIM, channels=m.BASE_CHANNEL, depth=m.TRANSFORMER_DEPTH, heads=m.TRANSFORMER_HEADS,
            mlp_dim=m.DIM * m.TRANSFORMER_MLP_RATIO, apply_init=m.INIT, hidden_heatmap_dim=m.HEATMAP_SIZE[1] * m.HEATMAP_SIZE[0] // 8,
            heatmap_dim=m.HEATMAP_SIZE[1] * m.HEATMAP_SIZE[0], heatmap_size=[m.HEATMAP_SIZE[1], m.HEATMAP_SIZE[0]],
            pos_embedding_type=m.POS_EMBEDDING_TYPE, estimate_jacobian=m.ESTIMATE_JACOBIAN, temperature=m.TEMPERATURE,
            fix_img2motion_attention=m.FIX_IMG2MOTION_ATTENTION)

    def _program(self, e: Ctx, x: torch.Tensor):
        outs = self.transformer.run(e, self.pre_feature.run(e, e.from_nchw(x)))
        return tuple(o.t for o in outs), tuple(o.add_grad for o in outs), (None,)

    def forward(self, x):
        if self.cfg.MODEL.DATA_PREPROCESS:                       # ImageNet normalisation (:41-47)
            mean = torch.tensor([0.485, 0.456, 0.406], device=x.device).view(1, 3, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225], device=x.device).view(1, 3, 1, 1)
            x = (x - mean) / std
        outs = run_program(self, self._program, [x])
        out = {'kp': outs[0]}
        if self.transformer.mlp_head_jacobian is not None:
            out['jacobian'] = outs[1]
        return out

    def init_weights(self, pretrained=''):
        self.pre_feature.init_weights(pretrained)


def get_pose_net(cfg, is_train, **kwargs):
    model = TokenPose_B(cfg, **kwargs)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
