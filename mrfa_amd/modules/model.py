"""MRFA model assembly (inference wiring).  reference: modules/model.py:145-216.

encoder (FOMM KPDetector or MTIA TokenPose_B) -> dense_motion -> decoder (RaftFlow), and with is_train=True the generator losses
of mrfa_amd/losses.py (VGG19 perceptual pyramid, equivariance).  BGMotionPredictor (resnet18) and the
background loss are wired in when train_params['bg_start'] < num_epochs (celebvhq.yaml); bench.py's headline step uses the surrogate
L1 loss SURVEY.md 8(d) defines."""
from __future__ import annotations

import torch
from torch import nn

from .dense_motion import DenseMotionNetwork
from .kp_detector import KPDetector
from .raft import RaftFlow
from .util import AntiAliasInterpolation2d


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class MRFA(nn.Module):
    """attributes used by the reference's callers: .encoder .dense_motion .decoder .down (train.py:21-24, demo.py:40-42)"""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.cfg = cfg
        train_params = _get(cfg, 'train_params')
        self.train_params = train_params
        # model.py:150-157: `pyramid` (and `vgg` when the perceptual weights are not all zero) are TOP-LEVEL sub-modules of the
        # reference's MRFA, registered before the networks: its checkpoints hold `pyramid.downs.*.weight` and `vgg.*`, and demo.py /
        # Logger.load_cpk load them with strict=True.  The loss container that uses them (mrfa_amd.losses.GeneratorFullLoss) is
        # therefore NOT a registered child: it only borrows the two modules registered here.
        object.__setattr__(self, 'losses', None)
        if 'loss_weights' in train_params and 'scales' in train_params:
            from ..losses import GeneratorFullLoss, ImagePyramide
            full = GeneratorFullLoss(train_params)
            self.scales = train_params['scales']
            self.loss_weights = train_params['loss_weights']
            if full.perceptual is not None:
                self.pyramid = full.perceptual.pyramid
                self.vgg = full.perceptual.vgg
            else:
                self.pyramid = ImagePyramide(self.scales, 3)
            object.__setattr__(self, 'losses', full)
        prior = train_params['prior_model']
        self.prior = prior
        if prior == 'fomm':
            self.encoder = KPDetector(**_get(cfg, 'fomm_kp_detector'))
            self.dense_motion = DenseMotionNetwork(**_get(cfg, 'dense_motion'))
        elif prior == 'mtia':
            from .transformer import get_pose_net
            self.encoder = get_pose_net(_get(cfg, 'mtia_kp_detector'), is_train=True)
            self.dense_motion = DenseMotionNetwork(**_get(cfg, 'dense_motion'))
        else:
            raise NotImplementedError(f"prior_model={prior!r}: 'fomm' and 'mtia' are built natively; TPSM is out of scope (SURVEY.md section 8)")
        self.bg_start = train_params['bg_start']
        if self.bg_start < train_params['num_epochs']:                     # model.py:174-176 (celebvhq.yaml: bg_start 0)
            from .bg_motion_predictor import BGMotionPredictor
            self.bg_predictor = BGMotionPredictor()
        self.decoder = RaftFlow(**_get(cfg, 'raft_flow'))
        self.down = AntiAliasInterpolation2d(3, 0.25)

    def forward(self, x, epoch=100, is_train=True):
        if self.training:
            kp_s = self.encoder(x['source'])
            kp_d = self.encoder(x['driving'])
        else:
            from ..train import encode_pair_eval
            kp_s, kp_d = encode_pair_eval(self.encoder, x['source'], x['driving'])
        img_down = self.down(x['source'])
        bg_param = self.bg_predictor(x['source'], x['driving']) if epoch >= self.bg_start else None        # model.py:189-192
        dense_motion = self.dense_motion(x['source'], kp_d, kp_s, bg_param=bg_param, dropout_flag=False, dropout_p=0)
        gen, warp_img, occlusion = self.decoder(kp_s['kp'], kp_d['kp'], dense_motion, img=img_down, img_full=x['source'])
        warp_img = torch.cat([warp_img, occlusion.repeat(1, 3, 1, 1)], dim=3)
        loss_values = {}
        if not is_train:
            return gen, warp_img, loss_values, kp_s['kp'], kp_d['kp']
        if self.losses is None:
            raise NotImplementedError("MRFA.forward(is_train=True) needs train_params['loss_weights'] / ['scales'] (model.py:148-157)")
        # model.py:219-246: perceptual pyramid + equivariance (+ Jacobian); the VGG19 weights are whatever self.losses.perceptual.vgg
        # holds (pretrained torchvision weights are not available offline: load the reference's Vgg19 state_dict into it)
        bg_rev = self.bg_predictor(x['driving'], x['source']) if bg_param is not None else None
        loss_values = self.losses(self.encoder, x['driving'], gen, kp_d, bg_param=bg_param, bg_param_reverse=bg_rev)
        return gen, warp_img, loss_values, kp_s['kp'], kp_d['kp']
