"""nn.Module surface of the MRFA hot path (same class names / kwargs / state_dict keys as the reference's
modules/ package), executing on the HIP engine."""
from .util import (AntiAliasInterpolation2d, AttributeDict, ChannelBlock2d, DownBlock2d, Hourglass, ResBlock2d,
                   SameBlock2d, UpBlock2d, batch_bilinear_sampler, bilinear_sampler, convert_dict_to_attrit_dict, coords_grid)
from .kp_detector import KPDetector
from .dense_motion import DenseMotionNetwork
from .generator import OcclusionAwareGenerator
from .raft import BasicMotionEncoder, CorrBlock, RaftFlow, RefineFlow
from .bg_motion_predictor import BGMotionPredictor
from .model import MRFA

__all__ = ["AntiAliasInterpolation2d", "AttributeDict", "ChannelBlock2d", "DownBlock2d", "Hourglass", "ResBlock2d",
           "SameBlock2d", "UpBlock2d", "convert_dict_to_attrit_dict", "KPDetector", "DenseMotionNetwork",
           "OcclusionAwareGenerator", "RaftFlow", "MRFA", "BGMotionPredictor", "CorrBlock", "BasicMotionEncoder", "RefineFlow",
           "bilinear_sampler", "batch_bilinear_sampler", "coords_grid"]
