from .hr_base import HRNET_base
from .pose_tokenpose_b import TokenPose_B, get_pose_net
from .tokenpose_base import TokenPose_TB_base

__all__ = ["HRNET_base", "TokenPose_B", "TokenPose_TB_base", "get_pose_net"]
