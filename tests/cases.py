"""Deterministic test cases shared by tools/make_goldens.py (which runs the REFERENCE on them in the build
container) and the parity tests (which run the oracle and the HIP path on them).  Inputs and weights are pure
functions of their names (mrfa_amd.utils.prng), so only the reference OUTPUTS are stored in tests/golden/."""
from __future__ import annotations

import copy

import torch

from mrfa_amd.utils.prng import det_normal, det_uniform, fill_state_dict, fill_tokenpose_state_dict

# vox1.yaml kwargs (config/vox1.yaml:17-64 in the reference), restated as literals
KP_DETECTOR_CFG = dict(block_expansion=32, num_kp=10, num_channels=3, max_features=1024, num_blocks=5,
                       temperature=0.1, scale_factor=0.25, estimate_jacobian=True, estimate_occlusion=False)
DENSE_MOTION_CFG = dict(block_expansion=64, max_features=1024, num_blocks=5, scale_factor=0.25, num_kp=10,
                        num_channels=3, estimate_occlusion_map=True)
RAFT_FLOW_CFG = dict(prior_only=False, num_kp=10, dim=256, size=256,
                     generator=dict(num_channels=3, block_expansion=64, max_features=512, num_up_blocks=5),
                     driving_encoder=dict(in_features=10, block_expansion=32, max_features=512, num_blocks=5),
                     source_encoder=dict(in_features=13, block_expansion=32, max_features=512, num_blocks=5))


def raft_cfg(size=256, prior_only=False):
    c = copy.deepcopy(RAFT_FLOW_CFG)
    c["size"] = size
    c["prior_only"] = prior_only
    if size == 64:          # 16x16 basic resolution: hourglass depth 4 (depth 5 would pool 1x1 -> 0x0)
        c["driving_encoder"]["num_blocks"] = 4
        c["source_encoder"]["num_blocks"] = 4
    return c


def images(tag: str, b: int, size: int):
    """Smooth-ish deterministic images in [0,1): low-res noise upsampled + fine noise (so warps matter)."""
    lo = det_uniform(f"{tag}/lo", (b, 3, size // 8, size // 8), 0.0, 1.0)
    hi = det_uniform(f"{tag}/hi", (b, 3, size, size), 0.0, 1.0)
    up = torch.nn.functional.interpolate(lo, size=(size, size), mode="bilinear", align_corners=False)
    return (0.75 * up + 0.25 * hi).contiguous()


def keypoints(tag: str, b: int, k: int = 10):
    kp = det_uniform(f"{tag}/kp", (b, k, 2), -0.8, 0.8)
    jac = torch.eye(2).view(1, 1, 2, 2) + 0.1 * det_normal(f"{tag}/jac", (b, k, 2, 2))
    return {"kp": kp, "jacobian": jac}


def synthetic_dense_motion(tag: str, b: int, h: int):
    """A prior deformation grid with multi-pixel, partly out-of-bounds displacements + occlusion logits."""
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, h), indexing="ij")
    ident = torch.stack([xs, ys], dim=-1)[None].expand(b, h, h, 2)
    lo = det_uniform(f"{tag}/dlo", (b, 2, 4, 4), -0.25, 0.25)
    disp = torch.nn.functional.interpolate(lo, size=(h, h), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    deformation = (ident + disp + det_uniform(f"{tag}/dhi", (b, h, h, 2), -0.02, 0.02)).contiguous()
    occ_lo = det_uniform(f"{tag}/olo", (b, 1, 8, 8), -2.0, 2.0)
    occlusion = torch.nn.functional.interpolate(occ_lo, size=(h, h), mode="bilinear", align_corners=True).contiguous()
    return {"deformation": deformation, "occlusion": occlusion}


def weights_for(sd_like: dict, tag: str, flow_head_gain: float = 0.3) -> dict:
    """Deterministic weights for a module's state_dict; the RAFT flow/occlusion heads can be scaled so that the
    per-level delta-flow spans pixels (SURVEY section 8c fixture recipe)."""
    sd = fill_state_dict(sd_like, tag=tag)
    for name in list(sd):
        if name.endswith("jacobian.weight"):
            sd[name] = sd[name] * 0.05
        if name.endswith("jacobian.bias"):
            sd[name] = torch.tensor([1.0, 0.0, 0.0, 1.0]) + sd[name] * 0.5
        if name.endswith(("refine.conv2.weight", "refine.convo2.weight")):
            sd[name] = sd[name] * flow_head_gain
    return sd


# vox1.yaml `mtia_kp_detector` (config/vox1.yaml:117-186 in the reference; identical in celebvhq.yaml): the literals live in
# mrfa_amd.train.VOX1, which bench.py uses too
from mrfa_amd.train import VOX1 as _VOX1  # noqa: E402

TOKENPOSE_CFG = _VOX1["mtia_kp_detector"]


def tokenpose_cfg(image_size: int = 256, depth: int = 12):
    c = copy.deepcopy(TOKENPOSE_CFG)
    c["MODEL"]["IMAGE_SIZE"] = [image_size, image_size]
    c["MODEL"]["HEATMAP_SIZE"] = [image_size // 4, image_size // 4]
    c["MODEL"]["TRANSFORMER_DEPTH"] = depth
    return c


def tokenpose_weights(sd_like: dict, tag: str) -> dict:
    """deterministic TokenPose_B weights (mrfa_amd.utils.prng.fill_tokenpose_state_dict)"""
    return fill_tokenpose_state_dict(sd_like, tag)


def vgg_weights(sd_like: dict) -> dict:
    """deterministic VGG19 weights for the loss tests (the pretrained ones do not exist offline): He-uniform convolutions with a
    small positive bias so that ReLU keeps about half of every feature map alive through 16 layers; mean / std stay analytic"""
    sd = fill_state_dict({k: v for k, v in sd_like.items() if k not in ("mean", "std")}, tag="vgg")
    for k in list(sd):
        if k.endswith(".bias"):
            sd[k] = sd[k].abs() * 0.5
    sd["mean"], sd["std"] = sd_like["mean"].detach().clone(), sd_like["std"].detach().clone()
    return sd


def bg_weights(sd_like: dict) -> dict:
    """deterministic BGMotionPredictor (resnet18) weights: He-uniform convolutions, BatchNorm scales in [0.9, 1.1], perturbed running
    statistics, and a small random fc around the identity transform (zero-initialised in the reference) so that the regressed
    affine is a real one"""
    sd = fill_state_dict(sd_like, tag="bg")
    for name, t in sd_like.items():
        if t.dim() == 1 and name.endswith(".weight"):
            sd[name] = det_uniform(f"bg:{name}", tuple(t.shape), 0.9, 1.1)
        elif name.endswith("fc.weight"):
            sd[name] = sd[name] * 0.2
        elif name.endswith("fc.bias"):
            sd[name] = torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0]) + sd[name] * 0.3
    return sd


# ---- the headline program: the MTIA-prior training step (BASELINE config 2), shared by tools/make_goldens.py:g12_chain_mtia, the CPU
# emulator leg and the hipGraph-replay test
def mtia_chain_weights(encoder_sd: dict, dense_motion_sd: dict, decoder_sd: dict) -> dict:
    """{"encoder.": sd, "dense_motion.": sd, "decoder.": sd}: deterministic weights of the TokenPose_B -> DenseMotion -> RaftFlow chain"""
    return {"encoder.": tokenpose_weights(encoder_sd, "g12/enc"), "dense_motion.": weights_for(dense_motion_sd, "g12/dm"),
            "decoder.": weights_for(decoder_sd, "g12/rf")}


def sample_index(numel: int, k: int = 64) -> torch.Tensor:
    """the <= k flat positions of a tensor that the gradient goldens keep (evenly spread, first and last included)"""
    if numel <= k:
        return torch.arange(numel)
    return torch.linspace(0, numel - 1, k).round().long()
