"""CPU oracle for the generator training losses  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional restatement of modules/model.py: Transform 26-76, Vgg19 79-121, ImagePyramide 123-141 and the loss wiring of
MRFA.forward 219-246 (perceptual pyramid, equivariance, equivariance_jacobian; not the background term).

VGG19 itself lives in a third-party dependency that is absent here and un-pinned in the reference (requirements.txt lists
torch==1.10.1+cu113 only; torchvision.models.vgg19(pretrained=True) is imported at model.py:11,86): its published
architecture (configuration E of Simonyan & Zisserman: 3x3 pad-1 convolutions [64,64,M,128,128,M,256x4,M,512x4,M,512x4,M],
ReLU after each, 2x2 max-pooling) is restated in `vgg19_features`; the pretrained WEIGHTS cannot be obtained offline, so that
part of the parity is pinned on arithmetic with shared deterministic weights only ("weights unpinned").

Parity pin: tests/golden/losses.npz, written by tools/make_goldens.py:g7_losses, which runs the reference's own
MRFA.forward(is_train=True) (its Vgg19 class, ImagePyramide, Transform and lines 219-246) with `models.vgg19` bound to the
restated architecture.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from .mrfa_oracle import antialias_down, batchnorm, coordinate_grid

VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
SLICE_ENDS = (2, 7, 12, 21, 30)                                    # model.py:91-100: features[0:2], [2:7], [7:12], [12:21], [21:30]


def vgg19_slices(x: torch.Tensor, P: Dict[str, torch.Tensor], pfx: str = "") -> List[torch.Tensor]:
    """model.py:108-118.  P holds slice<k>.<i>.weight / .bias (i = torchvision feature index), mean, std."""
    x = (x - P[pfx + "mean"].to(x.dtype)) / P[pfx + "std"].to(x.dtype)
    outs, idx, k = [], 0, 0
    for v in VGG19_CFG:
        if idx >= SLICE_ENDS[-1]:
            break
        if v == 'M':
            x = F.max_pool2d(x, 2)
            idx += 1
        else:
            name = f"{pfx}slice{k + 1}.{idx}"
            x = F.relu(F.conv2d(x, P[name + ".weight"].to(x.dtype), P[name + ".bias"].to(x.dtype), padding=1))
            idx += 2
        if idx == SLICE_ENDS[k]:
            outs.append(x)
            k += 1
    return outs


def image_pyramid(x: torch.Tensor, scales) -> Dict[str, torch.Tensor]:
    """model.py:123-141"""
    return {f"prediction_{s}": antialias_down(x, float(s)) for s in scales}


def perceptual(gen, real, P, scales, weights, pfx="") -> torch.Tensor:
    """model.py:219-229"""
    pg, pr = image_pyramid(gen, scales), image_pyramid(real, scales)
    total = 0
    for s in scales:
        xv, yv = vgg19_slices(pg[f"prediction_{s}"], P, pfx), vgg19_slices(pr[f"prediction_{s}"], P, pfx)
        for i, w in enumerate(weights):
            total = total + w * torch.abs(xv[i] - yv[i].detach()).mean()
    return total


def warp_coordinates(coords, theta, control_points=None, control_params=None):
    """Transform.warp_coordinates, model.py:48-68: affine part theta (B,2,3); TPS part sum_p U(|c - p|_1) * params with
    U(r) = r^2 log(r + 1e-6)"""
    theta = theta.to(coords).unsqueeze(1)
    out = (theta[:, :, :, :2] @ coords.unsqueeze(-1) + theta[:, :, :, 2:]).squeeze(-1)
    if control_points is not None:
        d = (coords.reshape(coords.shape[0], -1, 1, 2) - control_points.to(coords).reshape(1, 1, -1, 2)).abs().sum(-1)
        out = out + ((d ** 2) * torch.log(d + 1e-6) * control_params.to(coords)).sum(dim=2).reshape(theta.shape[0], coords.shape[1], 1)
    return out


def transform_frame(frame, theta, control_points=None, control_params=None):
    """model.py:42-46"""
    b, _, h, w = frame.shape
    grid = coordinate_grid(h, w, frame).reshape(1, h * w, 2)
    grid = warp_coordinates(grid, theta, control_points, control_params).reshape(b, h, w, 2)
    return F.grid_sample(frame, grid, padding_mode="reflection", align_corners=False)


def warp_jacobian(coords, theta, control_points=None, control_params=None):
    """model.py:70-75 (d warped / d coords per keypoint, rows = output x / y)"""
    coords = coords.detach().requires_grad_(True) if not coords.requires_grad else coords
    new = warp_coordinates(coords, theta, control_points, control_params)
    gx = torch.autograd.grad(new[..., 0].sum(), coords, create_graph=True)[0]
    gy = torch.autograd.grad(new[..., 1].sum(), coords, create_graph=True)[0]
    return torch.stack([gx, gy], dim=-2)


def equivariance(kp_d, transformed_kp, theta, control_points, control_params, w_value, w_jacobian):
    """model.py:231-246 given the encoder's output on the warped frame"""
    out = {"equivariance": w_value * torch.abs(kp_d["kp"] - warp_coordinates(transformed_kp["kp"], theta, control_points, control_params)).mean()}
    if w_jacobian != 0:
        jt = warp_jacobian(transformed_kp["kp"], theta, control_points, control_params) @ transformed_kp["jacobian"]
        value = torch.inverse(kp_d["jacobian"]) @ jt
        out["equivariance_jacobian"] = w_jacobian * torch.abs(torch.eye(2).view(1, 1, 2, 2).to(value) - value)
    return out


# --------------------------------------------------------------------------------------------- background motion predictor
def _rn_conv_bn(x, P, conv, bn, train, stride, relu, update=True):
    w = P[conv + ".weight"].to(x.dtype)
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = batchnorm(y, P, bn, train, update_stats=train and update)
    if train and update and (bn + ".num_batches_tracked") in P:
        P[bn + ".num_batches_tracked"] += 1
    return F.relu(y) if relu else y


def resnet18_pooled(x, P, pfx, train):
    """torchvision resnet18 up to the global average pool (resnet.py _forward_impl): 7x7/2 conv + BN + ReLU, 3x3/2 max-pool (pad 1),
    layer1..4 = 2 BasicBlocks each (the first of layers 2-4 with stride 2 and a 1x1/2 conv + BN projection of the skip path)"""
    y = _rn_conv_bn(x, P, pfx + "conv1", pfx + "bn1", train, 2, True)
    y = F.max_pool2d(y, 3, 2, 1)
    for layer in range(1, 5):
        for blk in range(2):
            b = f"{pfx}layer{layer}.{blk}."
            stride = 2 if (layer > 1 and blk == 0) else 1
            skip = y
            if (b + "downsample.0.weight") in P:
                skip = _rn_conv_bn(y, P, b + "downsample.0", b + "downsample.1", train, stride, False)
            z = _rn_conv_bn(y, P, b + "conv1", b + "bn1", train, stride, True)
            z = _rn_conv_bn(z, P, b + "conv2", b + "bn2", train, 1, False)
            y = F.relu(z + skip)
    return y.mean(dim=(2, 3))


def bg_motion_predictor(source, driving, P, pfx="", train=False):
    """BGMotionPredictor.forward, bg_motion_predictor.py:18-24 -> (B,3,3), third row [0 0 1]"""
    feat = resnet18_pooled(torch.cat([source, driving], dim=1), P, pfx + "bg_encoder.", train)
    pred = F.linear(feat, P[pfx + "bg_encoder.fc.weight"].to(feat.dtype), P[pfx + "bg_encoder.fc.bias"].to(feat.dtype))
    b = source.shape[0]
    return torch.cat([pred.view(b, 2, 3), torch.eye(3, dtype=feat.dtype)[2:3].expand(b, 1, 3)], dim=1)


def bg_loss(bg_param, bg_param_reverse):
    """model.py:248-253"""
    value = bg_param @ bg_param_reverse
    return 10 * torch.abs(torch.eye(3).view(1, 3, 3).to(value) - value).mean()
