"""BGMotionPredictor: a resnet18 over concat(source, driving) regressing the 2x3 affine background motion.
reference: modules/bg_motion_predictor.py:5-24 (torchvision.models.resnet18 with a 6-channel stem and a 6-way fc, fc initialised to
the identity transform); the torchvision architecture (resnet.py: 7x7/2 stem, BatchNorm, 3x3/2 max-pool, four stages of two
BasicBlocks with 64/128/256/512 channels, the first block of stages 2-4 strided with a 1x1/2 projection, global average pool, fc)
is restated with the same module names, so `bg_predictor.bg_encoder.*` checkpoints load.

Execution: one engine program -- the stem and the eight BasicBlocks on the conv / BatchNorm kernels the HRNet stem uses
(residual add + ReLU inside the BatchNorm pass, stride 2 = stride 1 sub-sampled), the 3x3/2 max-pool of K22, and a torch island
for the (B,8,8,512) global average + the 512 -> 6 linear layer."""
from __future__ import annotations

import torch
from torch import nn

from ..engine import Ctx, run_program
from .transformer.hr_base import BasicBlock, _conv_bn


class _ResNet18(nn.Module):
    """parameter container with torchvision's resnet18 attribute names"""

    def __init__(self, in_channels=3, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        for k, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            down = None
            if stride != 1 or inplanes != planes:
                down = nn.Sequential(nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(planes))
            setattr(self, f"layer{k}", nn.Sequential(_Block(inplanes, planes, stride, down), _Block(planes, planes)))
            inplanes = planes
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)


class _Block(BasicBlock):
    """torchvision's BasicBlock = the HRNet one with default BatchNorm momentum (both 0.1)"""


class BGMotionPredictor(nn.Module):
    def __init__(self):
        super().__init__()
        self.bg_encoder = _ResNet18()
        self.bg_encoder.conv1 = nn.Conv2d(6, 64, kernel_size=(7, 7), stride=(2, 2), padding=(3, 3), bias=False)
        self.bg_encoder.fc = nn.Linear(512, 6)
        self.bg_encoder.fc.weight.data.zero_()
        self.bg_encoder.fc.bias.data.copy_(torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float))

    def _program(self, e: Ctx, source_image, driving_image):
        enc = self.bg_encoder
        x = e.from_nchw(torch.cat([source_image, driving_image], dim=1))
        y = _conv_bn(e, x, enc.conv1, enc.bn1, relu=True, need_dx=False)
        y = e.maxpool3s2(y)
        for k in range(1, 5):
            for blk in getattr(enc, f"layer{k}"):
                y = blk.run(e, y)
        bs = source_image.shape[0]

        def head(feat, w, b):
            pred = torch.nn.functional.linear(feat.mean(dim=(1, 2)), w, b)                   # global average pool + fc
            eye = torch.eye(3, device=feat.device, dtype=feat.dtype)[2:3].expand(bs, 1, 3)
            return [torch.cat([pred.view(bs, 2, 3), eye], dim=1)]                              # third row [0 0 1]  (:20-23)
        (out,) = e.island(head, [y, enc.fc.weight, enc.fc.bias])
        return (out.t,), (out.add_grad,), (None, None)

    def forward(self, source_image, driving_image):
        return run_program(self, self._program, [source_image, driving_image])[0]
