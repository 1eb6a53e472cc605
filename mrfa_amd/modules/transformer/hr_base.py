"""HRNet-W32 stem of the MTIA prior on the HIP engine.  reference: modules/transformer/hr_base.py (BasicBlock 26-54,
Bottleneck 57-95, HighResolutionModule 120-289, HRNET_base 294-450).

The nn.Module tree only HOLDS parameters, in exactly the reference's nesting (conv1/bn1, layer1.<i>.conv<k>,
transition<k>.<i>.<j>.<0|1>, stage<k>.<m>.branches.<b>.<i>.*, stage<k>.<m>.fuse_layers.<i>.<j>.*), so the reference's
state_dict loads; the computation is `run(e, ...)` on NHWC Views:
  * conv -> BatchNorm(+ReLU) pairs: the conv epilogue accumulates the batch statistics, one bn_act pass applies them;
  * the residual add + ReLU closing every block is folded into that bn_act pass (mrfa_bn_act_fwd `res`);
  * stride-2 3x3 convs are the stride-1 conv sub-sampled at even pixels (mrfa_subsample_*);
  * the branch fusion (sum of identity / down-sampled / nearest-up-sampled branches, ReLU) uses mrfa_upsample_add_act_*.
"""
from __future__ import annotations

from typing import List

from torch import nn

from ...engine import Ctx, View
from ..util import run_block

BN_MOMENTUM = 0.1


def _conv_bn(e: Ctx, x: View, conv, bn, relu: bool, res: View = None, need_dx=True, out_sole=False) -> View:
    """out_sole: the caller feeds the result to ONE convolution and nothing else (the first BatchNorm of a residual block): the first phase of this
    BatchNorm's backward then rides in that convolution's data gradient (engine.Ctx.bn_act)"""
    raw, st = e.conv_bn_raw(x, conv, bn, need_dx=need_dx)
    return e.bn_act(raw, bn, st, relu=relu, res=res, sole_consumer=True, out_sole=out_sole)


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    """two 3x3 conv+BN, residual, ReLU.  reference: hr_base.py:26-54"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def run(self, e: Ctx, x: View) -> View:
        residual = x if self.downsample is None else _conv_bn(e, x, self.downsample[0], self.downsample[1], relu=False)
        raw1, st1 = e.conv_bn_raw(x, self.conv1, self.bn1)
        if e.prologue_ok(raw1, self.conv2):
            # relu(bn1(.)) rides in conv2's prologue -- forward, data gradient and weight gradient read conv1's raw output: three launches per block
            # instead of four, and one 8 MB tensor less to write and read (csrc/conv_lean.hip, csrc/wgrad_lean.hip)
            pre = e.prebn(raw1, self.bn1, st1)
            raw2, st2 = e.conv_bn_raw(raw1, self.conv2, self.bn2, pre=pre)
            return e.bn_act(raw2, self.bn2, st2, relu=True, res=residual, sole_consumer=True)
        y = e.bn_act(raw1, self.bn1, st1, relu=True, sole_consumer=True, out_sole=True)
        return _conv_bn(e, y, self.conv2, self.bn2, relu=True, res=residual)

    def forward(self, x):
        return run_block(self, x)

    @staticmethod
    def run_lockstep(e: Ctx, blocks: List["BasicBlock"], xs: List[View]) -> List[View]:
        """the same block position of independent branches under SyncBatchNorm (reference train.py:43): all first convolutions, ONE statistics
        collective, all BatchNorm applications; likewise the second half -- two collectives per depth and direction instead of two per block
        (engine.Ctx.sync_stats).  The arithmetic of every layer is that of run()."""
        raws = [e.conv_bn_raw(x, b.conv1, b.bn1) for b, x in zip(blocks, xs)]
        e.sync_stats([(b.bn1, st) for b, (_, st) in zip(blocks, raws)])
        ys = [e.bn_act(raw, b.bn1, st, relu=True, sole_consumer=True, out_sole=True) for b, (raw, st) in zip(blocks, raws)]
        raws = [e.conv_bn_raw(y, b.conv2, b.bn2) for b, y in zip(blocks, ys)]
        e.sync_stats([(b.bn2, st) for b, (_, st) in zip(blocks, raws)])
        return [e.bn_act(raw, b.bn2, st, relu=True, res=x, sole_consumer=True) for b, x, (raw, st) in zip(blocks, xs, raws)]


class Bottleneck(nn.Module):
    """1x1 -> 3x3 -> 1x1 (x4 channels) conv+BN, residual, ReLU.  reference: hr_base.py:57-95"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def run(self, e: Ctx, x: View) -> View:
        residual = x if self.downsample is None else _conv_bn(e, x, self.downsample[0], self.downsample[1], relu=False)
        y = _conv_bn(e, x, self.conv1, self.bn1, relu=True, out_sole=True)
        y = _conv_bn(e, y, self.conv2, self.bn2, relu=True, out_sole=True)
        return _conv_bn(e, y, self.conv3, self.bn3, relu=True, res=residual)

    def forward(self, x):
        return run_block(self, x)


blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}


def _residual_stack(block, inplanes, planes, blocks, stride=1):
    """`blocks` residual blocks, the first with a 1x1 conv+BN projection when the shape changes.
    reference: HRNET_base._make_layer hr_base.py:375-393 / HighResolutionModule._make_one_branch 152-193"""
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes * block.expansion, momentum=BN_MOMENTUM))
    layers = [block(inplanes, planes, stride, downsample)]
    layers += [block(planes * block.expansion, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """parallel residual branches at resolutions 1, 1/2, 1/4 ... followed by an all-to-all fusion.
    reference: hr_base.py:120-289"""

    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method, multi_scale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError(f'NUM_BRANCHES({num_branches}) <> NUM_BLOCKS({len(num_blocks)}) / NUM_CHANNELS({len(num_channels)}) / '
                             f'NUM_INCHANNELS({len(num_inchannels)})')
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        branches = []
        for i in range(num_branches):
            branches.append(_residual_stack(blocks, self.num_inchannels[i], num_channels[i], num_blocks[i]))
            self.num_inchannels[i] = num_channels[i] * blocks.expansion
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(True)

    def _make_fuse_layers(self):
        """fuse_layers[i][j] maps branch j to the resolution / width of branch i: 1x1 conv+BN+nearest upsample for
        coarser j, a chain of stride-2 3x3 conv+BN(+ReLU) for finer j.  reference: hr_base.py:203-261"""
        if self.num_branches == 1:
            return None
        nb, ch = self.num_branches, self.num_inchannels
        rows = []
        for i in range(nb if self.multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False), nn.BatchNorm2d(ch[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        cout = ch[i] if last else ch[j]
                        mods = [nn.Conv2d(ch[j], cout, 3, 2, 1, bias=False), nn.BatchNorm2d(cout)]
                        if not last:
                            mods.append(nn.ReLU(True))
                        steps.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def _term(self, e: Ctx, i: int, j: int, xj: View, res: View = None) -> View:
        """fuse_layers[i][j](x[j]) before the up-sampling (which ups_add fuses with the sum); for the down path the
        running sum `res` is added inside the last BatchNorm pass"""
        f = self.fuse_layers[i][j]
        if j > i:
            return _conv_bn(e, xj, f[0], f[1], relu=False)
        y = xj
        for k, step in enumerate(f):
            last = k == len(f) - 1
            y = _conv_bn(e, y, step[0], step[1], relu=not last, res=res if last else None)
        return y

    def _lockstep(self, e: Ctx) -> bool:
        """SyncBatchNorm with a statistics collective per layer: walk the branches side by side (BasicBlock.run_lockstep, _fuse_lockstep)"""
        blocks = [b for br in self.branches for b in br]
        return (e.syncbn_lockstep(blocks[0].bn1) and all(isinstance(b, BasicBlock) and b.downsample is None for b in blocks)
                and len({len(br) for br in self.branches}) == 1)

    def _fuse_lockstep(self, e: Ctx, x: List[View]) -> List[View]:
        """run()'s fuse sums with the FIRST convolution of every term issued up front and one statistics collective for all of them (the terms
        of a fuse layer read the branch outputs only): 2 collectives per module and direction instead of 7 (three branches)"""
        nb = self.num_branches
        first = {}
        for i in range(len(self.fuse_layers)):
            for j in range(nb):
                if j != i:
                    f = self.fuse_layers[i][j]
                    conv, bn = (f[0], f[1]) if j > i else (f[0][0], f[0][1])
                    first[(i, j)] = (bn,) + tuple(e.conv_bn_raw(x[j], conv, bn))
        e.sync_stats([(bn, st) for bn, _, st in first.values()])

        def term(i, j, res=None):
            bn, raw, st = first[(i, j)]
            if j > i:
                return e.bn_act(raw, bn, st, relu=False, sole_consumer=True)
            f = self.fuse_layers[i][j]
            y = None
            for k, step in enumerate(f):
                last = k == len(f) - 1
                if k == 0:
                    y = e.bn_act(raw, bn, st, relu=not last, res=res if last else None, sole_consumer=True)
                else:
                    y = _conv_bn(e, y, step[0], step[1], relu=not last, res=res if last else None)
            return y
        out = []
        for i in range(len(self.fuse_layers)):
            y = x[0] if i == 0 else term(i, 0)
            for j in range(1, nb):
                last = j == nb - 1
                if j == i:
                    y = e.ups_add(x[j], y, 1, relu=last)
                elif j > i:
                    y = e.ups_add(term(i, j), y, 2 ** (j - i), relu=last)
                else:
                    y = term(i, j, res=y)
            out.append(y)
        return out

    def run(self, e: Ctx, x: List[View]) -> List[View]:
        if self.num_branches == 1:
            y = x[0]
            for blk in self.branches[0]:
                y = blk.run(e, y)
            return [y]
        x = list(x)
        if self._lockstep(e):
            for d in range(len(self.branches[0])):
                x = BasicBlock.run_lockstep(e, [br[d] for br in self.branches], x)
            return self._fuse_lockstep(e, x)

        def chain(i):
            y = x[i]
            for blk in self.branches[i]:
                y = blk.run(e, y)
            return y
        # the resolution branches are independent chains of small latency-bound kernels: side by side on their own streams
        # while a hipGraph is captured (Ctx.lanes), in line otherwise
        lanes = e.lanes(self.num_branches - 1)
        e.fork(lanes)
        for i in range(1, self.num_branches):
            x[i] = e.branch(lanes[i - 1] if lanes else None, lambda i=i: chain(i))
        x[0] = chain(0)
        e.join(lanes)
        out = []
        for i in range(len(self.fuse_layers)):
            # y = sum_j fuse[i][j](x[j]) in the reference's order j = 0, 1, ..; ReLU closes the sum (hr_base.py:278-289)
            nb = self.num_branches
            y = x[0] if i == 0 else self._term(e, i, 0, x[0])
            for j in range(1, nb):
                last = j == nb - 1
                if j == i:
                    y = e.ups_add(x[j], y, 1, relu=last)
                elif j > i:
                    y = e.ups_add(self._term(e, i, j, x[j]), y, 2 ** (j - i), relu=last)
                else:
                    y = self._term(e, i, j, x[j], res=y)          # j < i <= nb-1: never the last term
            out.append(y)
        return out


class HRNET_base(nn.Module):
    """stem (two stride-2 3x3 convs) -> layer1 (4 Bottlenecks) -> stage2 (2 branches) -> stage3 (3 branches); returns the
    full-resolution branch (B, C0, H/4, W/4).  reference: hr_base.py:294-450"""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg['MODEL']['EXTRA']
        self.conv1 = nn.Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = _residual_stack(Bottleneck, 64, 64, 4)

        self.stage2_cfg = extra['STAGE2']
        block = blocks_dict[self.stage2_cfg['BLOCK']]
        num_channels = [c * block.expansion for c in self.stage2_cfg['NUM_CHANNELS']]
        self.transition1 = self._make_transition_layer([256], num_channels)
        self.stage2, pre_stage_channels = self._make_stage(self.stage2_cfg, num_channels)

        self.stage3_cfg = extra['STAGE3']
        block = blocks_dict[self.stage3_cfg['BLOCK']]
        num_channels = [c * block.expansion for c in self.stage3_cfg['NUM_CHANNELS']]
        self.transition2 = self._make_transition_layer(pre_stage_channels, num_channels)
        self.stage3, pre_stage_channels = self._make_stage(self.stage3_cfg, num_channels, multi_scale_output=False)
        self.pretrained_layers = extra['PRETRAINED_LAYERS']

    @staticmethod
    def _make_transition_layer(pre: List[int], cur: List[int]):
        """reference: hr_base.py:331-373"""
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                layers.append(None if c == pre[i] else
                              nn.Sequential(nn.Conv2d(pre[i], c, 3, 1, 1, bias=False), nn.BatchNorm2d(c), nn.ReLU(inplace=True)))
            else:
                steps = []
                for j in range(i + 1 - len(pre)):
                    cout = c if j == i - len(pre) else pre[-1]
                    steps.append(nn.Sequential(nn.Conv2d(pre[-1], cout, 3, 2, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    @staticmethod
    def _make_stage(layer_config, num_inchannels, multi_scale_output=True):
        """reference: hr_base.py:395-424"""
        n = layer_config['NUM_MODULES']
        block = blocks_dict[layer_config['BLOCK']]
        modules = []
        for i in range(n):
            mso = multi_scale_output or i != n - 1
            modules.append(HighResolutionModule(layer_config['NUM_BRANCHES'], block, layer_config['NUM_BLOCKS'], num_inchannels,
                                                layer_config['NUM_CHANNELS'], layer_config['FUSE_METHOD'], mso))
            num_inchannels = modules[-1].get_num_inchannels()
        return nn.Sequential(*modules), num_inchannels

    @staticmethod
    def _transition(e: Ctx, t, x: View) -> View:
        if isinstance(t[0], nn.Conv2d):                     # conv, BN, ReLU
            return _conv_bn(e, x, t[0], t[1], relu=True)
        for step in t:                                      # chain of stride-2 (conv, BN, ReLU)
            x = _conv_bn(e, x, step[0], step[1], relu=True)
        return x

    def run(self, e: Ctx, x: View) -> View:
        """x: (B,H,W,3) NHWC image view (no gradient is propagated into it)"""
        y = _conv_bn(e, x, self.conv1, self.bn1, relu=True, need_dx=False)
        y = _conv_bn(e, y, self.conv2, self.bn2, relu=True)
        e.mark("enc stem")
        for blk in self.layer1:
            y = blk.run(e, y)
        e.mark("enc layer1")
        xs = [y if t is None else self._transition(e, t, y) for t in self.transition1]
        for m in self.stage2:
            xs = m.run(e, xs)
        e.mark("enc stage2")
        nxt = []
        for i, t in enumerate(self.transition2):
            nxt.append(xs[i] if t is None else self._transition(e, t, xs[-1]))
        xs = nxt
        for k, m in enumerate(self.stage3):
            xs = m.run(e, xs)
            e.mark(f"enc stage3.{k}")
        return xs[0]

    def forward(self, x):
        def program(e: Ctx, xin):
            yv = self.run(e, e.from_nchw(xin))
            return (e.to_nchw(yv),), (lambda g: e.seed_grad_nchw(yv, g),), (None,)
        from ...engine import run_program
        return run_program(self, program, [x])[0]

    def init_weights(self, pretrained='', print_load_info=False):
        """reference: hr_base.py:452-478 (normal(std=0.001) convs, unit BatchNorms; `pretrained` checkpoints are not shipped)"""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if pretrained:
            import os
            import torch
            if not os.path.isfile(pretrained):
                raise ValueError(f'{pretrained} is not exist!')
            sd = torch.load(pretrained, map_location='cpu')
            own = self.state_dict()
            keep = {k: v for k, v in sd.items() if (k.split('.')[0] in self.pretrained_layers and k in own) or self.pretrained_layers[0] == '*'}
            self.load_state_dict(keep, strict=False)
