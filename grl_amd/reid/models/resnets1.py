"""ResNet-50 trunk *parameter schema* (stride-1 layer4).

Mirrors the state_dict layout of the reference's ``resnet50_s1``
(/root/reference/reid/models/resnets1.py:57-136, :180-189): Bottleneck
[3,4,6,3], layer4 stride 1, ``conv{1,2,3}/bn{1,2,3}/downsample.{0,1}``
names, so that reference (and torchvision ImageNet) checkpoints load.

These modules are parameter holders only.  No ``forward`` runs torch conv
ops: the compute is issued by :mod:`grl_amd.engine` through the C-ABI HIP
library.  Calling ``forward`` on one of them raises.
"""
import math

import torch.nn as nn

__all__ = ['Bottleneck', 'ResNetTrunk', 'resnet50_s1']

_LAYERS = (3, 4, 6, 3)
_PLANES = (64, 128, 256, 512)
_STRIDES = (1, 2, 2, 1)           # layer4 keeps 16x8 (resnets1.py:109)


class _Holder(nn.Module):
    def forward(self, *a, **k):   # pragma: no cover - never a compute path
        raise RuntimeError(
            'grl_amd parameter holder: compute goes through grl_amd.engine '
            '(HIP), not nn.Module.forward')


class Bottleneck(_Holder):
    """1x1 -> 3x3(stride) -> 1x1(x4) + residual (resnets1.py:57-93)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride


def _make_layer(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(
            nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
            nn.BatchNorm2d(planes * 4))
    mods = [Bottleneck(inplanes, planes, stride, down)]
    for _ in range(1, blocks):
        mods.append(Bottleneck(planes * 4, planes))
    return nn.Sequential(*mods)


class ResNetTrunk(_Holder):
    """conv1/bn1/maxpool/layer1..4 with the reference's init
    (normal(0, sqrt(2/(k*k*cout))), BN gamma=1 beta=0; resnets1.py:113-119)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inpl = 64
        for i, (n, p, s) in enumerate(zip(_LAYERS, _PLANES, _STRIDES)):
            setattr(self, 'layer%d' % (i + 1), _make_layer(inpl, p, n, s))
            inpl = p * 4
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


def resnet50_s1(pretrained=True, state_dict=None):
    """The reference downloads ImageNet weights at construction
    (resnets1.py:186-188).  There is no network here: pass ``state_dict``
    (e.g. torch.load of resnet50-19c8e357.pth) or set GRL_RESNET50_PTH;
    otherwise the reference's random init is kept."""
    import os
    import torch
    net = ResNetTrunk()
    if state_dict is None and pretrained:
        path = os.environ.get('GRL_RESNET50_PTH')
        if path and os.path.isfile(path):
            state_dict = torch.load(path, map_location='cpu')
    if state_dict is not None:
        own = net.state_dict()
        net.load_state_dict({k: v for k, v in state_dict.items() if k in own}, strict=False)
    return net
