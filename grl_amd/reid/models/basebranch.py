"""Backbone = ResNet-50 trunk + Global-guided Correlation Estimation (GCE).

Parameter schema of /root/reference/reid/models/basebranch.py:21-50
(``base.{0,1,4..7}``, ``glo_fc.{0,1}``, ``corr_atte.{0,1,2,3,5,6}``).
The forward (basebranch.py:52-68) is executed by grl_amd.engine on HIP.
"""
import torch.nn as nn

from .resnets1 import resnet50_s1, _Holder


class Backbone(_Holder):
    def __init__(self, height=256, width=128, pretrained=True):
        super().__init__()
        if (height, width) != (256, 128):
            # the reference hard-codes the 16x8 map in expand() (basebranch.py:59)
            raise ValueError('GRL backbone is defined for 256x128 inputs only')
        trunk = resnet50_s1(pretrained=pretrained)
        self.base = nn.Sequential(
            trunk.conv1, trunk.bn1, nn.ReLU(), trunk.maxpool,
            trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4)
        self.glo_fc = nn.Sequential(
            nn.Linear(2048, 1024), nn.BatchNorm1d(1024), nn.ReLU())
        self.corr_atte = nn.Sequential(
            nn.Conv2d(2048 + 1024, 1024, 1, 1, bias=False), nn.BatchNorm2d(1024),
            nn.Conv2d(1024, 256, 1, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(),
            nn.Conv2d(256, 1, 1, 1, bias=False), nn.BatchNorm2d(1))
