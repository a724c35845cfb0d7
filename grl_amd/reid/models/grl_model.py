"""ResNet50 + GCE + Temporal Reciprocal Learning model (drop-in surface).

Same constructor, attribute names and state_dict keys as the reference's
``ResNet50_GRL_Model`` (/root/reference/reid/models/grl_model.py:184-231),
``TRLBlock`` (:88-128) and its all-1x1 ``BasicBlock`` (:51-64).  The forward
(:211-228, TRL loop :131-180) is executed on MI355X by grl_amd.engine; there
is no torch-op fallback: a non-HIP input raises.
"""
import torch
from torch import nn
from torch.nn import init

from .basebranch import Backbone
from .resnets1 import _Holder

__all__ = ['resnet50_grl', 'ResNet50_GRL_Model', 'TRLBlock', 'BasicBlock']


class BasicBlock(_Holder):
    """Memo update block: three 1x1 convs (+BN) on ``x1 + x2`` with a
    residual (grl_model.py:51-85; the 3x3 is commented out upstream)."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)


def _biased_conv_relu():
    return nn.Sequential(nn.Conv2d(2048, 2048, 1, 1), nn.ReLU())


def _channel_mlp():
    return nn.Sequential(nn.Linear(2048, 128, bias=False), nn.ReLU(inplace=True),
                         nn.Linear(128, 2048, bias=False), nn.Sigmoid())


class TRLBlock(_Holder):
    """Bidirectional temporal reciprocal learning (grl_model.py:88-180).
    Note the upstream spelling ``channel_atte_foreward_corr`` is part of the
    checkpoint schema."""

    def __init__(self, feat_num):
        super().__init__()
        self.feat_num = feat_num
        self.uncorr_memo_forward = BasicBlock(2048, 512)
        self.forward_f1 = _biased_conv_relu()
        self.forward_f2 = _biased_conv_relu()
        self.channel_atte_foreward_corr = _channel_mlp()
        self.uncorr_memo_backward = BasicBlock(2048, 512)
        self.backward_f1 = _biased_conv_relu()
        self.backward_f2 = _biased_conv_relu()
        self.channel_atte_backward_corr = _channel_mlp()


class ResNet50_GRL_Model(nn.Module):
    def __init__(self, num_feat=2048, num_features=512, height=256, width=128,
                 pretrained=True, dropout=0, numclasses=0):
        super().__init__()
        self.pretrained = pretrained
        self.num_feat = num_feat
        self.dropout = dropout
        self.num_classes = numclasses
        self.output_dim = num_features
        print('Num of features: {}.'.format(self.num_feat))
        self.backbone = Backbone(height=height, width=width, pretrained=pretrained)
        self.temporal_learning_block = TRLBlock(2048)
        self.corr_bn = nn.BatchNorm1d(2048)
        init.constant_(self.corr_bn.weight, 1)
        init.constant_(self.corr_bn.bias, 0)
        self.uncorr_bn = nn.BatchNorm1d(2048)
        init.constant_(self.uncorr_bn.weight, 1)
        init.constant_(self.uncorr_bn.bias, 0)

    def forward(self, inputs, training=True):
        """inputs [B,T,3,256,128] on a HIP device -- fp32 (normalised by the loader, as upstream)
        or raw uint8 pixels (normalised on the device) ->
        (x_uncorr [B,2048], x_corr [B,T,2048])  (grl_model.py:211-228)."""
        from grl_amd import engine
        return engine.grl_forward(self, inputs)


def resnet50_grl(*args, **kwargs):
    return ResNet50_GRL_Model(*args, **kwargs)
