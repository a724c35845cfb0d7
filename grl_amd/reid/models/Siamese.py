"""Siamese temporal-attention pooling + pair verification head.

Schema and call surface of /root/reference/reid/models/Siamese.py:42-142:
``featQ/featK/featV`` (+``_bn``), ``classifierBN``, ``classifierlinear``;
``self_attention(x[b,T,D]) -> [b,D]`` (:79-106) and
``forward(x[B,T,D]) -> (cls[B/2,B/2,2], out[B,D])`` (:108-142).
``featV*`` is never used by the reference forward either; it exists so
checkpoints load.  Compute is issued by grl_amd.engine on HIP.
"""
import torch
from torch import nn

__all__ = ['Siamese']


def _init_kaiming(m):
    # Siamese.py:17-29
    if isinstance(m, nn.Linear):
        nn.init.kaiming_uniform_(m.weight, mode='fan_out')
        nn.init.constant_(m.bias, 0.0)
    elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)) and m.affine:
        nn.init.constant_(m.weight, 1.0)
        nn.init.constant_(m.bias, 0.0)


def _init_classifier(m):
    # Siamese.py:32-38
    if isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, std=0.001)
        nn.init.constant_(m.bias, 0.0)


class Siamese(nn.Module):
    def __init__(self, input_num, output_num, class_num):
        super().__init__()
        self.input_num = input_num
        self.output_num = output_num
        self.class_num = class_num
        self.feat_num = input_num
        for tag in ('Q', 'K', 'V'):
            lin = nn.Linear(input_num, output_num)
            bn = nn.BatchNorm1d(output_num)
            _init_kaiming(lin)
            _init_kaiming(bn)
            setattr(self, 'feat' + tag, lin)
            setattr(self, 'feat%s_bn' % tag, bn)
        self.softmax = nn.Softmax(dim=-1)
        self.classifierBN = nn.BatchNorm1d(self.feat_num)
        self.classifierlinear = nn.Linear(self.feat_num, self.class_num)
        _init_kaiming(self.classifierBN)
        _init_classifier(self.classifierlinear)

    def self_attention(self, input):
        from grl_amd import engine
        return engine.siamese_self_attention(self, input)

    def forward(self, x):
        if x.size(0) % 2 != 0:
            raise RuntimeError("the batch size should be even number!")
        from grl_amd import engine
        return engine.siamese_forward(self, x)
