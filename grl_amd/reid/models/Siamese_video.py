"""Pair verification head on already pooled clip features (uncorrelated
branch).  Schema / surface of /root/reference/reid/models/Siamese_video.py:
42-79 (``classifierBN``, ``classifierlinear``) and forward :129-184:
``forward(x[B,D]) -> (cls[B/2,B/2,2], out[B,D])`` with ``out`` reordered
probe-half then gallery-half.  Compute is issued by grl_amd.engine on HIP.
"""
from torch import nn

from .Siamese import _init_kaiming, _init_classifier

__all__ = ['Siamese_video']


class Siamese_video(nn.Module):
    def __init__(self, input_num=2048, output_num=2048, class_num=2):
        super().__init__()
        self.class_num = class_num
        self.feat_num = input_num
        self.classifierBN = nn.BatchNorm1d(self.feat_num)
        self.classifierlinear = nn.Linear(self.feat_num, self.class_num)
        _init_kaiming(self.classifierBN)
        _init_classifier(self.classifierlinear)
        self.muti_head = False

    def forward(self, x):
        from grl_amd import engine
        return engine.siamese_video_forward(self, x)
