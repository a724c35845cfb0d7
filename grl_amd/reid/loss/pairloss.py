"""Pair verification loss (reference: reid/loss/pairloss.py:8-45): BCE between the
match probability of every (probe_i, gallery_j) pair and [pid_i == pid_j], plus the
top-1 precision of the (1-s, s) pseudo-logits.  The label mask is built on the device
(the reference round-trips it through a Python list, a host sync per step)."""
import torch
from torch import nn

from grl_amd.reid.evaluator.eva_functions import accuracy


class PairLoss(nn.Module):
    def __init__(self):
        super(PairLoss, self).__init__()
        self.BCE = nn.BCELoss()

    def forward(self, score, tar_probe, tar_gallery):
        n = score.size(0)
        mask = tar_probe.unsqueeze(0).expand(n, n).eq(tar_gallery.unsqueeze(1).expand(n, n)).view(-1)
        samplers = score.contiguous().view(-1)
        loss = self.BCE(samplers, mask.to(samplers.dtype))
        s = samplers.detach()
        prec, = accuracy(torch.stack((1 - s, s), 1), mask.long())
        return loss, prec
