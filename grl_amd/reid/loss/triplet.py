"""Batch-hard soft-margin triplet loss (reference: reid/loss/triplet.py:16-90, the
`mode='id'`, `dis_func='eu'`, `batch_hard=True`, `margin='soft'` path the trainer uses:
trainer.py:12,141)."""
import torch
import torch.nn as nn


class TripletLoss(nn.Module):
    def __init__(self, margin=0, batch_hard=False, dim=2048):
        super(TripletLoss, self).__init__()
        self.batch_hard = batch_hard
        if isinstance(margin, float) or margin == 'soft':
            self.margin = margin
        else:
            raise NotImplementedError('The margin {} is not recognized in TripletLoss()'.format(margin))

    def forward(self, feat, id=None, pos_mask=None, neg_mask=None, mode='id', dis_func='eu', n_dis=0):
        if mode != 'id' or n_dis != 0 or not self.batch_hard:
            raise NotImplementedError('only the batch-hard id-mode path of the reference trainer is provided')
        if id is None:
            raise RuntimeError('foward is in id mode, please input id!')
        if dis_func == 'cdist':
            feat = feat / feat.norm(p=2, dim=1, keepdim=True)
        dist = self.cdist(feat, feat)
        same = torch.eq(id.unsqueeze(1), id.unsqueeze(0))
        eye = torch.eye(feat.size(0), dtype=torch.bool, device=feat.device)
        max_positive = (dist * (same ^ eye).float()).max(1)[0]
        min_negative = (dist + 1e5 * same.float()).min(1)[0]
        z = max_positive - min_negative
        if isinstance(self.margin, float):
            return torch.clamp(z + self.margin, min=0)
        return torch.log(1 + torch.exp(z))

    def cdist(self, a, b):
        diff = a.unsqueeze(1) - b.unsqueeze(0)
        return ((diff ** 2).sum(2) + 1e-12).sqrt()


class TripletLoss_OIM(TripletLoss):
    """Instantiated by the reference trainer (trainer.py:11) but never called."""
