"""Input pipeline of the reference (reid/data, reid/dataset) is outside the hot path
(SURVEY.md section 2, rows 17-24).  ``get_data`` keeps the reference's signature and
return tuple and serves synthetic MARS-shaped clips arranged as (anchor, positive)
pairs -- the invariant ``Siamese.forward`` relies on (sampler.py:104-123)."""
import torch
from torch.utils.data import DataLoader, Dataset

from grl_amd.synthetic import synth_clips


class SyntheticPairs(Dataset):
    """`n_pairs` x 2 clips; both clips of a pair share the pid, cameras differ."""

    def __init__(self, n_pairs, seq_len, num_classes=625, seed=0, raw=False, augment=False):
        self.n, self.t, self.k, self.seed = 2 * n_pairs, seq_len, num_classes, seed
        self.raw = raw or augment   # uint8 pixels (normalised on the device) instead of float32
        self.augment = augment      # also yield the flip / erase decisions: the device applies them

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        clip = synth_clips(1, self.t, seed=self.seed * 100003 + i, raw=self.raw)[0]
        item = (clip, (i // 2 * 7919 + self.seed) % self.k, i % 2)
        if self.augment:
            from .augment import draw_clip_params
            item += (torch.tensor(draw_clip_params(self.t, clip.shape[-2], clip.shape[-1]), dtype=torch.int32),)
        return item


def get_data(dataset_name, split_id, data_dir, batch_size, seq_len, seq_srd, workers, only_eval=False):
    if dataset_name != 'synthetic':
        raise NotImplementedError(
            "dataset '%s': the MARS/DukeMTMC parsers and PIL transforms of the reference are outside "
            "this build's scope (no dataset is available here); use dataset 'synthetic' or feed "
            "SEQTrainer/ATTEvaluator any loader that yields (imgs[B,T,3,256,128], pids, camids)" % dataset_name)
    train = SyntheticPairs(8 * batch_size, seq_len)
    loader = DataLoader(train, batch_size=batch_size, shuffle=False, drop_last=True, num_workers=0)
    q = DataLoader(SyntheticPairs(15, seq_len, seed=1), batch_size=30, num_workers=0)
    g = DataLoader(SyntheticPairs(60, seq_len, seed=2), batch_size=30, num_workers=0)
    return train, 625, loader, q, g
