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


class RawVideoDataset(Dataset):
    """Tracklets -> RAW uint8 clips for the on-device input pipeline (the counterpart of the
    reference's VideoDataset + T.Compose, reid/data/video_loader.py:18-155, dataloader.py:51-72):
    the worker only decodes; RectScale, flip, erase, ToTensor and Normalize run on the GPU.

    ``tracklets``: [(img_paths, pid, camid)] as the reference's dataset objects provide
    (reid/dataset/mars.py); ``sample``: 'rrs_train' | 'rrs_test' | 'dense' with the reference's frame
    selection (``augment.sample_frame_indices``).  Items: (uint8 [T,3,H,W], pid, camid) -- 'dense':
    [n_clips,T,3,H,W] -- plus, with ``augment=True`` (training), the int32 block of
    ``augment.draw_clip_params`` that SEQTrainer hands to grl_augment_normalize_u8.  MARS crops are all
    256 x 128; another uniform size is RectScale'd on the device; frames of DIFFERENT sizes (DukeMTMC-VideoReID):
    see ``host_rect_scale`` below."""

    def __init__(self, tracklets, seq_len=4, sample='rrs_train', augment=False, height=256, width=128, decode='host',
                 host_rect_scale=False):
        """``decode``: 'host' -- the worker decodes with Pillow (items carry uint8 tensors); 'device' -- the worker only
        reads the files, items carry the JPEG byte strings, the loader uses ``jpeg.jpeg_collate`` and
        engine.DevicePrefetcher decodes the batch on the GPU (grl_jpeg_decode_batch, bit-identical to Pillow).
        ``host_rect_scale``: datasets whose frames DIFFER in size (DukeMTMC-VideoReID) cannot be stacked raw: with
        decode='host' the worker then applies RectScale(height, width) itself (PIL BILINEAR, as the reference); with
        decode='device' nothing is needed -- the prefetcher decodes per size and resizes on the GPU, the same bits."""
        if decode not in ('host', 'device'):
            raise ValueError("decode must be 'host' or 'device'")
        self.tracklets, self.seq_len, self.sample = list(tracklets), seq_len, sample
        self.augment, self.height, self.width, self.decode = augment, height, width, decode
        self.host_rect_scale = host_rect_scale

    def __len__(self):
        return len(self.tracklets)

    def _decode(self, path):
        import numpy as np
        from PIL import Image
        with Image.open(path) as im:
            im = im.convert('RGB')
            if self.host_rect_scale and im.size != (self.width, self.height):
                im = im.resize((self.width, self.height), Image.BILINEAR)          # seqtransforms.py:30-47
            return torch.from_numpy(np.ascontiguousarray(np.asarray(im).transpose(2, 0, 1)))

    def __getitem__(self, index):
        from .augment import draw_clip_params, sample_frame_indices
        paths, pid, camid = self.tracklets[index]
        idx = sample_frame_indices(len(paths), self.seq_len, self.sample)
        if self.decode == 'device':
            from .jpeg import read_file
            if self.sample == 'dense':          # [n_clips][T] byte strings (test_all.py's mode: video_loader.py:86-123)
                item = ([[read_file(paths[int(i)]) for i in row] for row in idx], pid, camid)
            else:
                item = ([read_file(paths[int(i)]) for i in idx], pid, camid)
            if self.augment:
                item += (torch.tensor(draw_clip_params(self.seq_len, self.height, self.width), dtype=torch.int32),)
            return item
        if self.sample == 'dense':
            clip = torch.stack([torch.stack([self._decode(paths[int(i)]) for i in row]) for row in idx])
        else:
            clip = torch.stack([self._decode(paths[int(i)]) for i in idx])
        item = (clip, pid, camid)
        if self.augment:
            item += (torch.tensor(draw_clip_params(self.seq_len, self.height, self.width), dtype=torch.int32),)
        return item
