"""Host side of the on-device training augmentation.

The reference augments every training clip on the host with PIL: RectScale -> RandomHorizontalFlip
-> RandomSizedEarser -> ToTensor -> Normalize (reid/data/dataloader.py:51-57,
reid/data/seqtransforms.py:92-190).  Here the loader ships the RAW uint8 clip plus a few integers
per clip -- the random decisions, drawn with the same calls in the same order as the reference's
transforms -- and `grl_augment_normalize_u8` (grl_amd/csrc/pointwise.hip) applies flip, erase,
ToTensor and Normalize in one pass on the device, bit-identical to the PIL pipeline on
256 x 128 frames (tests/golden/augment.npz).

Per-clip parameter block (int32): [flip, then per frame (erase, left, top, w, h, R, G, B)].
"""
import math
import random as _random

import numpy as np
import torch

PARAMS_PER_FRAME = 8


def draw_clip_params(T, H, W, rnd=_random, sl=0.02, sh=0.2, asratio=0.3, p=0.5):
    """The random decisions of RandomHorizontalFlip + RandomSizedEarser for one clip of T frames,
    consuming ``rnd`` (the `random` module or a random.Random) exactly as seqtransforms.py:140-151
    and :92-137 do: one random() for the flip; per frame one uniform() for the erase coin and, if
    it erases, rejection sampling of (area, aspect, x, y) followed by three randint() colours.

    Returns a list of 1 + 8*T ints.  The erased rectangle is where the REFERENCE puts it:
    `frame.paste(I, part1.size)` (seqtransforms.py:132) uses the patch's SIZE (w, h) as the
    upper-left corner, not (x1, y1), so the patch covers [w, 2w) x [h, 2h), clipped by the frame."""
    out = [1 if rnd.random() < 0.5 else 0]
    area = H * W
    for _ in range(T):
        if rnd.uniform(0.0, 1.0) > p:
            out += [0] * PARAMS_PER_FRAME
            continue
        while True:
            Se = rnd.uniform(sl, sh) * area
            re = rnd.uniform(asratio, 1 / asratio)
            He, We = np.sqrt(Se * re), np.sqrt(Se / re)
            xe, ye = rnd.uniform(0, W - We), rnd.uniform(0, H - He)
            if xe + We <= W and ye + He <= H and xe > 0 and ye > 0:
                x1, y1 = int(np.ceil(xe)), int(np.ceil(ye))
                w, h = int(np.floor(x1 + We)) - x1, int(np.floor(y1 + He)) - y1
                rgb = [rnd.randint(0, 255) for _ in range(3)]
                out += [1, w, h, w, h] + rgb            # pasted at (left, top) = (w, h): see docstring
                break
    return out


def pack_params(per_clip):
    """list of draw_clip_params() lists -> int32 tensor [n_clips, 1 + 8*T]."""
    return torch.tensor(per_clip, dtype=torch.int32)


def sample_frame_indices(num, seq_len, mode, np_random=np.random):
    """Frame indices of a tracklet of `num` frames (reid/data/video_loader.py:30-48,86-141):
    'rrs_train' one random frame per segment, 'rrs_test' the first of each segment, 'dense' all
    frames cut into clips of seq_len (the last one wraps around).  Returns an int array [seq_len]
    ('dense': [n_clips, seq_len])."""
    S = seq_len
    if num < S:
        strip = list(range(num)) + [num - 1] * (S - num)
        clip = np.array([[strip[s]] for s in range(S)])
    else:
        inter = math.ceil(num / S)
        strip = list(range(num)) + [num - 1] * (inter * S - num)
        clip = np.array([strip[inter * s:inter * (s + 1)] for s in range(S)])
    if mode == 'rrs_train':
        idx = np_random.choice(clip.shape[1], clip.shape[0])
        return clip[np.arange(len(clip)), idx]
    if mode == 'rrs_test':
        return clip[:, 0]
    if mode == 'dense':
        frames, out, cur = list(range(num)), [], 0
        while num - cur > S:
            out.append(frames[cur:cur + S])
            cur += S
        last = frames[cur:]
        for i in last:                       # iterates the list it is extending: the tail wraps until full
            if len(last) >= S:
                break
            last.append(i)
        out.append(last)
        return np.array(out)
    raise KeyError("Unknown sample method: {}".format(mode))


# ----------------------------------------------------------------------------
# RectScale on the device: PIL's BILINEAR resize, bit for bit
# ----------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2          # Pillow's fixed-point coefficient scale for 8-bit images


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's resampling table for one axis (libImaging/Resample.c: precompute_coeffs +
    normalize_coeffs_8bpc with the bilinear filter, support 1): for every output index the first
    input index, the tap count and the taps as 22-bit fixed-point integers.  Computed in float64
    with Pillow's own expression order, so the device kernel's integer arithmetic reproduces
    `frame.resize(size, Image.BILINEAR)` (RectScale, seqtransforms.py:30-47) exactly.
    Returns (bounds int32 [out, 2], coefs int32 [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coefs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            w = 1.0 - v if v < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            kv = k[x] / ww if ww != 0.0 else k[x]
            coefs[xx, x] = int(-0.5 + kv * (1 << _PRECISION_BITS)) if kv < 0 else int(0.5 + kv * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coefs
