#!/usr/bin/env python
"""Generates tests/golden/jpeg_frames.npz: a few JPEG streams (encoded by Pillow here) with the pixels Pillow's own
decoder -- `Image.open(f).convert('RGB')`, the call of /root/reference/reid/data/video_loader.py:124-141 -- returns for
them.  The fixture pins oracle/ref_c/jpeg_baseline.c (CPU suite) and grl_jpeg_decode_batch (GPU suite) to Pillow's
output independently of the Pillow build on the machine that runs the tests.

  python tests/golden/make_jpeg_golden.py
"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def frame(h, w, seed, grey=False):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 90 * np.sin(xx / rng.uniform(5, 40) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(6, 60) + rng.uniform(0, 6))
                     for _ in range(3)], -1)
    tex = rng.normal(0, 14, ((h + 3) // 4, (w + 3) // 4, 3)).repeat(4, 0).repeat(4, 1)[:h, :w] + rng.normal(0, 5, (h, w, 3))
    img = np.clip(base + tex, 0, 255).astype(np.uint8)
    return img[..., 0] if grey else img


CASES = [   # (name, h, w, save kwargs, grey)
    ('mars_420_q90', 256, 128, dict(quality=90), False),
    ('mars_420_q60_opt', 256, 128, dict(quality=60, optimize=True), False),
    ('small_444_q95', 64, 40, dict(quality=95, subsampling=0), False),
    ('odd_422_q80', 33, 17, dict(quality=80, subsampling=1), False),
    ('odd_420_q75_rst', 100, 77, dict(quality=75, subsampling=2, restart_marker_blocks=3), False),
    ('grey_q85', 48, 32, dict(quality=85), True),
]


def main():
    out = {}
    for i, (name, h, w, kw, grey) in enumerate(CASES):
        buf = io.BytesIO()
        Image.fromarray(frame(h, w, 100 + i, grey)).save(buf, format='JPEG', **kw)
        data = buf.getvalue()
        out['jpeg.' + name] = np.frombuffer(data, np.uint8)
        out['rgb.' + name] = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
    np.savez_compressed(os.path.join(HERE, 'jpeg_frames.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
