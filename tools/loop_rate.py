#!/usr/bin/env python
"""The product's training loop as a user runs it -- SEQTrainer.train(epoch, loader, optimizer): DevicePrefetcher,
_parse_data, the 5-term step, the loss / precision meters of trainer.py:63-97 (read one step late), gc.freeze() after the
first step -- fed three ways: float clips from the host (what the reference's loader yields: 50 MB per 32 x 4 batch), raw uint8
clips + augmentation draws (device augmentation), JPEG bytes + draws (device decode + augmentation).
ms per iteration = (time of 70 iterations - time of 10) / 60.

`--mode eval`: ATTEvaluator.extract_feature(loader) -- the evaluator's own loop -- over 60 batches, clip-features/s.

  python tools/loop_rate.py [--mode train|eval] [--math bf16s] [--clips 32] [--seq-len 4]
"""
import argparse
import contextlib
import io
import json
import os
import random
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


class Batches(object):
    """a loader: n times through a short list of prepared batches (the loader workers' output)"""

    def __init__(self, rows, n):
        self.rows, self.n = rows, n

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.rows[i % len(self.rows)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='train', choices=['train', 'eval'])
    ap.add_argument('--math', default='bf16s')
    ap.add_argument('--seq-len', type=int, default=4)
    ap.add_argument('--dense', type=int, default=0, help='--mode eval: dense mode (test_all.py), this many clips per tracklet, one tracklet per batch')
    ap.add_argument('--clips', type=int, default=32)
    a = ap.parse_args()
    import decode_rate
    from grl_amd import train_engine
    from grl_amd.reid import models
    from grl_amd.reid.data.augment import draw_clip_params
    from grl_amd.reid.data.jpeg import JpegBatch
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.synthetic import synth_clips, synth_state_dict
    dev = torch.device('cuda', 0)
    B, T = a.clips, a.seq_len
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0))
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
    mods = [m.to(dev).train() for m in (cnn, siam, siamv)]
    tr = SEQTrainer(mods[0], mods[1], mods[2], PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                    OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
    opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
    train_engine.set_math(a.math)
    rnd = random.Random(3)
    pids = (torch.arange(B) // 2 * 7) % 625
    draws = [torch.tensor([draw_clip_params(T, 256, 128, rnd) for _ in range(B)], dtype=torch.int32) for _ in range(4)]
    frames = decode_rate.make_frames(B * T)
    # tensor batches are PINNED, as the reference's DataLoader(pin_memory=True) hands them over (dataloader.py:60-79)
    floats = [synth_clips(B, T, seed=s).pin_memory() for s in range(4)]
    u8 = [(f * 40 + 128).clamp_(0, 255).to(torch.uint8).pin_memory() for f in floats]
    unpinned = [(f * 40 + 128).clamp_(0, 255).to(torch.uint8) for f in floats]
    feeds = {
        "float clips from the host (the reference's loader output)": [(f, pids, pids) for f in floats],
        "uint8 clips + draws (device augmentation)": [(u, pids, pids, d) for u, d in zip(u8, draws)],
        "uint8 clips + draws, PAGEABLE batches (a loader without pin_memory)": [(u, pids, pids, d) for u, d in zip(unpinned, draws)],
        "jpeg bytes + draws (device decode + augmentation)": [(JpegBatch(frames, (B, T)), pids, pids, d) for d in draws],
    }
    if a.mode == 'eval':
        from grl_amd import engine
        from grl_amd.reid.evaluator import ATTEvaluator
        engine.set_math(a.math)
        ev = ATTEvaluator(mods[0].eval(), mods[1].eval(), False)
        out = {"mode": "eval", "math": a.math, "clips": B, "frames_per_clip": T, "loop": "ATTEvaluator.extract_feature"}
        feeds = {k: [r[:3] for r in rows] for k, rows in feeds.items()}
        per_batch = B
        if a.dense:
            # one tracklet of `dense` clips per loader item, as the reference's dense loaders (batch size 1)
            ev = ATTEvaluator(mods[0], mods[1], True)
            per_batch = a.dense
            out["loop"] += " (dense mode: %d clips per tracklet, clip features averaged per tracklet)" % a.dense
            d_float = [synth_clips(a.dense, T, seed=s)[None].pin_memory() for s in range(4)]
            d_u8 = [(f * 40 + 128).clamp_(0, 255).to(torch.uint8).pin_memory() for f in d_float]
            d_frames = decode_rate.make_frames(a.dense * T)
            feeds = {"float tracklets (pinned)": [(f, pids[:1], pids[:1]) for f in d_float],
                     "uint8 tracklets (pinned)": [(u, pids[:1], pids[:1]) for u in d_u8],
                     "jpeg bytes (device decode)": [(JpegBatch(d_frames, (1, a.dense, T)), pids[:1], pids[:1])]}
        for name, rows in feeds.items():
            def run(n):
                t0 = time.perf_counter()
                f, _, _ = ev.extract_feature(Batches(rows, n))
                torch.cuda.synchronize()
                assert f.shape == (n if a.dense else n * B, 6144)
                return time.perf_counter() - t0
            run(10)
            t10, t70 = run(10), run(70)
            out[name.replace(' + draws', '').replace(' (device augmentation)', '').replace(' + augmentation)', ')') + ": clip-features/s"] = round(60 * per_batch / (t70 - t10), 1)
        print(json.dumps(out))
        return
    out = {"mode": "train", "math": a.math, "clips": B, "frames_per_clip": T, "loop": "SEQTrainer.train (meters read one step late; GRL_LAZY_METERS=0: loss.item() per step as upstream)"}
    for name, rows in feeds.items():
        def run(n):
            with contextlib.redirect_stdout(io.StringIO()):
                t0 = time.perf_counter()
                tr.train(0, Batches(rows, n), opt)
                torch.cuda.synchronize()
                return time.perf_counter() - t0
        run(10)
        t10, t70 = run(10), run(70)
        out[name + ": ms/iteration"] = round((t70 - t10) / 60 * 1e3, 2)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
