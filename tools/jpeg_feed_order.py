#!/usr/bin/env python
"""Steps fed from compressed frames, and WHICH STREAMS TOUCH THE GPU FIRST (round 6 finding): HIP streams get their
hardware queue at first use; when the prefetcher's HIGH-priority streams were used before the engine's side streams, every
normal-priority stream created afterwards shared what was left and the bf16-storage train step ran at 32 ms instead of
17.8 (all its streams serialised).  This tool times, in one process and in the order asked for:
  resident clips (first), resident clips (after a prefetcher exists), JPEG bytes through engine.DevicePrefetcher.

  python tools/jpeg_feed_order.py --mode train|eval [--math bf16s] [--clips 32] [--seq-len 4] [--first model|prefetch|streams]
  GRL_PREFETCH_PRIORITY=-1|0 selects the prefetch streams' priority (engine.DevicePrefetcher)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='train', choices=['train', 'eval'])
    ap.add_argument('--math', default='bf16s')
    ap.add_argument('--clips', type=int, default=32)
    ap.add_argument('--seq-len', type=int, default=4)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--first', default='model', choices=['model', 'prefetch', 'streams'],
                    help='what uses the GPU first: the model step (default), a prefetcher with a decode, or only two '
                         'streams of the prefetch priority')
    ap.add_argument('--gc', default='freeze', choices=['freeze', 'default', 'off'],
                    help="freeze (what SEQTrainer.train does after its first step): gc.freeze() after three steps; default: Python's collector as it comes; off: gc.disable()")
    ap.add_argument('--trace', action='store_true', help='print issue / wall time per block of 10 resident steps first')
    a = ap.parse_args()
    import decode_rate
    from grl_amd import engine, train_engine
    from grl_amd.reid import models
    from grl_amd.reid.data.augment import draw_clip_params
    from grl_amd.reid.data.jpeg import JpegBatch
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.synthetic import synth_clips, synth_state_dict
    dev = torch.device('cuda', 0)
    B, T = a.clips, a.seq_len
    train = a.mode == 'train'
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0))
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
    mods = [m.to(dev).train(train) for m in (cnn, siam, siamv)]
    frames = decode_rate.make_frames(B * T)
    jb = JpegBatch(frames, (B, T))
    pids = (torch.arange(B) // 2 * 7) % 625
    rnd = random.Random(3)
    if train:
        tr = SEQTrainer(mods[0], mods[1], mods[2], PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                        OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
        opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
        train_engine.set_math(a.math)

        # (the augmentation draws are a loader WORKER's job: drawn once here, not in the timed thread)
        drawn = [torch.tensor([draw_clip_params(T, 256, 128, rnd) for _ in range(B)], dtype=torch.int32) for _ in range(8)]

        def loader(k):
            for i in range(k):
                yield jb, pids, pids, drawn[i % 8]

        def step(batch):
            inputs, targets = batch
            loss = tr._forward(inputs, targets, 0, 0)[0]
            opt.zero_grad()
            loss.backward()
            opt.step()

        def parse(batch):
            return tr._parse_data(batch)
        resident = ([synth_clips(B, T, seed=0).to(dev)], pids.to(dev))
    else:
        engine.set_math(a.math)

        def loader(k):
            for _ in range(k):
                yield jb, pids, pids

        def step(batch):
            engine.extract_features(mods[0], mods[1], batch)

        def parse(batch):
            return batch[0]
        resident = (synth_clips(B, T, seed=0) * 40 + 128).clamp_(0, 255).to(torch.uint8).to(dev)

    out = {"mode": a.mode, "math": a.math, "clips": B, "frames_per_clip": T, "first on the GPU": a.first,
           "prefetch priority": int(os.environ.get('GRL_PREFETCH_PRIORITY', '0'))}

    def timed(name, fn):
        fn(a.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(a.steps)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / a.steps * 1e3, 2)
        out[name.replace(': ms/step', '') + ': host issue ms/step'] = round(t_issue / a.steps * 1e3, 2)

    def run_resident(k):
        for _ in range(k):
            step(resident)

    def run_jpeg(k):
        for batch in engine.DevicePrefetcher(loader(k), dev):
            step(parse(batch))

    keep = None
    if a.first == 'prefetch':
        keep = parse(next(iter(engine.DevicePrefetcher(loader(1), dev))))
    elif a.first == 'streams':
        keep = [torch.cuda.Stream(dev, priority=out["prefetch priority"]) for _ in range(2)]
        for st in keep:
            with torch.cuda.stream(st):
                torch.zeros(16, device=dev)
    import gc
    if a.gc == 'off':
        gc.disable()
    elif a.gc == 'freeze':
        run_resident(3)
        gc.collect()
        gc.freeze()
    out["gc"] = a.gc
    if a.trace:                            # per-10-step trace (shows the collector's pauses)
        for blk in range(8):
            t0 = time.perf_counter()
            run_resident(10)
            ti = time.perf_counter() - t0
            torch.cuda.synchronize()
            print('steps %d-%d: issue %.2f ms/step, wall %.2f' % (blk * 10, blk * 10 + 9, ti * 100, (time.perf_counter() - t0) * 100), flush=True)
    timed("resident, first timed: ms/step", run_resident)
    keep = parse(next(iter(engine.DevicePrefetcher(loader(1), dev))))
    timed("resident, a prefetcher exists: ms/step", run_resident)
    timed("jpeg bytes through the prefetcher: ms/step", run_jpeg)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
