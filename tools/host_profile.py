#!/usr/bin/env python
"""Where the HOST time of a train step goes (round 6: at B x T = 32 x 4 the bf16-storage step takes as long as the host
needs to issue it -- bench.py's `host.launch_bound_frac` 0.99; at 4 clips per GPU every train step is host-bound).
cProfile over a few steps of the SEQTrainer step at a batch small enough that the GPU never back-pressures the host.

  python tools/host_profile.py [--math bf16s] [--clips 4] [--steps 6] [--top 45]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--math', default='bf16s')
    ap.add_argument('--clips', type=int, default=4)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--top', type=int, default=45)
    ap.add_argument('--sort', default='tottime')
    a = ap.parse_args()
    import contextlib
    from grl_amd.reid import models
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.synthetic import synth_clips, synth_state_dict
    from grl_amd import train_engine
    dev = torch.device('cuda', 0)
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0))
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
    cnn, siam, siamv = cnn.to(dev).train(), siam.to(dev).train(), siamv.to(dev).train()
    tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                    OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
    params = tr._all_params()
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
    clips = synth_clips(a.clips, 4, seed=0).to(dev)
    pids = (torch.arange(a.clips, device=dev) // 2 * 7) % 625

    def step():
        loss, _, _, _ = tr._forward([clips], pids, 0, 0)
        opt.zero_grad()
        loss.backward()
        opt.step()

    train_engine.set_math(a.math)
    for _ in range(4):
        step()
    from grl_amd.reid.train.trainer import _freeze_collector_once
    _freeze_collector_once()                 # as the product's loop does after its first step (the collector's pauses are step time)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('unprofiled: issue %.2f ms / step, wall %.2f ms / step' % (t_issue / a.steps * 1e3, t_all / a.steps * 1e3))
    # the backward closures run on the autograd engine's device thread, which a profiler enabled here does not see:
    # Tape.backward is wrapped so that it profiles itself on whatever thread calls it
    bw = cProfile.Profile()
    orig_bw = train_engine.Tape.backward

    def profiled_backward(self):
        bw.enable()
        try:
            return orig_bw(self)
        finally:
            bw.disable()
    train_engine.Tape.backward = profiled_backward
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    train_engine.Tape.backward = orig_bw
    s = io.StringIO()
    pstats.Stats(bw, stream=s).strip_dirs().sort_stats('tottime').print_stats(a.top)
    print('==== Tape.backward (autograd thread), %d steps ====' % a.steps)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(bw, stream=s).strip_dirs().sort_stats('cumulative').print_stats(40)
    print(s.getvalue())
    print('==== main thread ====')
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).strip_dirs().sort_stats(a.sort)
    st.print_stats(a.top)
    print('(%d steps profiled; divide by that for per-step figures)' % a.steps)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats('cumulative').print_stats(35)
    print(s.getvalue())


if __name__ == '__main__':
    main()
