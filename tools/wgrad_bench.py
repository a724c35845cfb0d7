#!/usr/bin/env python
"""Per-shape timing of grl_conv_wgrad_f32 on the weight-gradient GEMMs of one train step
(B x T = 32 x 4).  The shapes are recorded from a real SEQTrainer-style forward/backward, then
each unique shape is timed alone with the library's split choice and with forced splits
(GRL_WGRAD_SPLITS).   python tools/wgrad_bench.py [split ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from grl_amd import train_engine as TE
from grl_amd.synthetic import synth_clips, synth_state_dict


def record_shapes(b=32, t=4):
    import contextlib, io
    from grl_amd.reid import models
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0))
    cnn = cnn.cuda().train()
    seen = collections.OrderedDict()
    orig = TE.wgrad

    def spy(dz, x, dw, M, N, K, ldz=None, ldx=None, conv=None, k_out=0, accumulate=1, math=None):
        key = (M, N, K, ldz or N, ldx or K, conv, k_out, dz.dtype == torch.bfloat16)
        seen[key] = seen.get(key, 0) + 1
        return orig(dz, x, dw, M, N, K, ldz=ldz, ldx=ldx, conv=conv, k_out=k_out, accumulate=accumulate, math=math)
    TE.wgrad = spy
    old = TE.set_math(MATH)
    try:
        xu, xc = cnn(synth_clips(b, t, seed=0).cuda())
        (xu.sum() + xc.sum()).backward()
    finally:
        TE.set_math(old)
    torch.cuda.synchronize()
    TE.wgrad = orig
    return seen


def time_shape(key, iters=4):
    M, N, K, ldz, ldx, conv, k_out, b16 = key
    dev = torch.device('cuda:0')
    dt = torch.bfloat16 if b16 else torch.float32
    dz = torch.randn(M, ldz, device=dev).to(dt)
    if conv is None:
        x = torch.randn(M, ldx, device=dev).to(dt)
    else:
        H, W, Cc, Ho, Wo = conv[:5]
        x = torch.randn(M // (Ho * Wo) * H * W, Cc, device=dev).to(dt)
    dw = torch.zeros(N, k_out or K, device=dev)
    for _ in range(2):
        TE.wgrad(dz, x, dw, M, N, K, ldz=ldz, ldx=ldx, conv=conv, k_out=k_out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        TE.wgrad(dz, x, dw, M, N, K, ldz=ldz, ldx=ldx, conv=conv, k_out=k_out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


MATH = os.environ.get('GRL_WGRAD_BENCH_MATH', 'f32')       # 'bf16s': the bf16-in kernel on bf16 operands
BT = tuple(int(v) for v in os.environ.get('GRL_WGRAD_BENCH_BT', '32x4').split('x'))

if __name__ == '__main__':
    forced = [int(a) for a in sys.argv[1:]]
    shapes = record_shapes(*BT)
    tot = collections.OrderedDict([('auto', 0.0)] + [(s, 0.0) for s in forced])
    best_tot = 0.0
    print('%-62s %5s' % ('shape (M,N,K,ldz,ldx,conv,k_out)', 'calls') + '%16s' % 'auto' +
          ''.join('%10s' % ('s=%d' % s) for s in forced))
    for key, calls in shapes.items():
        os.environ.pop('GRL_WGRAD_SPLITS', None)
        ms = time_shape(key)
        fl = 2.0 * key[0] * key[1] * key[2]
        line = '%-62s %5d   %6.3f %5.1fTF' % (str(key)[:62], calls, ms, fl / ms / 1e9)
        tot['auto'] += ms * calls
        best = ms
        for s in forced:
            os.environ['GRL_WGRAD_SPLITS'] = str(s)
            m2 = time_shape(key)
            tot[s] += m2 * calls
            best = min(best, m2)
            line += '   %7.3f' % m2
        best_tot += best * calls
        print(line)
    os.environ.pop('GRL_WGRAD_SPLITS', None)
    print('per-step totals (ms): ' + '  '.join('%s=%.2f' % (k, v) for k, v in tot.items()) +
          '  best-of=%.2f' % best_tot)
