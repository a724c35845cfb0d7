#!/usr/bin/env python
"""Is the fp32 GEMM clock/power limited?  Same launch (16384 x 2048 x 2048, 128x128 tiles), operands of different
bit activity: zeros, constants, post-ReLU-like (half zeros), dense random.  Prints TFLOP/s per operand kind and the
sclk rocm-smi reports while the launch loops.
  python tools/gemm_power_probe.py"""
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import engine


def smi():
    try:
        out = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if 'sclk' in l or 'Power' in l or 'mclk' in l]
        return ' | '.join(keep)
    except Exception as e:      # noqa
        return 'rocm-smi: %r' % (e,)


def main():
    dev = torch.device('cuda:0')
    M, N, K = 16384, 2048, 2048
    kinds = {
        'zeros': lambda *s: torch.zeros(*s, device=dev),
        'ones': lambda *s: torch.ones(*s, device=dev),
        'relu(randn)': lambda *s: torch.relu(torch.randn(*s, device=dev)),
        'randn': lambda *s: torch.randn(*s, device=dev),
    }
    y = torch.empty(M, N, device=dev)
    sc, sh = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    for name, mk in kinds.items():
        a, w = mk(M, K), mk(N, K) * 0.05
        for _ in range(3):
            engine.gemm(a, w, y, M, N, K, scale=sc, shift=sh, relu=True)
        torch.cuda.synchronize()
        box = {}
        th = threading.Thread(target=lambda: box.setdefault('smi', smi()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 600                                   # ~0.6 s: long enough for the clock governor and for one smi sample
        th.start()
        e0.record()
        for _ in range(iters):
            engine.gemm(a, w, y, M, N, K, scale=sc, shift=sh, relu=True)
        e1.record()
        torch.cuda.synchronize()
        th.join()
        ms = e0.elapsed_time(e1) / iters
        print('%-12s %.3f ms  %6.1f TFLOP/s   %s' % (name, ms, 2.0 * M * N * K / ms / 1e9, box.get('smi')), flush=True)
        time.sleep(1.0)


if __name__ == '__main__':
    main()
