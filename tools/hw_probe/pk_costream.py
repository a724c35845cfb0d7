#!/usr/bin/env python
"""Reproduces the packed-fp32 / co-resident-kernel corruption that made the build switch packed fp32
VALU instructions off (grl_amd/csrc/Makefile NOPK).

    make -C grl_amd/csrc clean all NOPK=        # library WITH v_pk_mul_f32 / v_pk_add_f32
    python tools/hw_probe/pk_costream.py        # -> "bad 45/60" on MI355X (ROCm 7.2)
    make -C grl_amd/csrc clean all              # default build (no packed fp32)
    python tools/hw_probe/pk_costream.py        # -> "bad 0/60"

Stream 1 runs one bf16-storage GEMM (4096 x 2048 x 2048, ~60 us); stream 2 runs one TRL channel-attention step
(`grl_channel_atte`: channel_hidden + channel_atte_out, ~25 us) that lands on the CUs while the GEMM's waves
retire.  Inputs are constant, so every run must reproduce the quiet result bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from grl_amd.engine import ptr, _call, gemm, MATH_BF16S

dev = torch.device('cuda:0')
torch.manual_seed(0)
b, t, Cc, Hd, Mb = 32, 4, 2048, 128, 4096
shift = torch.randn(Cc, device=dev) * 0.1
dvec = torch.rand(b, Cc, device=dev)
w1 = torch.randn(Hd, Cc, device=dev) * 0.05
w2t = torch.randn(Hd, Cc, device=dev) * 0.05
gapc = torch.rand(b * t, Cc, device=dev)
a = (torch.randn(Mb, Cc, device=dev) * 0.5).bfloat16()
w = (torch.randn(Cc, Cc, device=dev) * 0.02).bfloat16()
y = torch.empty(Mb, Cc, device=dev, dtype=torch.bfloat16)
fc, hid = torch.empty(b, t, Cc, device=dev), torch.empty(b, Hd, device=dev)


def victim():
    _call('grl_channel_atte', ptr(dvec), ptr(w1), ptr(w2t), ptr(gapc), t * Cc, None, ptr(fc), t * Cc, 0, b, Cc, Hd,
          ptr(hid))


victim()
torch.cuda.synchronize()
ref = fc[:, 0].clone()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
bad, worst = 0, 0
for it in range(60):
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        gemm(a, w, y, Mb, Cc, Cc, shift=shift, relu=True, math=MATH_BF16S)
    with torch.cuda.stream(s2):
        victim()
    torch.cuda.synchronize()
    n = int((fc[:, 0] != ref).sum())
    bad += n != 0
    worst = max(worst, n)
print('channel_atte next to a bf16s GEMM on another stream: bad %d/60 (max differing elements %d)' % (bad, worst))
