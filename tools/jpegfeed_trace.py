#!/usr/bin/env python
"""The eval step fed from host-resident uint8 clips (MODE=u8) or from JPEG bytes decoded on the device (MODE=jpeg), for
`rocprofv3 --kernel-trace --stats`: comparing the two runs' per-kernel averages shows whether the compute kernels run
slower next to the decoder or wait for it.   MODE=jpeg python tools/jpegfeed_trace.py"""
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    import bench
    import decode_rate
    from grl_amd import engine
    from grl_amd.reid.data.jpeg import JpegBatch
    from grl_amd.synthetic import synth_clips
    mode = os.environ.get('MODE', 'jpeg')
    steps = int(os.environ.get('STEPS', '20'))
    dev = torch.device('cuda', 0)
    cnn, siam, _, _ = bench.build_models(dev)
    B, T = 32, 4
    if mode == 'jpeg':
        item = JpegBatch(decode_rate.make_frames(B * T), (B, T))
    else:
        item = synth_clips(B, T, seed=0, raw=True).pin_memory()

    def loader(k):
        for _ in range(k):
            yield item, None, None
    for d, _, _ in engine.DevicePrefetcher(loader(4), dev):
        engine.extract_features(cnn, siam, d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for d, _, _ in engine.DevicePrefetcher(loader(steps), dev):
        engine.extract_features(cnn, siam, d)
    torch.cuda.synchronize()
    print('%s: %.3f ms per step' % (mode, (time.perf_counter() - t0) / steps * 1e3))


if __name__ == '__main__':
    main()
