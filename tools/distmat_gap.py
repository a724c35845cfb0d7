#!/usr/bin/env python
"""the distance matrix measured 2.42 ms in the default bench line (after the train series)
and 2.17 ms in `--mode distmat` (fresh process).  Same kernel, same data: this tool times the SAME launch, per launch
with HIP events, in the states the two measurements see -- fresh process / after a sustained GEMM load / after train
steps / after the allocator was emptied -- with the clocks rocm-smi reports next to each.

  python tools/distmat_gap.py
"""
import json
import os
import subprocess
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def clocks():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, timeout=20).stdout.decode()
        j = json.loads(out)
        c = list(j.values())[0]
        keep = {}
        for k, v in c.items():
            kl = k.lower()
            if 'sclk' in kl or 'mclk' in kl or 'power' in kl or ('temperature' in kl and ('junction' in kl or 'hotspot' in kl or 'edge' in kl)):
                keep[k] = v
        return keep
    except Exception as e:                                  # noqa: BLE001
        return {'error': repr(e)[:80]}


def main():
    from grl_amd import engine, _lib
    from grl_amd.synthetic import synth_eval_features
    import bench
    dev = torch.device('cuda', 0)
    _lib.load()
    qf, gf = synth_eval_features(1980, 11310, seed=1, noise=6.0)[:2]
    qd, gd = qf.to(dev), gf.to(dev)
    out = torch.empty(1980, 11310, device=dev)

    def series(tag, n=20, warm=0, own_out=True):
        for _ in range(warm):
            engine.cosin_dist(qd, gd)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(n):
            d = engine.cosin_dist(qd, gd)
            ev[i + 1].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        per = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        rec = {"tag": tag, "wall_ms": round(wall, 3), "first5": [round(x, 3) for x in per[:5]], "min": round(min(per), 3),
               "median": round(sorted(per)[n // 2], 3), "max": round(max(per), 3), "clocks": clocks()}
        print(json.dumps(rec), flush=True)
        return rec

    series('fresh process, no warm-up, 7 launches (what secondary_block did: 2 warm-ups + 5)', n=7)
    series('fresh process, 5 warm-ups + 20 (what --mode distmat does)', n=20, warm=5)
    # sustained load: 3 s of the same GEMM
    t0 = time.time()
    while time.time() - t0 < 3.0:
        for _ in range(50):
            engine.cosin_dist(qd, gd)
        torch.cuda.synchronize()
    series('after 3 s of back-to-back launches', n=20)
    time.sleep(2.0)
    series('after 2 s idle', n=7)
    # the state of the default line: the train series just ran
    for m in ('f32', 'bf16s'):
        bench.release_cached_blocks()
        r = bench.train_series(dev, m, steps=10, warmup=3, b=32, t=4)
        print(json.dumps({"train": m, "ms": r["ms_per_step"]}), flush=True)
        series('right after the %s train series (no release)' % m, n=7)
    bench.release_cached_blocks()
    series('after release_cached_blocks()', n=7)
    series('again, 20 launches', n=20)


if __name__ == '__main__':
    main()
