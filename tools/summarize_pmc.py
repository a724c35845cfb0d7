#!/usr/bin/env python
"""Per-kernel table from three separate rocprofv3 --pmc passes of one command (MFMA busy, FETCH_SIZE,
WRITE_SIZE -- the guide's HBM recipe: separate passes, FETCH_SIZE doubled on gfx950):

  python tools/summarize_pmc.py <title> <steps> <mfma_dir> <fetch_dir> <write_dir> > profiles/rNN_x.md
"""
import collections
import csv
import glob
import os
import sys


DROPPED = [0]


def load(d):
    """Per kernel: counter sums, dispatch count, total duration.  A dispatch that took more than 10 ms (no launch of these
    workloads runs longer than ~2 ms; a first-launch code-object load inside the profiled pass does: one 31 ms dispatch of
    a 0.4 ms kernel was seen) is left out and the kernel's sums are scaled back to the full dispatch count."""
    f = glob.glob(os.path.join(d, '*', '*counter_collection.csv'))[0]
    disp = {}
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        e = disp.get(r['Dispatch_Id'])
        if e is None:
            e = disp[r['Dispatch_Id']] = (name, float(r['End_Timestamp']) - float(r['Start_Timestamp']), collections.defaultdict(float))
        e[2][r['Counter_Name']] += float(r['Counter_Value'])
    by = collections.defaultdict(list)
    for name, t, c in disp.values():
        by[name].append((t, c))
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur = collections.Counter(), collections.defaultdict(float)
    for name, lst in by.items():
        kept = [(t, c) for t, c in lst if t <= 10e6] or lst
        DROPPED[0] += len(lst) - len(kept)
        k = len(lst) / float(len(kept))
        cnt[name] = len(lst)
        dur[name] = sum(t for t, _ in kept) * k
        for _, c in kept:
            for cn, v in c.items():
                val[name][cn] += v * k
    return val, cnt, dur


def main():
    # optional: --json <out.json> --series <name> [--dominant <regex>]: the totals bench.py reads back (with the
    # fingerprint of the library the passes ran on, tools/fingerprint.py)
    opts = {}
    argv = list(sys.argv)
    for flag in ('--json', '--series', '--dominant'):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    sys.argv = argv
    title, steps = sys.argv[1], int(sys.argv[2])
    sq, cnt, dur = load(sys.argv[3])
    if steps <= 0:
        # auto: the forward stem kernel runs exactly once per step (warm-up, timed and live-timing steps alike)
        stems = [k for k in cnt if 'stem' in k and 'wgrad' not in k and 'pack' not in k and 'tail' not in k and 'bwd' not in k]
        steps = int(max(cnt[k] for k in stems)) if stems else 1
    fe, _, _ = load(sys.argv[4])
    wr, _, _ = load(sys.argv[5])
    rows = []
    for k in sq:
        gui, mf = sq[k].get('GRBM_GUI_ACTIVE', 0.0), sq[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        rd = fe[k].get('FETCH_SIZE', 0.0) * 1024 * 2 / steps
        w = wr[k].get('WRITE_SIZE', 0.0) * 1024 / steps
        rows.append((dur[k] / steps / 1e6, k, cnt[k] / steps, (mf / 1024.0) / (gui / 8.0) if gui else 0.0, rd, w))
    rows.sort(reverse=True)
    print('# %s' % title)
    print()
    print('Per step, over %d profiled steps.  Three separate `rocprofv3 --pmc` passes (SQ_VALU_MFMA_BUSY_CYCLES + '
          'GRBM_GUI_ACTIVE; FETCH_SIZE; WRITE_SIZE).  HBM bytes = FETCH_SIZE x 1024 x 2 (gfx950: the counter tallies 64 B '
          'per 128-B request) + WRITE_SIZE x 1024; they count requests leaving the XCD L2s, Infinity-Cache hits included '
          '(upper bound on HBM).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs.  '
          'ms/step is from the profiled (counter) pass, a few percent slower than an unprofiled run.' % steps)
    print()
    print('| kernel | launches/step | ms/step | MFMA busy % | read GB/step | write GB/step | TB/s |')
    print('|---|---|---|---|---|---|---|')
    tot = 0.0
    for ms, k, c, frac, rd, w in rows:
        tot += ms
        if ms < 0.05:
            continue
        print('| %s | %.1f | %.3f | %.1f | %.2f | %.2f | %.2f |' % (k, c, ms, 100 * frac, rd / 1e9, w / 1e9,
                                                                  (rd + w) / 1e12 / (ms / 1e3) if ms else 0.0))
    print()
    print('sum of kernel time %.2f ms/step; read %.1f GB + write %.1f GB per step' % (
        tot, sum(r[4] for r in rows) / 1e9, sum(r[5] for r in rows) / 1e9))
    if '--json' in opts:
        import json
        import re
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import fingerprint
        pat = re.compile(opts.get('--dominant', 'gemm_|bneck_tail|wgrad|conv3x3_c64'))
        dom = [r for r in rows if pat.search(r[1])]
        gui = sum(sq[r[1]].get('GRBM_GUI_ACTIVE', 0.0) for r in dom)
        mf = sum(sq[r[1]].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for r in dom)
        rec = {"series": opts.get('--series', title), "title": title, "steps_profiled": steps,
               "hbm_read_bytes_per_step": int(sum(r[4] for r in rows)), "hbm_write_bytes_per_step": int(sum(r[5] for r in rows)),
               "hbm_bytes_per_step": int(sum(r[4] + r[5] for r in rows)), "kernel_ms_per_step": round(tot, 4),
               "dominant": {"pattern": pat.pattern, "launches_per_step": round(sum(r[2] for r in dom), 1),
                            "ms_per_step": round(sum(r[0] for r in dom), 4),
                            "hbm_bytes_per_step": int(sum(r[4] + r[5] for r in dom)),
                            "mfma_busy_frac": round((mf / 1024.0) / (gui / 8.0), 4) if gui else None},
               "note": "three separate rocprofv3 --pmc passes (MFMA busy; FETCH_SIZE; WRITE_SIZE), single stream; HBM bytes = "
                       "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, summed over every kernel of a step"}
        rec.update(fingerprint.fingerprint())
        with open(opts['--json'], 'w') as fh:
            json.dump(rec, fh, indent=1)
    if DROPPED[0]:
        print()
        print('(%d dispatches over the three passes took more than 10 ms (one-off stalls inside the profiled pass) and were left out, their '
              'kernels\' sums scaled back to the full count)' % DROPPED[0])


if __name__ == '__main__':
    main()
