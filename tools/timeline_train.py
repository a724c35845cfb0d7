#!/usr/bin/env python
"""Train-step timeline from a rocprofv3 --kernel-trace CSV taken with the DEFAULT side streams: per step the wall time,
the time at least one kernel ran (busy), the idle time between kernels, the sum of kernel durations (sum / busy =
average concurrency), and the kernels ranked by their share of the step.
  python tools/timeline_train.py <kernel_trace.csv> [skip_steps]"""
import collections
import csv
import re
import sys


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    gaps = []
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0), gaps


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    k = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
    # a step starts at the forward stem kernel
    stems = [i for i, r in enumerate(k) if re.search(r'stem_(mfma|b16|conv7x7)_kernel|stem_pool', r[2]) and 'wgrad' not in r[2]]
    steps = [k[a:b] for a, b in zip(stems[skip:], stems[skip + 1:])]
    print('%d steps' % len(steps))
    W = B = S = 0
    big_gaps = collections.Counter()
    tot = collections.Counter(); cnt = collections.Counter()
    for st in steps:
        wall = st[-1][1] - st[0][0]
        wall = max(e for _, e, _ in st) - st[0][0]
        busy, gaps = union([(s, e) for s, e, _ in st])
        W += wall; B += busy; S += sum(e - s for s, e, _ in st)
        for g in gaps:
            big_gaps[min(int(g / 1000) // 5 * 5, 50)] += g
        for s, e, nm in st:
            nm = re.sub(r'\(anonymous namespace\)::', '', nm)
            nm = re.sub(r'^void ', '', nm)[:60]
            tot[nm] += e - s; cnt[nm] += 1
    n = len(steps)
    print('wall %.2f ms  busy %.2f ms  idle %.2f ms  kernel sum %.2f ms  (concurrency %.2f)  kernels/step %.0f' % (
        W / n / 1e6, B / n / 1e6, (W - B) / n / 1e6, S / n / 1e6, S / B, sum(cnt.values()) / n))
    print('idle time by gap length (us bucket: ms/step): ' + ', '.join('%d+: %.2f' % (b, v / n / 1e6) for b, v in sorted(big_gaps.items())))
    for nm, v in tot.most_common(25):
        print('  %8.3f ms %6.1f  %s' % (v / n / 1e6, cnt[nm] / n, nm))


if __name__ == '__main__':
    main()
