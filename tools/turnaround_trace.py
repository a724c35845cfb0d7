#!/usr/bin/env python
"""Launch-by-launch listing of the fwd -> bwd turnaround of one training step from a rocprofv3 --kernel-trace CSV
(streams as in production):   python tools/turnaround_trace.py <..._kernel_trace.csv> [step]
The window is [first kernel after the CNN forward's tail (`affine_l2norm`), first `bn_bwd` / data-gradient kernel of the
CNN backward).  Per launch: queue, start offset (us), duration (us), gap to the previous END on the same queue."""
import csv
import re
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    if 'at::native' in name:
        m = re.search(r'at::native::([A-Za-z_0-9]+)', name)
        inner = re.findall(r'([A-Za-z_]+Functor|[a-z_]+_kernel[a-z_]*)', name)
        return 'torch:' + (m.group(1) if m else '?') + ('/' + inner[0] if inner else '')
    m = re.match(r'([A-Za-z_0-9:]+)', name)
    return (m.group(1) if m else name)[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    step = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    tails = [i for i, r in enumerate(rows) if 'affine_l2norm' in r['Kernel_Name']]
    # the forward's tail launches affine_l2norm twice (x_uncorr, x_corr): take the LAST of each step's pair
    ends = [i for k, i in enumerate(tails) if k + 1 == len(tails) or tails[k + 1] - i > 50]
    a = ends[step]
    b = next(i for i in range(a + 1, len(rows)) if 'bn_bwd' in rows[i]['Kernel_Name'] or 'relu_bwd' in rows[i]['Kernel_Name'])
    t0 = int(rows[a]['End_Timestamp'])
    last_end = {}
    qids = {}
    print('window: %d launches, %.1f us' % (b - a - 1, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
    busy = 0
    for r in rows[a + 1:b + 1]:
        q = qids.setdefault(r['Queue_Id'], len(qids))
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - last_end[q]) / 1e3 if q in last_end else float('nan')
        last_end[q] = e
        busy += e - s
        print('q%d %8.1f  %6.1f  gap %6.1f  %s' % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short(r['Kernel_Name'])))
    print('sum of kernel durations %.1f us' % (busy / 1e3))


if __name__ == '__main__':
    main()
