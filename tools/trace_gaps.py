#!/usr/bin/env python
"""Where does a step's wall time go that is NOT kernel time?  From a rocprofv3 --kernel-trace CSV (streams as in
production, i.e. NOT the single-stream profile mode):
  python tools/trace_gaps.py <..._kernel_trace.csv> [n_gaps]
Per step (delimited by the stem kernel, first two dropped): wall span, the union of kernel intervals (chip busy with
at least one kernel), idle = span - union, the histogram of idle gaps, the idle time attributed to the kernel that
FOLLOWS the gap (who was the chain waiting to launch), and the time during which only "tiny" kernels (< 12 us) ran."""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', name)
    s = m.group(1) if m else name
    if s.startswith('at::native'):
        s = 'torch: ' + name[name.find('at::native') + 12:][:50]
    return s[:70]


def main():
    path = sys.argv[1]
    ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    stems = [i for i, r in enumerate(rows) if 'stem_' in r['Kernel_Name'] and 'im2col' not in r['Kernel_Name']
             and 'pack' not in r['Kernel_Name']]
    skip = 2 if len(stems) > 3 else 0
    bounds = stems[skip:] + [len(rows)]
    nsteps = len(bounds) - 1
    after = collections.Counter()
    after_n = collections.Counter()
    before = collections.Counter()
    hist = collections.Counter()
    tot_span = tot_union = tot_tiny = 0.0
    biggest = []
    for s in range(nsteps - 1):                       # the last step has no following stem to close it: skip it
        body = rows[bounds[s]:bounds[s + 1]]
        t0 = int(body[0]['Start_Timestamp'])
        t1 = int(rows[bounds[s + 1]]['Start_Timestamp'])
        tot_span += t1 - t0
        cur_end, cur_name = t0, 'step start'
        union = 0
        # "tiny-only" time: sweep over interval end points
        ev = []
        for r in body:
            a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            big = (b - a) >= 12000
            ev.append((a, 1, big))
            ev.append((b, -1, big))
            if a > cur_end:
                g = a - cur_end
                k = short(r['Kernel_Name'])
                after[k] += g
                after_n[k] += 1
                before[cur_name] += g
                hist[min(int(g / 1000) // 5 * 5, 50)] += 1
                biggest.append((g, cur_name, k))
                union += b - a
                cur_end, cur_name = b, k
            elif b > cur_end:
                union += b - cur_end
                cur_end, cur_name = b, short(r['Kernel_Name'])
        tot_union += union
        ev.sort()
        nbig = nall = 0
        last = t0
        for t, d, big in ev:
            if nall > 0 and nbig == 0:
                tot_tiny += t - last
            last = t
            nall += d
            if big:
                nbig += d
    n = max(nsteps - 1, 1)
    print('%d steps: span %.2f ms/step, chip busy (union of kernels) %.2f, idle %.2f, only tiny (<12us) kernels running %.2f'
          % (n, tot_span / n / 1e6, tot_union / n / 1e6, (tot_span - tot_union) / n / 1e6, tot_tiny / n / 1e6))
    print('idle gap histogram (us bucket: gaps/step): ' +
          '  '.join('%d+:%.0f' % (k, v / n) for k, v in sorted(hist.items())))
    print('\nidle time by the kernel that FOLLOWS the gap (ms/step, gaps/step):')
    for k, v in after.most_common(ngaps):
        print('  %-72s %7.3f  %6.1f' % (k, v / n / 1e6, after_n[k] / n))
    print('\nidle time by the kernel that PRECEDES the gap (ms/step):')
    for k, v in before.most_common(ngaps):
        print('  %-72s %7.3f' % (k, v / n / 1e6))
    biggest.sort(reverse=True)
    print('\nlargest gaps (us): ' + '; '.join('%.0f %s -> %s' % (g / 1e3, a, b) for g, a, b in biggest[:12]))


if __name__ == '__main__':
    main()
