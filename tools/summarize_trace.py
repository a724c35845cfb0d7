#!/usr/bin/env python
"""Per-step kernel table from a rocprofv3 --kernel-trace CSV (any bench mode).
  python tools/summarize_trace.py <..._kernel_trace.csv> "<title>" [out.md] [out_stats.csv]
Steps are delimited by the stem kernel (it runs once per step); the first `skip` steps
(warm-up, default 2) are dropped."""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', name)
    s = m.group(1) if m else name
    if s.startswith('at::native'):
        s = 'torch: ' + name[name.find('at::native') + 12:][:60]
    return s[:80]


def main():
    path, title = sys.argv[1], sys.argv[2]
    out_md = sys.argv[3] if len(sys.argv) > 3 else None
    out_csv = sys.argv[4] if len(sys.argv) > 4 else None
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    stems = [i for i, r in enumerate(rows) if 'stem_' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name'] and 'im2col' not in r['Kernel_Name']
             and 'pack' not in r['Kernel_Name']]
    skip = 2 if len(stems) > 3 else 0
    body = rows[stems[skip]:]
    steps = len(stems) - skip
    cnt, tot = collections.Counter(), collections.Counter()
    for r in body:
        k = short(r['Kernel_Name'])
        cnt[k] += 1
        tot[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    busy = sum(tot.values()) / 1e6 / steps
    span = (int(body[-1]['End_Timestamp']) - int(body[0]['Start_Timestamp'])) / 1e6 / steps
    lines = ['# %s' % title, '', '%d steps; GPU busy %.2f ms/step, wall (profiled) %.2f ms/step' % (steps, busy, span), '',
             '| kernel | launches/step | ms/step | avg us |', '|---|---|---|---|']
    for k, v in tot.most_common(40):
        lines.append('| %s | %.1f | %.3f | %.1f |' % (k, cnt[k] / steps, v / 1e6 / steps, v / 1e3 / cnt[k]))
    text = '\n'.join(lines) + '\n'
    if out_md:
        open(out_md, 'w').write(text)
    if out_csv:
        with open(out_csv, 'w') as f:
            f.write('kernel,launches_per_step,ms_per_step,avg_us\n')
            for k, v in tot.most_common():
                f.write('"%s",%.2f,%.4f,%.2f\n' % (k, cnt[k] / steps, v / 1e6 / steps, v / 1e3 / cnt[k]))
    print(text)


if __name__ == '__main__':
    main()
