#!/usr/bin/env python
"""Timeline of one eval step from a rocprofv3 --kernel-trace CSV taken with the DEFAULT streams (kernels of the two
TRL directions overlap): per phase -- trunk, GCE, TRL, tail -- the wall time, the time at least one kernel was running,
the idle gaps, and the sum of kernel durations (sum / busy = average concurrency).
  python tools/timeline.py <kernel_trace.csv> [skip_steps]"""
import csv
import sys


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    k = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
    stems = [i for i, r in enumerate(k) if 'stem_' in r[2] and 'pack' not in r[2]]
    steps = []
    for a, b in zip(stems[skip:], stems[skip + 1:]):
        steps.append(k[a:b])
    print('%d steps analysed' % len(steps))
    acc = {}
    for st in steps:
        names = [n for _, _, n in st]
        # phase boundaries: GCE starts at the first group_mean after the trunk; TRL at temporal_mean; tail at affine_l2norm
        def first(pred, start=0):
            for i in range(start, len(st)):
                if pred(names[i]):
                    return i
            return len(st)
        i_gce = first(lambda n: 'group_mean' in n)
        i_trl = first(lambda n: 'temporal_mean' in n, i_gce)
        i_tail = first(lambda n: 'affine_l2norm' in n, i_trl)
        bounds = [('trunk', 0, i_gce), ('GCE', i_gce, i_trl), ('TRL', i_trl, i_tail), ('tail', i_tail, len(st))]
        for name, a, b in bounds:
            seg = st[a:b]
            if not seg:
                continue
            t0 = min(s for s, _, _ in seg)
            t1 = max(e for _, e, _ in seg)
            nxt = st[b][0] if b < len(st) else t1
            wall = max(t1, nxt) - t0 if name != 'tail' else t1 - t0
            busy = union([(s, e) for s, e, _ in seg])
            ksum = sum(e - s for s, e, _ in seg)
            d = acc.setdefault(name, [0, 0, 0, 0])
            d[0] += wall; d[1] += busy; d[2] += ksum; d[3] += len(seg)
    n = len(steps)
    print('%-6s %10s %10s %10s %10s %8s' % ('phase', 'wall ms', 'busy ms', 'idle ms', 'sum ms', 'kernels'))
    tw = 0
    for name in ('trunk', 'GCE', 'TRL', 'tail'):
        if name in acc:
            w, b, s, c = acc[name]
            tw += w
            print('%-6s %10.3f %10.3f %10.3f %10.3f %8.1f' % (name, w / n / 1e6, b / n / 1e6, (w - b) / n / 1e6, s / n / 1e6, c / n))
    print('total wall %.3f ms/step' % (tw / n / 1e6))
    # the TRL phase in detail: per-kernel sums
    import collections, re
    tot = collections.Counter(); cnt = collections.Counter()
    for st in steps:
        names = [x[2] for x in st]
        a = next((i for i, x in enumerate(names) if 'temporal_mean' in x), len(st))
        b = next((i for i in range(a, len(st)) if 'affine_l2norm' in names[i]), len(st))
        for s, e, nm in st[a:b]:
            nm = re.sub(r'\(anonymous namespace\)::', '', nm)[:70]
            tot[nm] += e - s; cnt[nm] += 1
    print('\nTRL phase kernels (ms/step, launches/step):')
    for nm, v in tot.most_common(14):
        print('  %8.3f %6.1f  %s' % (v / n / 1e6, cnt[nm] / n, nm))


if __name__ == '__main__':
    main()
