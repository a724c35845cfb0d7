#!/usr/bin/env python
"""Condense rocprofv3 outputs (kernel trace + separate PMC passes) into the
per-round files kept under profiles/.

  python tools/summarize_prof.py r01 gpurun_out/prof_stats gpurun_out/prof_fetch \
         gpurun_out/prof_write gpurun_out/prof_mfma

Per-step figures are normalised by the number of stem_conv7x7_kernel dispatches in
each run (the stem runs exactly once per step).  HBM bytes follow
MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so the read side
is doubled.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    for tag in ('gemm_f32_kernel', 'bneck_tail_f32_kernel', 'stem_pool_f32', 'stem_mfma', 'stem_conv7x7', 'maxpool3x3s2', 'group_mean', 'sqdiff_mean', 'gce_gate',
                'temporal_mean', 'add_strided', 'channel_hidden', 'channel_atte_out', 'affine_l2norm',
                'siamese_attn', 'mean_T', 'row_sqnorm', 'pair_verify', 'bn_fold', 'pack_conv_weight', 'splitk_finish'):
        if tag in name:
            if tag in ('gemm_f32_kernel', 'bneck_tail_f32_kernel'):
                return name[name.index(tag):].split('(')[0]
            return tag
    return 'other:' + name[:40]


def read_pmc(d):
    out = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    f = glob.glob(os.path.join(d, '*', '*counter_collection.csv'))
    if not f:
        return out, calls
    seen = set()
    for row in csv.DictReader(open(f[0])):
        k = short(row['Kernel_Name'])
        out[k][row['Counter_Name']] += float(row['Counter_Value'])
        key = (row['Dispatch_Id'], row['Counter_Name'])
        if (row['Dispatch_Id']) not in seen:
            seen.add(row['Dispatch_Id'])
            calls[k] += 1
    return out, calls


def main():
    tag, d_stats, d_fetch, d_write, d_mfma = sys.argv[1:6]
    os.makedirs('profiles', exist_ok=True)
    trace = glob.glob(os.path.join(d_stats, '*', '*kernel_trace.csv'))[0]
    stats = glob.glob(os.path.join(d_stats, '*', '*kernel_stats.csv'))[0]
    shutil.copy(stats, 'profiles/%s_kernel_stats.csv' % tag)
    dur, calls = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(trace)):
        k = short(row['Kernel_Name'])
        dur[k] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
        calls[k] += 1
    stem = 'stem_pool_f32' if calls.get('stem_pool_f32') else ('stem_mfma' if calls.get('stem_mfma') else 'stem_conv7x7')
    steps = calls[stem]
    # launches that are not this library's, INSIDE the steps (between the first and the last stem launch) -- the kernel-stats
    # file also counts the model's host->device weight upload before the first step (hundreds of __amd_rocclr_copyBuffer)
    trows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
    tnames = [short(r['Kernel_Name']) for r in trows]
    sidx = [i for i, n in enumerate(tnames) if n == stem]
    foreign = defaultdict(int)
    for i in range(sidx[0], sidx[-1]):
        if tnames[i].startswith('other:'):
            foreign[trows[i]['Kernel_Name'].replace('void ', '').replace('at::native::', '').replace('(anonymous namespace)::', '').split('<')[0].split('(')[0][:48]] += 1
    before = sum(1 for i in range(0, sidx[0]) if 'copyBuffer' in trows[i]['Kernel_Name'])
    fetch, fcalls = read_pmc(d_fetch)
    write, wcalls = read_pmc(d_write)
    mfma, mcalls = read_pmc(d_mfma)
    fsteps, wsteps, msteps = fcalls[stem], wcalls[stem], mcalls[stem]
    lines = ['# %s: rocprofv3 summary (bench.py, B x T = 32 x 4, fp32), per step' % tag, '',
             '| kernel | launches/step | ms/step | avg us/launch | HBM read MB/step | HBM write MB/step | HBM GB/s | MFMA busy % |',
             '|---|---|---|---|---|---|---|---|']
    tot_ms = 0.0
    gem = dict(ms=0.0, rd=0.0, wr=0.0, launches=0, busy=0.0, active=0.0)
    for k in sorted(dur, key=lambda k: -dur[k]):
        if k.startswith('other:') or k in ('bn_fold', 'pack_conv_weight'):
            continue
        ms = dur[k] / steps / 1e6
        rd = fetch[k]['FETCH_SIZE'] * 1024 * 2 / max(fsteps, 1) / 1e6
        wr = write[k]['WRITE_SIZE'] * 1024 / max(wsteps, 1) / 1e6
        busy = mfma[k]['SQ_VALU_MFMA_BUSY_CYCLES']
        active = mfma[k]['GRBM_GUI_ACTIVE']
        # MFMA_BUSY is summed over the 1024 SIMDs, GUI_ACTIVE over the 8 XCDs
        pct = 100.0 * (busy / 1024.0) / (active / 8.0) if active else 0.0
        lines.append('| %s | %.1f | %.3f | %.1f | %.0f | %.0f | %.0f | %.1f |' % (
            k, calls[k] / steps, ms, dur[k] / calls[k] / 1e3, rd, wr, (rd + wr) / max(ms, 1e-9), pct))
        tot_ms += ms
        if k.startswith('gemm_f32_kernel') or k.startswith('bneck_tail_f32_kernel'):      # (the fused tails carry conv3 + conv1)
            gem['ms'] += ms; gem['rd'] += rd; gem['wr'] += wr; gem['launches'] += calls[k] / steps
            gem['busy'] += busy; gem['active'] += active
    lines += ['', 'HBM GB/s = (FETCH_SIZE x2 + WRITE_SIZE bytes) / kernel time, against ~8000 GB/s HBM3E peak '
              '(MI355X_MICROARCH.md); MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs.',
              '', 'sum of kernel time: %.3f ms/step over %d profiled steps' % (tot_ms, steps), '',
              'launches inside the steps that are not this library\'s: %s per step (%d __amd_rocclr_copyBuffer launches of the stats file precede the first step: the weight upload)' % (', '.join('%s %.2f' % (k, v / max(steps - 1, 1)) for k, v in sorted(foreign.items())) or 'none', before), '',
              'gemm_f32_kernel + bneck_tail_f32_kernel (all instantiations): %.3f ms/step, %.0f launches/step, HBM read %.0f MB + write %.0f MB '
              'per step (FETCH_SIZE x2 gfx950 correction applied), MFMA busy %.1f %% of kernel-active cycles' % (
                  gem['ms'], gem['launches'], gem['rd'], gem['wr'],
                  100.0 * (gem['busy'] / 1024.0) / (gem['active'] / 8.0) if gem['active'] else 0.0)]
    open('profiles/%s_summary.md' % tag, 'w').write('\n'.join(lines) + '\n')
    json.dump({'hbm_bytes_per_step': int((gem['rd'] + gem['wr']) * 1e6),
               'hbm_read_bytes_per_step': int(gem['rd'] * 1e6), 'hbm_write_bytes_per_step': int(gem['wr'] * 1e6),
               'gemm_ms_per_step': gem['ms'], 'gemm_launches_per_step': gem['launches'],
               'mfma_busy_frac': (gem['busy'] / 1024.0) / (gem['active'] / 8.0) if gem['active'] else None,
               'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)'},
              open('profiles/%s_gemm_pmc.json' % tag, 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
