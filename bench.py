#!/usr/bin/env python
"""Headline benchmark: clip-features/sec of the GRL eval path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic MARS-shaped clips
(BASELINE.json configs[1]: B x T = 32 x 4, 256 x 128, fp32): ResNet-50 trunk + GCE +
TRL + Siamese temporal attention + concat -> one 6144-d feature row per clip
(reference: reid/evaluator/attevaluator.py:100-112).  Inputs are resident in HBM
before the timed region.  With --gpus N every rank processes its own batch: clips are
independent, there is no data-path collective (weak scaling); the barrier only brackets the
timed region.  `python bench.py --gpus N` by itself (no WORLD_SIZE in the environment) STARTS the
N ranks -- `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process,
before anything in this process touches the GPU -- and forwards the child's line; launched by
torch.distributed.run it is one of the ranks.  The N > 1 line carries `rccl_ranks` /
`dist_backend` from the initialised process group, and every line carries a `train` block: the
SEQTrainer step of BASELINE configs[3] (64 clips x 4 frames per GPU, GradSync over RCCL) -- the
series in which the ranks actually exchange data.

Prints ONE JSON line (rank 0).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import threading
import time

# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC); set before HIP initialises
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, T = 32, 4
GFLOP_PER_FRAME = 14.485                # conv+linear forward, per frame (57.94 per clip at T=4)
PEAK_FP32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA (the 5 PF headline figure is 2:1 sparse)
TRAIN_FLOP_FACTOR = 3.0                 # one train step = 3 x the forward's conv + linear FLOPs
# peak of the datapath that carries the series' GEMMs ('mixed': exact fp32 forward = a third of the FLOPs at the fp32
# peak, two thirds on split-bf16 products at 2500/3: the time-weighted harmonic peak)
SERIES_PEAK = {'f32': PEAK_FP32_MFMA_TFLOPS, 'bf16': PEAK_BF16_MFMA_TFLOPS, 'bf16s': PEAK_BF16_MFMA_TFLOPS,
               'bf16x3': PEAK_BF16_MFMA_TFLOPS / 3,
               'mixed': 3.0 / (1.0 / PEAK_FP32_MFMA_TFLOPS + 2.0 / (PEAK_BF16_MFMA_TFLOPS / 3))}


def max_over_ranks(dist, dev, seconds):
    """MAX of a per-rank wall time (RCCL reduces device tensors, gloo host tensors)."""
    if dist is None:
        return seconds
    tt = torch.tensor([seconds], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def build_models(dev):
    from grl_amd.reid import models
    from grl_amd.synthetic import synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625,
                            pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    sd = synth_state_dict(cnn, seed=0)
    ssd = synth_state_dict(siam, seed=0, prefix='siamese.')
    cnn.load_state_dict(sd)
    siam.load_state_dict(ssd)
    return cnn.to(dev).eval(), siam.to(dev).eval(), sd, ssd


PEAK_HBM_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def _nbytes(rows, cols, t):
    return float(rows) * cols * t.element_size()


def gemm_roofline(cnn, siam, clips, iters=3, stages=None):
    """Live per-launch timing of the dominant kernel (gemm_f32_kernel: every conv /
    linear of the path) with HIP events on the launch stream, outside the timed
    region.  achieved = algorithmic FLOPs of all its launches in one step (2*M*N*K per
    launch, counted by the host wrapper) / sum of their measured durations.
    ``stages``: a dict to fill with the per-stage record (stem, layer1-4, gce, trl, tail): wall ms of the stage on one
    stream (events at engine.STAGE_HOOK boundaries), the algorithmic FLOPs and the ALGORITHMIC bytes (operands read once,
    outputs written once, weights once per launch) of the GEMM / fused-tail launches booked to it."""
    from grl_amd import engine
    recs = []
    cur = ['pre']
    marks = []
    orig, orig_tail, orig_tail32, orig_c64 = engine.gemm, engine.bneck_tail_bf16, engine.bneck_tail_f32, engine.conv3x3_c64_bf16

    def hook(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))
        cur[0] = name

    def timed(a, w, y, M, N, K, *args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(a, w, y, M, N, K, *args, **kw)
        e1.record()
        conv = kw.get('conv')
        rows_in, cin = (M, K) if conv is None else ((M // (conv[3] * conv[4])) * conv[0] * conv[1], conv[2])
        by = _nbytes(rows_in, cin, a) + _nbytes(N, K, w)
        sq = kw.get('epilogue', engine.EPI_AFFINE) == engine.EPI_SQDIFF
        by += _nbytes(M // 32 if sq else M, N, y)
        if kw.get('res') is not None:
            by += _nbytes(M, N, kw['res'])
        recs.append((2.0 * M * N * K, e0, e1, (M, N, K, conv), cur[0], by))
        return out

    def timed_tail_of(fn):
        def timed_tail(t2, c3, res, c1n, M, **kw):
            # a fused bottleneck tail (fuse_bf16.hip / fuse_f32.hip) carries two of the path's convolutions: conv3 and the
            # next block's conv1 -- counted with the GEMM launches they replace (same algorithmic FLOPs, their own time)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(t2, c3, res, c1n, M, **kw)
            e1.record()
            pn = c1n.N if c1n is not None else 0
            kd = kw['down'].K if kw.get('down') is not None else 0         # (+ the block's downsample conv)
            by = _nbytes(M, c3.K, t2) + _nbytes(M, c3.N, t2) + _nbytes(M, pn, t2)       # t2 in, 4P-wide out, conv1' out
            by += _nbytes(M, kd, t2) if kd else _nbytes(M, c3.N, t2)                     # the block input x0, or the residual
            by += (c3.N * (c3.K + pn + kd)) * t2.element_size()
            recs.append((2.0 * M * c3.N * (c3.K + pn + kd), e0, e1,
                         (M, c3.N, c3.K, 'fused tail + conv1 -> %d%s' % (pn, ' + downsample' if kd else '')), cur[0], by))
            return out
        return timed_tail

    def timed_c64(x, c, n_img, H, W, relu=True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_c64(x, c, n_img, H, W, relu)
        e1.record()
        recs.append((2.0 * n_img * H * W * 64 * 576, e0, e1, (n_img * H * W, 64, 576, 'conv3x3_c64'), cur[0],
                     2 * _nbytes(n_img * H * W, 64, x) + 64 * 576 * x.element_size()))
        return out

    streams, engine.TRL_STREAMS = engine.TRL_STREAMS, False     # one stream: a launch's events bracket that launch alone
    engine.extract_features(cnn, siam, clips)                   # untimed: the caching allocator re-settles on one stream
    torch.cuda.synchronize()                                    # (a hipMalloc between two events would count as GEMM time)
    engine.gemm, engine.bneck_tail_bf16, engine.bneck_tail_f32 = timed, timed_tail_of(orig_tail), timed_tail_of(orig_tail32)
    engine.conv3x3_c64_bf16 = timed_c64
    engine.STAGE_HOOK = hook
    try:
        for _ in range(iters):
            engine.extract_features(cnn, siam, clips)
            hook('end')
        torch.cuda.synchronize()
    finally:
        engine.gemm, engine.bneck_tail_bf16, engine.bneck_tail_f32, engine.conv3x3_c64_bf16 = orig, orig_tail, orig_tail32, orig_c64
        engine.TRL_STREAMS = streams
        engine.STAGE_HOOK = None
    flops = sum(r[0] for r in recs) / iters
    ms = sum(r[1].elapsed_time(r[2]) for r in recs) / iters
    launches = len(recs) // iters
    if stages is not None:
        wall = {}
        for (name, e), (_, e_next) in zip(marks[:-1], marks[1:]):
            if name != 'end':
                wall[name] = wall.get(name, 0.0) + e.elapsed_time(e_next) / iters
        peak = SERIES_PEAK['bf16s' if engine.get_math() == 'bf16s' else 'f32']
        for name, w_ms in wall.items():
            sel = [r for r in recs if r[4] == name]
            fl = sum(r[0] for r in sel) / iters
            by = sum(r[5] for r in sel) / iters
            k_ms = sum(r[1].elapsed_time(r[2]) for r in sel) / iters
            stages[name] = {"wall_ms": round(w_ms, 3), "gemm_ms": round(k_ms, 3), "gemm_launches": len(sel) // iters,
                            "gflop": round(fl / 1e9, 1), "tflops": round(fl / max(w_ms, 1e-9) / 1e9, 1),
                            "mfma_util": round(fl / max(w_ms, 1e-9) / 1e9 / peak, 4),
                            "mfma_util_in_gemm_time": round(fl / max(k_ms, 1e-9) / 1e9 / peak, 4) if sel else None,
                            "algorithmic_gb": round(by / 1e9, 3), "hbm_gbs": round(by / max(w_ms, 1e-9) / 1e6, 1),
                            "hbm_frac": round(by / max(w_ms, 1e-9) / 1e6 / PEAK_HBM_GBS, 4)}
    if os.environ.get('GRL_GEMM_REPORT'):
        agg = {}
        for f, a, b2, shape, _, _ in recs:
            e = agg.setdefault(str(shape), [0, 0.0, f])
            e[0] += 1
            e[1] += a.elapsed_time(b2)
        rows = sorted(((k, v[0] // iters, v[1] / iters, v[2]) for k, v in agg.items()), key=lambda r: -r[2])
        with open(os.environ['GRL_GEMM_REPORT'], 'w') as fh:
            for k, cnt, tms, f in rows:
                fh.write('%-60s calls %3d  total %8.3f ms  %7.1f TF/s\n' % (k, cnt, tms, f * cnt / (tms * 1e-3) / 1e12))
    return flops, ms, launches


PMC_ROUND = 'r06'


def pmc_record(series):
    """Counter traffic of `series` from profiles/r06_pmc_<series>.json -- ONLY if that file was written by
    tools/profile_round.sh on the library this process runs (sha256 of libgrl_hip.so, or of the sources it is built
    from: tools/fingerprint.py).  Returns (record, None) or (None, reason): never an older round's file, never a file
    of another build (VERDICT r5 measurement item 8)."""
    path = os.path.join(ROOT, 'profiles', '%s_pmc_%s.json' % (PMC_ROUND, series))
    if not os.path.isfile(path):
        return None, 'no profiles/%s_pmc_%s.json' % (PMC_ROUND, series)
    try:
        rec = json.load(open(path))
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import fingerprint
        fp = fingerprint.fingerprint()
    except Exception as e:                              # noqa: BLE001
        return None, 'unreadable: %r' % (e,)
    if rec.get('lib_sha256') != fp['lib_sha256'] and rec.get('src_sha256') != fp['src_sha256']:
        return None, ('profiles/%s_pmc_%s.json was taken on another build (library %s..., sources %s...; running %s... / %s...)'
                      % (PMC_ROUND, series, str(rec.get('lib_sha256'))[:10], str(rec.get('src_sha256'))[:10],
                         str(fp['lib_sha256'])[:10], fp['src_sha256'][:10]))
    return rec, None


def series_roofline(math, clips, frames_per_clip, ms_per_step, train=False, kernel=None, pmc=None, alg_bytes=None):
    """`roofline` object of a secondary series.  MFMA view: ALGORITHMIC FLOPs of one step (14.485 GFLOP
    per frame forward, x 3 for a train step) / the step's measured wall time against the dense MFMA peak of the datapath.
    HBM view: `alg_bytes` (every GEMM / weight-gradient operand read once, every output written once: what a step with
    every BatchNorm / pointwise pass fused away would move) / wall time against 8 TB/s.  `bound` = whichever of
    traffic / 8 TB/s (counter traffic when a profile of THIS build is committed, else the algorithmic bytes) and
    FLOPs / peak is larger; `achieved` / `peak` / `frac` are quoted in that bound's unit, the other view rides along.
    `kernel` = the dominant kernel family of that step timed live with HIP events (one stream)."""
    gflop = clips * frames_per_clip * GFLOP_PER_FRAME * (TRAIN_FLOP_FACTOR if train else 1.0)
    achieved = gflop / ms_per_step          # GFLOP / ms = TFLOP/s
    peak = SERIES_PEAK[math]
    rec, why = pmc_record(pmc) if pmc else (None, 'no profiled series for this configuration')
    traffic = rec['hbm_bytes_per_step'] if rec else None
    mfma = {"achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4)}
    t_mfma = gflop / peak                                                   # ms
    byts = traffic if traffic is not None else alg_bytes
    t_hbm = byts / (PEAK_HBM_GBS * 1e6) if byts else 0.0                    # ms
    r = dict(mfma, bound="mfma")
    if t_hbm > t_mfma and alg_bytes:
        gbs = alg_bytes / ms_per_step / 1e6
        r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
             "mfma_view": mfma}
    r["algorithmic_gflop_per_step"] = round(gflop, 1)
    r["algorithmic_gb_per_step"] = round(alg_bytes / 1e9, 2) if alg_bytes else None
    r["min_ms_at_peak"] = {"mfma": round(t_mfma, 3), "hbm": round(t_hbm, 3),
                           "hbm_from": "counter traffic" if traffic is not None else "algorithmic bytes"}
    r["traffic"] = traffic
    if rec:
        r["traffic_source"] = "profiles/%s_pmc_%s.json (same build: fingerprint checked), %.2f x the algorithmic bytes" % (
            PMC_ROUND, pmc, traffic / alg_bytes) if alg_bytes else "profiles/%s_pmc_%s.json (same build)" % (PMC_ROUND, pmc)
        r["mfma_busy_dominant_kernels"] = rec.get('dominant', {}).get('mfma_busy_frac')
    else:
        r["traffic_reason_null"] = why
    if kernel is not None:
        r["kernel"] = kernel
    return r


def eval_kernel_timing(cnn, siam, clips, math):
    """Dominant-kernel record of an eval series: every GEMM launch (engine.gemm) of one step, HIP events; and the
    per-stage table (north_star's 40 % target is per ResNet-50 stage)."""
    from grl_amd import engine
    stages = {}
    with engine.math_mode(math):
        flops, ms, launches = gemm_roofline(cnn, siam, clips, stages=stages)
    return {"name": "gemm_bf16_256_kernel / gemm_f32_kernel<.., bf16 storage> / bneck_tail_kernel (implicit-GEMM conv + linear, fused bottleneck tails)" if math == 'bf16s'
            else "gemm_f32_kernel", "launches_per_step": launches, "ms_per_step": round(ms, 3),
            "gflop_per_step": round(flops / 1e9, 1), "tflops": round(flops / ms / 1e9, 1),
            "frac_of_peak": round(flops / ms / 1e9 / SERIES_PEAK[math], 4),
            "algorithmic_bytes_per_step": sum(v["algorithmic_gb"] for v in stages.values()) * 1e9,
            "stages (one stream; mfma_util = algorithmic FLOPs / stage wall time / MFMA peak, hbm_gbs = algorithmic bytes / stage wall time)": stages}


def train_kernel_timing(tr, clips, pids, math, iters=2):
    """Dominant-kernel record of a train series: every forward / data-gradient GEMM (engine.gemm) and every weight
    gradient (train_engine.wgrad) launch of one step, HIP events, with the weight-gradient and TRL side streams off so
    that a launch's two events bracket that launch alone.  Outside any timed region."""
    from grl_amd import engine, train_engine as TE
    recs = []
    og, ow = engine.gemm, TE.wgrad

    def tg(a, w, y, M, N, K, *args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = og(a, w, y, M, N, K, *args, **kw); e1.record()
        conv = kw.get('conv')
        rows_in, cin = (M, K) if conv is None else ((M // (conv[3] * conv[4])) * conv[0] * conv[1], conv[2])
        by = _nbytes(rows_in, cin, a) + _nbytes(N, K, w) + _nbytes(M, N, y)
        if kw.get('res') is not None:
            by += _nbytes(M, N, kw['res'])
        recs.append(('gemm', 2.0 * M * N * K, e0, e1, by))
        return r

    def tw(dz, x, dw, M, N, K, *args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = ow(dz, x, dw, M, N, K, *args, **kw); e1.record()
        conv = kw.get('conv')
        rows_in, cin = (M, K) if conv is None else ((M // (conv[3] * conv[4])) * conv[0] * conv[1], conv[2])
        recs.append(('wgrad', 2.0 * M * N * K, e0, e1, _nbytes(M, N, dz) + _nbytes(rows_in, cin, x) + 4.0 * N * K))
        return r

    def one():
        loss, _, _, _ = tr._forward([clips], pids, 0, 0)
        for p in tr._all_params():
            p.grad = None
        loss.backward()

    saved = (engine.TRL_STREAMS, TE.WGRAD_STREAM)
    engine.TRL_STREAMS, TE.WGRAD_STREAM = False, False
    old = TE.set_math(math)
    try:
        one()                                   # untimed: allocator settles on one stream
        torch.cuda.synchronize()
        engine.gemm, TE.wgrad = tg, tw
        for _ in range(iters):
            one()
        torch.cuda.synchronize()
    finally:
        engine.gemm, TE.wgrad = og, ow
        engine.TRL_STREAMS, TE.WGRAD_STREAM = saved
        TE.set_math(old)
    out = {}
    for kind in ('gemm', 'wgrad'):
        sel = [r for r in recs if r[0] == kind]
        ms = sum(r[2].elapsed_time(r[3]) for r in sel) / iters
        fl = sum(r[1] for r in sel) / iters
        out[kind] = {"launches_per_step": len(sel) // iters, "ms_per_step": round(ms, 3),
                     "gflop_per_step": round(fl / 1e9, 1), "tflops": round(fl / max(ms, 1e-9) / 1e9, 1),
                     "algorithmic_gb_per_step": round(sum(r[4] for r in sel) / iters / 1e9, 2)}
    tot_ms = out['gemm']['ms_per_step'] + out['wgrad']['ms_per_step']
    tot_fl = out['gemm']['gflop_per_step'] + out['wgrad']['gflop_per_step']
    return {"name": "forward + data-gradient GEMMs (gemm_f32_kernel / gemm_bf16_256_kernel) and weight gradients "
                    "(wgrad_kernel / wgrad_b16in*), one stream",
            "forward_and_dgrad": out['gemm'], "wgrad": out['wgrad'], "ms_per_step": round(tot_ms, 3),
            "algorithmic_bytes_per_step": sum(r[4] for r in recs) / iters,
            "tflops": round(tot_fl / max(tot_ms, 1e-9), 1),
            "frac_of_peak": round(tot_fl / max(tot_ms, 1e-9) / SERIES_PEAK[math], 4)}


def cpu_baseline(sd, ssd):
    """The oracle (plain PyTorch-CPU restatement of the reference path) timed on this
    node's host cores on a bounded sample of the same workload."""
    from oracle import grl_oracle as O
    from grl_amd.synthetic import synth_clips
    # PyTorch-CPU convs stop scaling (and then regress) long before a 2-socket host's
    # full core count; 16 threads was the fastest setting measured on the GPU node.
    cores = min(os.cpu_count() or 1, int(os.environ.get('GRL_CPU_THREADS', '16')))
    torch.set_num_threads(cores)
    O.extract_features(sd, ssd, synth_clips(2, T, seed=1))        # warm-up (allocator, threads)
    nb, passes = B, 3
    clips = synth_clips(nb, T, seed=0)
    t0 = time.time()
    for _ in range(passes):
        O.extract_features(sd, ssd, clips)
    dt = time.time() - t0
    return {"value": round(nb * passes / dt, 3), "unit": "clip-features/sec", "cores": cores,
            "node_cores": os.cpu_count(), "kind": "port",
            "sample": "oracle.extract_features (torch CPU fp32, %d threads of the node's %d cores: the fastest "
                      "setting measured, PyTorch-CPU convs regress beyond it) on the %d clips of one "
                      "step, T=%d, %d timed passes (%.1f s)" % (cores, os.cpu_count() or 0, nb, T, passes, dt)}


def train_series(dev, math, steps=20, warmup=5, b=32, t=4, rank=0, world=1, dist=None, profile='default', graph=False,
                 roofline=False):
    """One SEQTrainer step (forward + 5-term loss + HIP backward + bucketed gradient all-reduce when a process
    group is up + SGD) on b x t synthetic pair-interleaved clips per rank.  Timed with a barrier + device sync on
    both sides, MAX over ranks.  Returns a dict (ms_per_step, and the gradient-sync bookkeeping)."""
    from grl_amd.reid import models
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.synthetic import synth_clips, synth_state_dict
    from grl_amd import train_engine
    from grl_amd import dist as grl_dist
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile=profile))
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
    cnn, siam, siamv = cnn.to(dev).train(), siam.to(dev).train(), siamv.to(dev).train()
    tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                    OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
    params = tr._all_params()
    # mars_train.py's optimizer (torch.optim.SGD, nesterov) in its single-kernel `fused` form: the update is one pass
    # over parameters / gradients / momentum buffers instead of torch's four foreach passes (0.77 -> ~0.3 ms per step)
    try:
        opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
    except (TypeError, RuntimeError):
        opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)
    clips = synth_clips(b, t, seed=rank).to(dev)
    pids = (torch.arange(b, device=dev) // 2 * 7 + rank * 131) % 625
    sync = grl_dist.GradSync(params) if grl_dist.is_distributed() else None   # (GRL_SYNC_FORCE=1: also in a world of one)
    waits = []                     # per timed step: [(bucket label, event, event)] -- read AFTER the timed region
    if sync is not None:
        sync.timing = True

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        loss, _, _, _ = tr._forward([clips], pids, 0, 0)
        opt.zero_grad()
        if sync is not None:
            sync.begin()
        loss.backward()
        if sync is not None:
            sync.finish()              # the launch stream waits here for whatever the backward did not cover;
            waits.append(sync._waits)  # HIP events around every bucket's wait -- nothing blocks the host (N = 1 and
        opt.step()                     # N > 1 steps keep the same host run-ahead: one measurement regime)
        return loss

    old = train_engine.set_math(math)
    try:
        if graph and sync is None:
            # the SAME step -- forward, loss block, backward on its three streams, fused SGD -- captured once into a HIP
            # graph and replayed (grl_amd/train_graph.py; bit-identical to the eager step, tested): no Python between
            # the ~1500 launches.  Matters where the step is host-bound (bf16 storage); single GPU only.
            from grl_amd.train_graph import GraphedTrainStep
            gstep = GraphedTrainStep(tr, opt, clips, pids, warmup=max(warmup, 2))
            step = lambda: gstep()[0]
        for _ in range(warmup):
            loss = step()
        if warmup > 0:
            # what the product's training loop does after its first step (reid/train/trainer.py: gc.freeze(), off with
            # GRL_GC_FREEZE=0): the long-lived objects leave the cyclic collector's sight -- its pauses are step time on
            # a host-bound step
            from grl_amd.reid.train.trainer import _freeze_collector_once
            _freeze_collector_once()
        barrier()
        del waits[:]
        # a HIP event on the launch stream after every step (read after the timed region: nothing blocks the host):
        # the spread of the step times says whether a series is stable (round 4: 6 steps after 3 warm-ups could not
        # tell a regression from allocator / clock-ramp jitter -- driver 58.8 ms against 53.5 here)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(steps):
            loss = step()
            marks[k + 1].record()
        t_issue = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps))
        assert bool(torch.isfinite(loss).all())
    finally:
        train_engine.set_math(old)
    dt = max_over_ranks(dist, dev, dt)
    ms = dt / steps * 1e3
    out = {"ms_per_step": round(ms, 2), "clips_per_sec": round(max(world, 1) * b / ms * 1e3, 1),
           "steps": steps, "warmup": warmup,
           "step_ms_on_stream": {"median": round(per_step[len(per_step) // 2], 2), "min": round(per_step[0], 2),
                                 "max": round(per_step[-1], 2)},
           "side_streams": side_streams_state(),
           "host": {"issue_ms_per_step": round(t_issue / steps * 1e3, 3), "launch_bound_frac": round(min(1.0, t_issue / dt), 3)}}
    if roofline:
        # per GPU: algorithmic FLOPs of this rank's step / the step time; the dominant kernels timed live (N = 1 only:
        # the timing pass runs extra steps, which at N > 1 would have to stay in lockstep over the collectives)
        kern = train_kernel_timing(tr, clips, pids, math) if (sync is None and not graph) else None
        pmc = {(32, 4, 'f32'): 'train_f32', (32, 4, 'bf16s'): 'train_bf16s', (64, 8, 'bf16s'): 'train_c3_bf16s'}.get((b, t, math))
        out["roofline"] = series_roofline(math, b, t, ms, train=True, kernel=kern, pmc=pmc,
                                          alg_bytes=kern["algorithmic_bytes_per_step"] if kern else None)
    if sync is not None:
        per_bucket = {}
        for w in waits:                                             # (the final barrier synchronised the device)
            for lab, a, b2 in w:
                per_bucket.setdefault(lab, []).append(a.elapsed_time(b2))
        exposed = [[lab, round(sum(v) / len(v), 3)] for lab, v in per_bucket.items()]
        out.update({"allreduce_bytes_per_step": 4 * sum(n for _, n in sync.launched),
                    "allreduce_buckets": [[lab, 4 * n] for lab, n in sync.launched],
                    "gradsync_collectives_per_step": sync.collectives, "stray_reductions": sync.stray,
                    "allreduce_exposed_ms": round(sum(v for _, v in exposed), 3),
                    "allreduce_exposed_ms_per_bucket": exposed,
                    "reduce_op": "AVG in the collective" if sync.avg_op else "SUM + scale pass"})
    return out


def side_streams_state():
    """which of the train step's side streams are switched on (environment: GRL_TRL_STREAMS / GRL_WGRAD_STREAM /
    GRL_HEAD_STREAMS): a series measured without them is a different regime (EXPERIMENTS.md, round 4: 7-9 %)"""
    from grl_amd import engine, train_engine
    from grl_amd.reid.train import trainer
    return {"trl": bool(engine.TRL_STREAMS), "wgrad": bool(train_engine.WGRAD_STREAM), "heads": bool(trainer.HEAD_STREAMS)}


def train_step_ms(dev, math, steps=20, warmup=5, b=32, t=4, graph=False):
    return train_series(dev, math, steps, warmup, b, t, graph=graph)["ms_per_step"]


def train_step_record(dev, math, steps=20, warmup=5, b=32, t=4):
    r = train_series(dev, math, steps, warmup, b, t, roofline=True)
    return {k: r[k] for k in ("ms_per_step", "clips_per_sec", "steps", "warmup", "step_ms_on_stream", "side_streams", "host", "roofline")}


def train_block(dev, rank, world, dist, backend):
    """BASELINE configs[3] per GPU: 64 clips x 4 frames (global P x K = 128 x 4 over 8 GPUs), SEQTrainer step with
    the gradient buckets all-reduced under the backward.  Reported on every line (N = 1: no exchange), so that one
    driver run at N = 1, 2, 4, 8 records the training series next to the eval headline."""
    out = {"workload": "BASELINE configs[3] per GPU: SEQTrainer step (fwd + 5-term loss + HIP bwd + bucketed "
                       "gradient all-reduce + SGD), 64 clips x 4 frames per GPU, global batch %d clips" % (64 * max(world, 1)),
           "n_gpus": max(world, 1), "dist_backend": backend if dist is not None else None,
           "rccl_ranks": dist.get_world_size() if (dist is not None and backend == 'nccl') else None,
           "rccl_version": rccl_version()}
    for m in ('f32', 'mixed'):
        release_cached_blocks()
        out[m] = train_series(dev, m, steps=12, warmup=4, b=64, t=4, rank=rank, world=world, dist=dist, roofline=True)
    return out


def rccl_version():
    """The RCCL build torch is linked against (torch.cuda.nccl.version() IS RCCL's on ROCm), e.g. '2.26.6'."""
    try:
        return '.'.join(str(x) for x in torch.cuda.nccl.version())
    except Exception:                                   # noqa: BLE001
        return None


def release_cached_blocks():
    """Between two series of one process, never inside a timed region: drop the previous series' model / optimizer
    and hand its cached blocks back to the driver, so that every series starts from the allocator state of a fresh
    process.  (Round 3 found a later series up to 40 % slower than in a fresh process -- bf16x3 train 54-64 ms vs
    39.3; the cause was a Tape <-> closure-list reference cycle in train_engine that kept every finished step's flat
    gradient buffer alive until a full garbage collection.  Fixed there; EXPERIMENTS.md.)"""
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def decode_block():
    """Frame decode next to the model: grl_jpeg_decode_batch (device, bit-identical to Pillow) on synthetic MARS-size
    JPEG frames (256 x 128, 4:2:0, quality 90) against Pillow on ONE host core (the reference's eval loaders run with 0
    workers, dataloader.py:77-79), and the frame rates the eval series consume.  Informational; bounded to ~3 s."""
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import decode_rate
        frames = decode_rate.make_frames(64)
        decode_rate._init(frames)
        decode_rate._decode_range((0, 32))
        t0 = time.perf_counter()
        decode_rate._decode_range((0, 512))
        host = 512 / (time.perf_counter() - t0)
        dev_r = decode_rate.device_rates(frames, batches=(128, 512))
        return {"host_pillow_frames_per_sec_one_core": round(host), "device": dev_r,
                "mean_jpeg_kb": round(sum(len(f) for f in frames) / len(frames) / 1024.0, 1),
                "consumers_frames_per_sec": "fp32 eval headline ~8.9 k, bf16-storage configs[2] ~53 k, bf16s train ~7.3 k",
                "profiles": "profiles/r06_decode_rate.json: Pillow with 1..128 worker processes on the node (5.8 k / worker, 88 k at 16)"}
    except Exception as e:                                  # noqa: BLE001 (informational block: never fails the line)
        return {"error": repr(e)[:300]}


def secondary_block(dev, cnn, siam, steps):
    """Informational series measured by the SAME default run (so the driver's record holds them too):
    BASELINE configs[2] (64 clips x 8 frames, bf16 storage), the train step (fp32 and the bf16x3
    datapath) and the MARS-size distance matrix of configs[4].  Never `value`."""
    from grl_amd import engine
    from grl_amd.synthetic import synth_clips, synth_eval_features
    out = {}
    release_cached_blocks()
    c3 = synth_clips(64, 8, seed=0).to(dev)
    with engine.math_mode('bf16s'):
        for _ in range(3):
            engine.extract_features(cnn, siam, c3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            engine.extract_features(cnn, siam, c3)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    out["configs[2] bf16s, 64 clips x 8 frames"] = {
        "clip_features_per_sec": round(64 / ms * 1e3, 1), "ms_per_step": round(ms, 3), "frames_per_sec": round(512 / ms * 1e3),
        "roofline": (lambda k: series_roofline('bf16s', 64, 8, ms, kernel=k, pmc='eval_c3_bf16s',
                                               alg_bytes=k["algorithmic_bytes_per_step"]))(eval_kernel_timing(cnn, siam, c3, 'bf16s'))}
    del c3

    def fresh(m, **kw):
        release_cached_blocks()
        return train_step_record(dev, m, **kw)
    out["train step, B x T = 32 x 4 (fwd + loss + bwd + SGD)"] = {m: fresh(m) for m in ('f32', 'mixed', 'bf16x3', 'bf16s')}
    # (`--mode train --graph` replays the same step from a HIP graph -- grl_amd.train_graph, bit-identical; measured
    # slower than the eager step on this stack, EXPERIMENTS.md, so it is not part of the default line)
    v = fresh('bf16s', b=64, t=8, steps=12, warmup=4)
    release_cached_blocks()
    v["frames_per_sec"] = round(512 / v["ms_per_step"] * 1e3)
    out["configs[2] as a training batch: P x K = 16 x 4, T = 8, bf16 storage (fwd + loss + bwd + SGD)"] = v
    out["input pipeline: frame decode (SURVEY 8(f) rank 4; video_loader.py:124-141)"] = decode_block()
    qf, gf = synth_eval_features(1980, 11310, seed=1, noise=6.0)[:2]
    qd, gd = qf.to(dev), gf.to(dev)
    # 15 warm-ups / 20 timed launches (round 6): generating the 13290 feature rows on the host above leaves the GPU idle for
    # about a second, and the first ~10 launches after an idle gap run on the clock ramp -- 2.7 -> 2.3 ms per launch,
    # 2.14 from the 15th on (tools/distmat_gap.py, profiles/r06_distmat_gap.txt).  Rounds 3-5 timed 5 launches after 2
    # warm-ups here (2.40-2.42 ms) against 20 after 5 in `--mode distmat` (2.17): the whole "gap" between the two.
    for _ in range(15):
        d = engine.cosin_dist(qd, gd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        d = engine.cosin_dist(qd, gd)
    torch.cuda.synchronize()
    dms = (time.perf_counter() - t0) / 20 * 1e3
    engine.rank_rows(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        engine.rank_rows(d)
    torch.cuda.synchronize()
    out["configs[4] distance matrix 1980 x 11310 x 6144 (fp32)"] = {
        "ms": round(dms, 3), "tflops": round(2.0 * 1980 * 11310 * 6144 / dms / 1e9, 1),
        "row_argsort_ms": round((time.perf_counter() - t0) / 3 * 1e3, 3),
        "roofline": {"bound": "mfma", "achieved": round(2.0 * 1980 * 11310 * 6144 / dms / 1e9, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(2.0 * 1980 * 11310 * 6144 / dms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                     "traffic": None, "kernel": "gemm_f32_kernel<128,128> NEGDOT epilogue, one launch"}}
    return out


def train_bench(args, dev, dist, rank, world, backend):
    """Secondary series: train clips/sec.  One step = SEQTrainer's
    forward (train-mode BN) + reference loss composition + HIP backward + the bucketed
    gradient all-reduce when world > 1 + SGD(nesterov) step, on B x T synthetic
    pair-interleaved clips per rank (mars_train.py -b 32 --seq_len 4)."""
    if args.math not in ('f32', 'mixed', 'bf16x3', 'bf16', 'bf16s'):
        raise SystemExit('--mode train supports --math f32 | mixed | bf16x3 | bf16 | bf16s')
    r = train_series(dev, args.math, steps=args.steps, warmup=args.warmup, b=B, t=T, rank=rank, world=world, dist=dist,
                     graph=args.graph, roofline=True)
    if rank == 0:
        n = max(world, 1)
        line = {
            "metric": "train clips/sec", "value": r["clips_per_sec"], "unit": "clips/sec", "n_gpus": n,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "mixed": "f32 forward (exact), split-bf16 products in the backward GEMMs (dgrad + wgrad), f32 accumulate / storage",
                      "bf16x3": "bf16x3 products (fwd + dgrad GEMMs), f32 accumulate / storage / wgrad",
                      "bf16": "bf16 products (fwd + dgrad GEMMs), f32 accumulate / storage / wgrad",
                      "bf16s": "bf16 storage (activations, saved tensors, activation gradients), bf16 MFMA, f32 accumulate / statistics / parameter gradients"}[args.math],
            "data": "synthetic",
            "config": {"workload": "GRL train step (fwd + loss + bwd + allreduce + SGD), B x T = %d x %d per GPU" % (B, T),
                       "clips_per_gpu": B, "seq_len": T, "math": args.math, "hip_graph_replay": bool(args.graph and dist is None),
                       "parallelism": "dp%d (4 gradient buckets all-reduced over RCCL under the backward)" % n},
            "dist_backend": backend if dist is not None else None,
            "rccl_ranks": dist.get_world_size() if (dist is not None and backend == 'nccl') else None,
            "roofline": r["roofline"],
            "gradsync": {k: v for k, v in r.items() if k not in ("ms_per_step", "clips_per_sec", "roofline")},
            "end_to_end_tflops": round(r["clips_per_sec"] / n * 173.8 / 1e3, 2)}
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def distmat_bench(args, dev, rank):
    """Third series: the evaluator's query x gallery matrix at MARS size
    (BASELINE configs[4]): -Q.G^T, Q [1980,6144], G [11310,6144], exact fp32 MFMA."""
    from grl_amd import engine
    from grl_amd.synthetic import synth_eval_features
    qf, gf, qp, qc, gp, gc = synth_eval_features(1980, 11310, seed=1, noise=6.0)
    qd, gd = qf.to(dev), gf.to(dev)
    for _ in range(max(args.warmup, 15)):          # (the first ~10 launches after an idle gap run on the clock ramp: secondary_block)
        d = engine.cosin_dist(qd, gd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d = engine.cosin_dist(qd, gd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    idx = engine.rank_rows(d)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        idx = engine.rank_rows(d)
    torch.cuda.synchronize()
    sort_ms = (time.perf_counter() - t1) / args.steps * 1e3
    import numpy as np
    dh = d.cpu().numpy()
    t2 = time.perf_counter()
    ref_idx = np.argsort(dh[:256], axis=1, kind='stable')
    host_s = (time.perf_counter() - t2) * 1980 / 256
    assert np.array_equal(idx[:256].cpu().numpy(), ref_idx)
    # the rest of the evaluator on the device: per-query CMC / AP, and k-reciprocal re-ranking
    from grl_amd.reid.evaluator.rerank import re_ranking

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3, r
    metrics_ms, (cmc, mAP) = timed(lambda: engine.rank_metrics(idx, qp, gp, qc, gc))
    # (on Euclidean q-g distances, what the algorithm is defined on; the reference's evaluator
    # hands it the NEGATED-cosine matrix, attevaluator.py:150-155 -- ATTEvaluator keeps that call)
    dqq, dgg = engine.pairwise_distance_tensor(qd, qd), engine.pairwise_distance_tensor(gd, gd)
    dqg = engine.pairwise_distance_tensor(qd, gd)
    rerank_ms, rr = timed(lambda: re_ranking(dqg, dqq, dgg), n=2)
    cmc_rr, map_rr = engine.rank_metrics(engine.rank_rows(rr), qp, gp, qc, gc)
    del dqq, dgg, dqg, rr
    flops = 2.0 * 1980 * 11310 * 6144
    if rank == 0:
        emit(({"metric": "distance-matrix ms (1980x11310x6144)", "value": round(dt * 1e3, 3), "unit": "ms",
                          "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
                          "higher_is_better": False, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "cosin_dist at MARS size, BASELINE configs[4]",
                                     "row_argsort_ms_gpu": round(sort_ms, 3),
                                     "row_argsort_s_numpy_1core_extrapolated": round(host_s, 2),
                                     "cmc_ap_ms_gpu": round(metrics_ms, 3),
                                     "rerank_ms_gpu (13290^2 k-reciprocal, incl. its argsort)": round(rerank_ms, 2),
                                     "synthetic mAP / Rank-1": [round(mAP, 4), round(float(cmc[0]), 4)],
                                     "after re-ranking (Euclidean q-g)": [round(map_rr, 4), round(float(cmc_rr[0]), 4)]},
                          "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 2),
                                       "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(flops / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None}}))


from grl_amd.rank_guard import EXIT_TRAIN_BLOCK_FAILED, TrainBlockGuard   # noqa: E402 (no torch, no GPU: files + a thread)


def run_guarded(guard, world, fn):
    """fn() under the guard.  world == 1: exceptions propagate (nothing to hang).  world > 1: an exception on this
    rank drops the step's gradient-sync state WITHOUT issuing a collective (the peers are elsewhere in the sequence),
    skips every later barrier, and leaves through the guard."""
    if world == 1:
        return fn()
    guard.start()
    try:
        r = fn()
    except BaseException as e:                      # noqa: B902 (SystemExit from an assert helper included)
        with contextlib.suppress(Exception):
            from grl_amd import train_engine
            train_engine.set_grad_sync(None)
        guard.leave("train block raised on rank %d: %s" % (guard.rank, repr(e)[:300]))
        raise
    guard.finished()
    return r


def preflight(dist, rank, world, dev, backend):
    """First contact of the process group (N > 1), BEFORE anything is measured: a 1 MB all-reduce whose result is
    checked, timed twice (the first includes the communicator set-up), and what every rank sits on -- so that a hang or
    a wrong topology at the first real multi-GPU run is attributable from the one JSON line.  Replaces nothing in the
    reference (nn.DataParallel, mars_train.py:80-82, has no such step); `dev` None = CPU tensors (gloo dry run)."""
    x = torch.ones(1 << 18, dtype=torch.float32, device=dev if dev is not None else 'cpu')
    times = []
    for _ in range(2):
        if dev is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = x.clone()
        dist.all_reduce(y)
        if dev is not None:
            torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        if not bool((y == float(world)).all()):
            raise RuntimeError('preflight all-reduce returned %r, expected %d on every element' % (float(y[0]), world))
    me = {"rank": rank, "pid": os.getpid(), "device": None}
    if dev is not None:
        pr = torch.cuda.get_device_properties(dev)
        me.update({"device": dev.index, "name": pr.name, "cus": pr.multi_processor_count,
                   "pci_bus_id": getattr(pr, 'pci_bus_id', None), "hbm_gib": round(pr.total_memory / 2 ** 30, 1)})
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    return {"backend": backend, "ranks": world, "rccl_version": rccl_version() if backend == 'nccl' else None,
            "allreduce_1MB_ms": {"first (with communicator set-up)": round(times[0], 3), "second": round(times[1], 3)},
            "ipc": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                    "NCCL_env": {k: v for k, v in os.environ.items() if k.startswith(('NCCL_', 'RCCL_'))}},
            "distinct_devices": len({(r["device"], r.get("pci_bus_id")) for r in ranks}) if dev is not None else None,
            "per_rank": ranks}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process group
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) and return its exit code.  This
    parent has not touched the GPU and never does (no HIP call, no exec of a GPU-initialised process); the child's
    rank 0 prints the JSON line to the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    env['GRL_BENCH_NONCE'] = '%d_%d' % (os.getpid(), int(time.time() * 1e3))      # (TrainBlockGuard's file key)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


_REAL_STDOUT = [None]


def quiet_stdout():
    """From here on file descriptor 1 is the process's stderr: RCCL prints a version banner on stdout when a communicator
    comes up (seen: five lines before the result), and the contract is ONE JSON line there.  `emit` writes to the real one."""
    if _REAL_STDOUT[0] is None:
        sys.stdout.flush()
        _REAL_STDOUT[0] = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + '\n').encode()
    if _REAL_STDOUT[0] is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT[0], line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--math', default='f32', choices=['f32', 'mixed', 'bf16x3', 'bf16', 'bf16s'],
                    help="multiplier datapath of the conv GEMMs for the headline `value` "
                         "(default: exact fp32 MFMA = BASELINE configs[1])")
    ap.add_argument('--no-alt', action='store_true', help='skip the secondary bf16x3 / bf16 / bf16s measurements')
    ap.add_argument('--clips', type=int, default=B, help='clips per GPU per step (default 32 = BASELINE configs[1]; '
                                                         'configs[2] is --clips 64 --seq-len 8 --math bf16s)')
    ap.add_argument('--seq-len', type=int, default=T, help='frames per clip (default 4)')
    ap.add_argument('--mode', default='eval', choices=['eval', 'train', 'distmat'],
                    help="eval (default): the headline clip-features/sec; train: secondary series, one "
                         "SEQTrainer step (forward + 5-term loss + HIP backward + grad all-reduce + SGD)")
    ap.add_argument('--no-train-block', action='store_true', help='skip the configs[3] `train` block of the eval line')
    ap.add_argument('--no-preflight', action='store_true',
                    help='N > 1: skip the first-contact step (a checked 1 MB all-reduce + every rank\'s device) that '
                         'otherwise runs before anything is measured and is reported as `preflight` in the line')
    ap.add_argument('--graph', action='store_true', help='--mode train: replay the step from a HIP graph (single GPU)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher check (CPU tests): every rank joins a gloo group, the ranks are counted with an '
                         'all-reduce, rank 0 prints {"dry_run": true, "world_size": N}; no GPU is touched')
    args = ap.parse_args()
    globals()['B'], globals()['T'] = args.clips, args.seq_len

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != max(args.gpus, 1):
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.dry_run:
        import torch.distributed as dist
        n = torch.ones(1)
        if world > 1:
            dist.init_process_group('gloo')
            dist.all_reduce(n)
        line = {"dry_run": True, "world_size": int(n.item()), "n_gpus": args.gpus}
        if world > 1 and not args.no_preflight:
            line["preflight"] = preflight(dist, rank, world, None, 'gloo')
        fault = os.environ.get('GRL_BENCH_DRY_TRAIN')        # tests: 'ok' | 'hang' | 'raise' (on rank 1)
        if fault and world > 1:
            def fake_train_block():
                if rank == 1 and fault == 'hang':
                    time.sleep(3600)
                if rank == 1 and fault == 'raise':
                    raise RuntimeError('injected failure')
                dist.barrier()
                return {"ok": True}
            guard = TrainBlockGuard(rank, world, float(os.environ.get('GRL_BENCH_TRAIN_TIMEOUT', '300')),
                                    lambda reason: print(json.dumps(dict(line, train={"error": reason})), flush=True))
            line["train"] = run_guarded(guard, world, fake_train_block)
        if world > 1:
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(line))
        return
    quiet_stdout()
    dist = None
    # GRL_DIST_BACKEND=gloo + GRL_SINGLE_DEVICE=1: functional test of the N>1 path on a box with
    # one GPU (every rank on cuda:0, collectives through gloo); the real runs use nccl = RCCL.
    backend = os.environ.get('GRL_DIST_BACKEND', 'nccl')
    if os.environ.get('GRL_SINGLE_DEVICE'):
        local = 0
    if world > 1 or (os.environ.get('GRL_SYNC_FORCE') == '1' and 'RANK' in os.environ):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    else:
        local = 0
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local)
    pre = None
    if dist is not None and world > 1 and not args.no_preflight:
        # under the same guard as the train block: a hang in RCCL's first collective leaves ONE line and a red exit code
        def pre_error(reason):
            emit({"metric": "clip-features/sec", "value": None, "n_gpus": world, "preflight": {"error": reason}})
        pguard = TrainBlockGuard(rank, world, float(os.environ.get('GRL_BENCH_PREFLIGHT_TIMEOUT', '120')), pre_error, tag='pre')
        pre = run_guarded(pguard, world, lambda: preflight(dist, rank, world, dev, backend))

    from grl_amd import engine, _lib
    from grl_amd.synthetic import synth_clips
    _lib.load()
    cnn, siam, sd, ssd = build_models(dev)
    clips = synth_clips(B, T, seed=rank).to(dev)
    engine.set_math('f32' if args.math == 'mixed' else args.math)     # ('mixed' is a training datapath)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == 'train':
        del cnn, siam
        return train_bench(args, dev, dist, rank, world, backend)
    if args.mode == 'distmat':
        return distmat_bench(args, dev, rank)

    for _ in range(args.warmup):
        feat = engine.extract_features(cnn, siam, clips)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feat = engine.extract_features(cnn, siam, clips)
    t_issue = time.perf_counter() - t0          # host time to ISSUE the steps (nothing in a step waits for the device)
    barrier()
    dt = max_over_ranks(dist, dev, time.perf_counter() - t0)
    assert bool(torch.isfinite(feat).all())
    train = None
    want_train = not args.no_train_block and (B, T, args.math) == (32, 4, 'f32')
    out, guard = None, None
    if rank == 0:
        n = max(world, 1)
        value = n * B * args.steps / dt
        stages = {}
        flops, gemm_ms, launches = gemm_roofline(cnn, siam, clips, stages=stages)
        achieved = flops / (gemm_ms * 1e-3) / 1e12
        # dense MFMA peak of the datapath (MI355X_MICROARCH.md): fp32 157.3, bf16 2500; bf16x3
        # issues three bf16 MFMAs per product, so its fp32-equivalent peak is 2500/3
        peak = {'f32': PEAK_FP32_MFMA_TFLOPS, 'bf16': 2500.0, 'bf16s': 2500.0, 'bf16x3': 2500.0 / 3}[args.math]
        # counter traffic of the SAME kernels (GEMM + fused-tail launches of one step): only from a profile of this build
        pmc_series = {(32, 4, 'f32'): 'eval_f32', (64, 8, 'bf16s'): 'eval_c3_bf16s'}.get((B, T, args.math))
        prec, pwhy = pmc_record(pmc_series) if pmc_series else (None, 'no profiled series for this configuration')
        traffic = prec['dominant']['hbm_bytes_per_step'] if prec else None
        alg_bytes = sum(v["algorithmic_gb"] for v in stages.values()) * 1e9
        out = {
            "metric": "clip-features/sec", "value": round(value, 2), "unit": "clip-features/sec",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "mixed": "f32 forward (exact), split-bf16 products in the backward GEMMs (dgrad + wgrad), f32 accumulate / storage",
                      "bf16x3": "bf16x3 (split operands, f32 accumulate)",
                      "bf16": "bf16 operands, f32 accumulate", "bf16s": "bf16 storage, f32 accumulate"}[args.math],
            "data": "synthetic",
            "config": {"workload": "GRL eval clip features (ResNet-50 s1 trunk + GCE + TRL + "
                                   "Siamese attention -> 6144-d), BASELINE configs[%s]" % (
                                       '1' if (B, T, args.math) == (32, 4, 'f32') else
                                       '2' if (B, T) == (64, 8) and args.math in ('bf16', 'bf16s') else '1 (variant)'),
                       "clips_per_gpu": B, "seq_len": T, "frame": "256x128",
                       "parallelism": "replicas x%d (no collective)" % n},
            "roofline": {"bound": "mfma" if flops / (peak * 1e12) >= (traffic or alg_bytes) / (PEAK_HBM_GBS * 1e9) else "hbm",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_note": ("profiles/%s_pmc_%s.json, taken on THIS build (fingerprint checked): rocprofv3 FETCH_SIZE x2 + "
                                          "WRITE_SIZE, separate passes, summed over the same kernels' launches of ONE step "
                                          "like the GFLOP figure; %.2f x their algorithmic bytes; MFMA busy %s" % (
                                              PMC_ROUND, pmc_series, traffic / alg_bytes, prec['dominant'].get('mfma_busy_frac')))
                         if prec else "null: %s" % pwhy,
                         "algorithmic_gb_per_step": round(alg_bytes / 1e9, 2),
                         "kernel": "gemm_f32_kernel + fused bottleneck tails (%s MFMA implicit-GEMM conv), %d launches/step, "
                                   "%.3f ms/step, %.1f algorithmic GFLOP/step" % (
                                       'fp32' if args.math == 'f32' else 'bf16', launches, gemm_ms, flops / 1e9),
                         "stages (one stream; mfma_util = algorithmic FLOPs / stage wall time / MFMA peak; hbm_gbs = algorithmic bytes / stage wall time)": stages},
            "end_to_end_tflops": round(value / n * GFLOP_PER_FRAME * T / 1e3, 2),
            # strong-scaling readiness: how much of a step's wall time the host needs just to issue its
            # launches.  >= 1: the step is launch-bound (the GPU waits for the host) -- what `--clips 4` (configs[1] split
            # over 8 GPUs) is about.
            "host": {"issue_ms_per_step": round(t_issue / args.steps * 1e3, 3),
                     "launch_bound_frac": round(min(1.0, t_issue / dt), 3)},
        }
        out["config"]["math"] = args.math
        if dist is not None:
            # what the process group itself reports (the eval data path has no collective; the timed region is
            # bracketed by dist.barrier and the per-rank times are MAX-reduced through it)
            out["dist_backend"] = dist.get_backend()
            out["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == 'nccl' else None
            out["world_size"] = dist.get_world_size()
            if pre is not None:
                out["preflight"] = pre
    if want_train:
        # Every rank: the training step has collectives (RCCL gradient all-reduce at N > 1).  The headline line must
        # not depend on them: at N > 1 a guard prints it with `train: {"error"}` if the block hangs or raises on any
        # rank -- and every rank then exits NON-ZERO, so the launcher's return code is red (TrainBlockGuard).
        def error_line(reason):
            out["train"] = {"error": reason}
            emit(out)
        guard = TrainBlockGuard(rank, world, float(os.environ.get('GRL_BENCH_TRAIN_TIMEOUT', '300')), error_line)
        train = run_guarded(guard, world, lambda: train_block(dev, rank, world, dist, backend))
    if rank == 0:
        if train is not None:
            out["train"] = train
        if n == 1 and not args.no_alt:
            # secondary, informational: the opt-in bf16 multiplier datapaths on the SAME
            # workload, with their deviation from the exact-fp32 features measured live
            alt = {}
            for mode in ('bf16x3', 'bf16', 'bf16s'):
                if mode == args.math:
                    continue
                with engine.math_mode(mode):
                    for _ in range(3):
                        f2 = engine.extract_features(cnn, siam, clips)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        f2 = engine.extract_features(cnn, siam, clips)
                    torch.cuda.synchronize()
                    d2 = time.perf_counter() - t1
                alt[mode] = {"value": round(B * args.steps / d2, 2), "ms_per_step": round(d2 / args.steps * 1e3, 3),
                             "max_rel_dev_vs_headline_features": float(
                                 ((f2 - feat).abs().max() / feat.abs().max()).item())}
            out["alt_math"] = alt
        if n == 1 and not args.no_alt:
            # informational (never `value`): the same step fed from HOST-resident pinned batches through
            # engine.DevicePrefetcher (next batch's PCIe copy overlapped on a side stream)
            from grl_amd.synthetic import synth_clips as _sc
            host = {}
            for name, batch in (("f32", _sc(B, T, seed=0).pin_memory()), ("u8", _sc(B, T, seed=0, raw=True).pin_memory())):
                def loader(k):
                    for _ in range(k):
                        yield batch, None, None
                for d_, _, _ in engine.DevicePrefetcher(loader(3), dev):
                    engine.extract_features(cnn, siam, d_)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for d_, _, _ in engine.DevicePrefetcher(loader(args.steps), dev):
                    engine.extract_features(cnn, siam, d_)
                torch.cuda.synchronize()
                host[name] = round(B * args.steps / (time.perf_counter() - t1), 2)
            try:
                # ... and from COMPRESSED frames: B x T JPEG byte strings per step (256 x 128, 4:2:0, quality 90), header parse on
                # the host, bytes over PCIe, grl_jpeg_decode_batch on the prefetch streams next to the previous step's compute
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                import decode_rate
                from grl_amd.reid.data.jpeg import JpegBatch
                jf = decode_rate.make_frames(B * T)
                jb = JpegBatch(jf, (B, T))

                def jloader(k):
                    for _ in range(k):
                        yield jb, None, None
                for d_, _, _ in engine.DevicePrefetcher(jloader(3), dev):
                    engine.extract_features(cnn, siam, d_)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for d_, _, _ in engine.DevicePrefetcher(jloader(args.steps), dev):
                    engine.extract_features(cnn, siam, d_)
                torch.cuda.synchronize()
                host["jpeg (device decode)"] = round(B * args.steps / (time.perf_counter() - t1), 2)
            except Exception as e:                              # noqa: BLE001 (informational)
                host["jpeg (device decode)"] = 'error: %r' % (e,)
            out["host_resident_inputs"] = {"clip_features_per_sec": host,
                                           "note": "pinned host batches, H2D overlapped; uint8 is normalised in the stem; jpeg: compressed "
                                                   "frames decoded on the device (bit-identical to Pillow), two batches in flight"}
        if n == 1 and not args.no_alt and (B, T, args.math) == (32, 4, 'f32'):
            out["secondary"] = secondary_block(dev, cnn, siam, args.steps)
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, ssd)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
