#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun; every pass under `timeout`).
#   bash tools/profile_round.sh <out_dir_under_gpurun_out>
# Passes per workload: kernel trace + stats, then three separate --pmc passes (MFMA busy, FETCH_SIZE, WRITE_SIZE) --
# never combined with trace domains other than the kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
python3 $R/tools/fingerprint.py > $O/fingerprint.json      # the build these passes run on (bench.py checks it before quoting them)
cd /tmp && export TMPDIR=/tmp
# per-kernel durations and counters are only additive when kernels do not overlap: one stream for the profile passes
export GRL_TRL_STREAMS=0 GRL_WGRAD_STREAM=0
pass() {  # name, rocprof args..., -- program args        (ONLY=ev|tr|c3 restricts the workloads)
    local name=$1; shift
    if [ -n "${ONLY:-}" ] && [[ $name != ${ONLY}* ]]; then return; fi
    timeout 400 rocprofv3 "$@" > $O/$name.log 2>&1
    echo "$name rc=$?"
}
EV="--no-alt --no-cpu-baseline --no-train-block --steps 12 --warmup 4"
pass ev_stats --kernel-trace --stats --output-format csv -d $O/ev_stats -- python3 $R/bench.py $EV
pass ev_mfma  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/ev_mfma -- python3 $R/bench.py $EV
pass ev_fetch --pmc FETCH_SIZE --output-format csv -d $O/ev_fetch -- python3 $R/bench.py $EV
pass ev_write --pmc WRITE_SIZE --output-format csv -d $O/ev_write -- python3 $R/bench.py $EV
TR="--mode train --steps 4 --warmup 2"
pass tr_stats --kernel-trace --stats --output-format csv -d $O/tr_stats -- python3 $R/bench.py $TR
pass trm_stats --kernel-trace --stats --output-format csv -d $O/trm_stats -- python3 $R/bench.py $TR --math mixed
pass tr_mfma  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/tr_mfma -- python3 $R/bench.py $TR
pass tr_fetch --pmc FETCH_SIZE --output-format csv -d $O/tr_fetch -- python3 $R/bench.py $TR
pass tr_write --pmc WRITE_SIZE --output-format csv -d $O/tr_write -- python3 $R/bench.py $TR
# bf16-storage training (round 3): B x T = 32 x 4 and BASELINE configs[2] as a training batch (64 x 8)
TB="--mode train --math bf16s --steps 4 --warmup 2"
pass trb_stats --kernel-trace --stats --output-format csv -d $O/trb_stats -- python3 $R/bench.py $TB
pass trb_mfma  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/trb_mfma -- python3 $R/bench.py $TB
pass trb_fetch --pmc FETCH_SIZE --output-format csv -d $O/trb_fetch -- python3 $R/bench.py $TB
pass trb_write --pmc WRITE_SIZE --output-format csv -d $O/trb_write -- python3 $R/bench.py $TB
TB2="--mode train --math bf16s --clips 64 --seq-len 8 --steps 3 --warmup 2"
pass trc_stats --kernel-trace --stats --output-format csv -d $O/trc_stats -- python3 $R/bench.py $TB2
pass trc_mfma  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/trc_mfma -- python3 $R/bench.py $TB2
pass trc_fetch --pmc FETCH_SIZE --output-format csv -d $O/trc_fetch -- python3 $R/bench.py $TB2
pass trc_write --pmc WRITE_SIZE --output-format csv -d $O/trc_write -- python3 $R/bench.py $TB2
C3="--math bf16s --clips 64 --seq-len 8 --no-alt --no-cpu-baseline --no-train-block --steps 8 --warmup 3"
pass c3_stats --kernel-trace --stats --output-format csv -d $O/c3_stats -- python3 $R/bench.py $C3
pass c3_mfma  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c3_mfma -- python3 $R/bench.py $C3
pass c3_fetch --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -- python3 $R/bench.py $C3
pass c3_write --pmc WRITE_SIZE --output-format csv -d $O/c3_write -- python3 $R/bench.py $C3
# keep only what the summaries need (the merge-back limit is 64 MiB)
find $O -name '*agent_info.csv' -delete
du -sh $O
