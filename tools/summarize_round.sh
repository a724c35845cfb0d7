#!/bin/bash
# Turns the passes of tools/profile_round.sh (merged back under gpurun_out/<dir>) into the files kept under profiles/:
#   bash tools/summarize_round.sh <dir_under_gpurun_out> r06
# Per workload: <round>_<series>_kernel_stats.csv (rocprofv3 --stats), <round>_<series>_pmc.md (per-kernel table) and
# <round>_pmc_<series>.json (the totals bench.py quotes as `roofline.traffic` -- with the fingerprint of the library /
# sources the passes ran on, written by the profile run itself into <dir>/fingerprint.json).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/gpurun_out/$1
TAG=${2:-r06}
P=$R/profiles
one() {   # prefix series steps title
    local pre=$1 ser=$2 steps=$3 title=$4
    [ -d $D/${pre}_mfma ] || { echo "skip $ser (no passes)"; return; }
    local st=$(ls $D/${pre}_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$st" ] && cp $st $P/${TAG}_${ser}_kernel_stats.csv
    python3 $R/tools/summarize_pmc.py "$title" $steps $D/${pre}_mfma $D/${pre}_fetch $D/${pre}_write \
        --json $P/${TAG}_pmc_${ser}.json --series $ser > $P/${TAG}_${ser}_pmc.md || echo "summarize $ser failed"
    # the fingerprint of the build the PASSES ran on (not of whatever is in the tree when this script runs)
    if [ -f $D/fingerprint.json ]; then
        python3 - $P/${TAG}_pmc_${ser}.json $D/fingerprint.json <<'PY'
import json, sys
rec = json.load(open(sys.argv[1])); rec.update(json.load(open(sys.argv[2]))); json.dump(rec, open(sys.argv[1], 'w'), indent=1)
PY
    fi
    echo "$ser done"
}
# steps: 0 = counted from the pass itself (dispatches of the forward stem kernel: one per step, warm-up and live-timing steps included)
one ev eval_f32 ${EV_STEPS:-0} "eval B x T = 32 x 4, exact fp32 (BASELINE configs[1])"
one c3 eval_c3_bf16s ${C3_STEPS:-0} "eval 64 x 8, bf16 storage (BASELINE configs[2])"
one tr train_f32 ${TR_STEPS:-0} "train step 32 x 4, fp32"
one trb train_bf16s ${TRB_STEPS:-0} "train step 32 x 4, bf16 storage"
one trc train_c3_bf16s ${TRC_STEPS:-0} "train step 64 x 8, bf16 storage (configs[2] as a training batch)"
