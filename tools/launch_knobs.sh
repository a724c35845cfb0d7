#!/bin/bash
# Eager launches under the CLR / ROCr runtime knobs (round 6): does any setting shorten the inter-kernel gap of the
# ~1200-launch bf16-storage train step (GPU side ~17.4 ms: 15.2 ms of main-stream kernels + ~2 us per launch) or of the
# eval steps?   bash tools/launch_knobs.sh > gpurun_out/launch_knobs.txt      (knob names: strings libamdhip64.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # label, env..., -- args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    local out=$(env "${envs[@]}" timeout 200 python3 $R/bench.py "$@" 2>/dev/null | python3 -c "
import sys,json
try:
    b=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=(b.get('gradsync') or {}).get('host') or b.get('host') or {}
    print(b['ms_per_step'], 'issue', h.get('issue_ms_per_step'))
except Exception as e:
    print('FAILED', type(e).__name__)")
    echo "$label | $* | $out"
}
knobs() {
  run "default" X=1 -- "$@"
  run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0 -- "$@"
  run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1 -- "$@"
  run "DEBUG_HIP_KERNARG_COPY_OPT=0" DEBUG_HIP_KERNARG_COPY_OPT=0 -- "$@"
  run "ROC_SYSTEM_SCOPE_SIGNAL=0" ROC_SYSTEM_SCOPE_SIGNAL=0 -- "$@"
  run "ROC_ACTIVE_WAIT_TIMEOUT=0" ROC_ACTIVE_WAIT_TIMEOUT=0 -- "$@"
  run "ROC_ACTIVE_WAIT_TIMEOUT=1000" ROC_ACTIVE_WAIT_TIMEOUT=1000 -- "$@"
  run "DEBUG_HIP_DYNAMIC_QUEUES=0" DEBUG_HIP_DYNAMIC_QUEUES=0 -- "$@"
  run "GPU_MAX_HW_QUEUES=2" GPU_MAX_HW_QUEUES=2 -- "$@"
  run "GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8 -- "$@"
  run "ROC_AQL_QUEUE_SIZE=65536" ROC_AQL_QUEUE_SIZE=65536 -- "$@"
  run "default (again)" X=1 -- "$@"
}
echo "== train 32 x 4 bf16s"; knobs --mode train --math bf16s
echo "== eval 32 x 4 f32";    knobs --no-alt --no-train-block --no-cpu-baseline
echo "== eval 64 x 8 bf16s";  knobs --clips 64 --seq-len 8 --math bf16s --no-alt --no-train-block --no-cpu-baseline
