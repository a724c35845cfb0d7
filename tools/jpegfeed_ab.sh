#!/bin/bash
# Eval step fed from COMPRESSED frames (bench.py host_resident_inputs "jpeg"): does the 10 ms decode kernel overlap the
# compute?  Streams of one priority share GPU_MAX_HW_QUEUES in-order hardware queues.   bash tools/jpegfeed_ab.sh
for E in GRL_PREFETCH_PRIORITY=-1 GRL_PREFETCH_PRIORITY=0 "GRL_PREFETCH_PRIORITY=0 GPU_MAX_HW_QUEUES=8" "GRL_PREFETCH_PRIORITY=-1 GPU_MAX_HW_QUEUES=8"; do
  for A in "" "--math bf16s --clips 64 --seq-len 8"; do
    r=$(env $E python bench.py --no-cpu-baseline --no-train-block $A 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['host_resident_inputs']['clip_features_per_sec'])")
    echo "$E $A : $r"
  done
done
