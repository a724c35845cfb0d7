#!/bin/bash
# Eval step fed from COMPRESSED frames (bench.py host_resident_inputs "jpeg"): does the 10 ms decode kernel overlap the
# compute?  A = the library as built, B = tools/_ko/libgrl_hip_noprio.so (GRL_GEMM_SETPRIO=0).   bash tools/jpegfeed_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for E in X=1 GRL_HIP_LIB=$R/tools/_ko/libgrl_hip_noprio.so; do
  for A in "" "--math bf16s --clips 64 --seq-len 8"; do
    r=$(env $E python bench.py --no-cpu-baseline --no-train-block $A 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'], b['host_resident_inputs']['clip_features_per_sec'])")
    echo "$E $A : $r"
  done
done
done
