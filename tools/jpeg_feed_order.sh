#!/bin/bash
# tools/jpeg_feed_order.py over priorities x orders x series.   bash tools/jpeg_feed_order.sh > gpurun_out/jpeg_feed_order.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
for P in 0 -1; do
  for F in model prefetch; do
    for A in "--mode train --math bf16s" "--mode train --math f32" "--mode eval --math f32" "--mode eval --math bf16s --clips 64 --seq-len 8"; do
      GRL_PREFETCH_PRIORITY=$P timeout 300 python3 $R/tools/jpeg_feed_order.py $A --first $F 2>/dev/null | tail -1
    done
  done
done
