#!/bin/bash
# HIP-graph replay of the train step under the CLR graph knobs (round 6): does any setting keep the side streams'
# concurrency?  Eager vs replay, bf16s and f32, B x T = 32 x 4.   bash tools/graph_knobs.sh > gpurun_out/graph_knobs.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # label, env..., -- args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    local out=$(env "${envs[@]}" python3 $R/bench.py --mode train "$@" 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['gradsync']['host'])")
    echo "$label | $* | $out"
}
for M in bf16s f32; do
  run "eager" X=1 -- --math $M
  run "graph default" X=1 -- --math $M --graph
  run "graph PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- --math $M --graph
  run "graph QUEUES=8" DEBUG_HIP_FORCE_GRAPH_QUEUES=8 -- --math $M --graph
  run "graph QUEUES=8 PACKET_CAPTURE=0" DEBUG_HIP_FORCE_GRAPH_QUEUES=8 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- --math $M --graph
  run "graph QUEUES=2" DEBUG_HIP_FORCE_GRAPH_QUEUES=2 -- --math $M --graph
  run "graph BATCH=1" DEBUG_HIP_GRAPH_BATCH_SIZE=1 -- --math $M --graph
  run "graph BATCH=4096" DEBUG_HIP_GRAPH_BATCH_SIZE=4096 -- --math $M --graph
done
