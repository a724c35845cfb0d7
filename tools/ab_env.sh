#!/bin/bash
# Same-box A/B of an environment switch: tools/ab_env.sh VAR=a VAR=b [bench.py args...]   (alternates a b a b a b)
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for E in "$A" "$B"; do
    ms=$(env $E python bench.py --no-cpu-baseline --no-alt --no-train-block "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$E $* : $ms"
  done
done
