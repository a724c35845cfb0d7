#!/bin/bash
# Same-box A/B of two builds of libgrl_hip.so: tools/ab_bench.sh <libA> <libB> [bench.py args...]
# Alternates the two libraries (A B A B A B) and prints ms_per_step of every run: boxes differ by 2-5 %, so only numbers
# from one gpurun call are comparable.
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in "$A" "$B"; do
    ms=$(GRL_HIP_LIB=$L python bench.py --no-cpu-baseline --no-alt --no-train-block "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$(basename $L) $* : $ms"
  done
done
