#!/bin/bash
# Every fusion / stream switch of the eval and train pipelines off, ONE at a time, against the default -- all in one gpurun
# call (boxes differ by 2-5 %): the check that each of them still pays in the current pipeline.  ms per step.
#   bash tools/switch_ab.sh > gpurun_out/switch_ab.txt
run() { env $1 python bench.py --no-cpu-baseline --no-alt --no-train-block ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run_train() { env $1 python bench.py --no-cpu-baseline --mode train ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
C2="--math bf16s --clips 64 --seq-len 8"
for rep in 1 2; do
  echo "configs[2] default $(run X=1 $C2)"
  for f in GRL_FUSE_BNECK=0 GRL_FUSE_DOWN=0 GRL_CONV3X3_C64=0 GRL_FUSE_STEM_POOL=0 GRL_STEM_POOL2=0 GRL_FUSE_TAIL_L23=1 GRL_TRL_STREAMS=0 GRL_TRL_ATT_STREAMS=0 GRL_GEMM_PANEL=0; do
    echo "configs[2] $f $(run $f $C2)"; done
  echo "fp32 headline default $(run X=1)"
  for f in GRL_FUSE_BNECK=0 GRL_FUSE_STEM_POOL_F32=0 GRL_TRL_STREAMS=0 GRL_GEMM_PANEL=0 GRL_GEMM_WIDE=0; do echo "fp32 headline $f $(run $f)"; done
done
for m in f32 bf16s; do
  for rep in 1 2; do
    echo "train $m default $(run_train X=1 --math $m)"
    for f in GRL_BN_REDUCE_FUSED=0 GRL_BN_REDUCE_FUSED_BF16=0 GRL_STEM_WGRAD_FUSED=0 GRL_PREP_ASYNC=0 GRL_RELU_BITS=0 GRL_STEM_TAIL_FUSED=0 GRL_WGRAD_STREAM=0 GRL_TRL_STREAMS=0 GRL_WGRAD_XCD=0 GRL_WGRAD_STACK=0; do
      echo "train $m $f $(run_train $f --math $m)"; done
  done
done
