#!/bin/bash
# tools/asm_order.sh <file.hip> <kernel-name-substring>: run-length summary of the MFMA / LDS / LDS-DMA /
# wait / barrier order hipcc emitted for one kernel (tuning aid).
set -e
src=$1; pat=$2
d=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -save-temps=obj -c "$src" -o $d/o.o 2>/dev/null
s=$(ls $d/*amdgcn*.s)
awk "/^[A-Za-z_0-9]*$pat[A-Za-z_0-9]*:/,/s_endpgm/" $s > $d/k.s
grep -E "v_mfma|global_load_lds|s_barrier|ds_read_b128|ds_write|s_waitcnt|global_load_dword|global_store|s_cbranch|^\.LBB" $d/k.s | awk '{print $1, ($1=="s_waitcnt"? $2 $3 : "")}' | awk '{k=$0; if (k!=prev){ if(prev!="") printf "%s x%d\n", prev, n; n=0; prev=k}; n++} END{printf "%s x%d\n", prev, n}'
rm -rf $d
