#!/bin/bash
# kernel resource usage of one .hip unit: tools/kres.sh grl_amd/csrc/fuse_bf16.hip [name filter]
f=$1; pat=${2:-.}
out=$(/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
  $KRES_EXTRA -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | grep -v "not a recognized")
echo "$out" | grep -E "error" -A3 | head -12
echo "$out" | awk '/Function Name/{name=$5} / VGPRs:/{v=$4} /AGPRs:/{a=$4} /VGPRs Spill/{sp=$5} /LDS Size/{print name, "vgpr", v, "agpr", a, "spill", sp}' | while read n rest; do echo "$(echo $n | c++filt | cut -c1-100) $rest"; done | grep -E "$pat"
