#!/bin/bash
# builds tools/hw_probe/ring_probe (and ring_probe_ko<N> for every N given) -- exits non-zero on any compile error
set -e
cd "$(dirname "$0")/../.."
B="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-value tools/hw_probe/ring_probe.hip"
rm -f tools/hw_probe/ring_probe tools/hw_probe/ring_probe_ko*
$B $EXTRA -o tools/hw_probe/ring_probe -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | grep -E "error|ring.hip" | sort | uniq -c
test -x tools/hw_probe/ring_probe
for ko in "$@"; do $B $EXTRA -DGRL_RING_KO=$ko -o tools/hw_probe/ring_probe_ko$ko 2>&1 | grep error || true; test -x tools/hw_probe/ring_probe_ko$ko; done
echo built
