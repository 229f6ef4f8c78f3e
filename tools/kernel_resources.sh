#!/bin/bash
# VGPRs / AGPRs (one unified file on gfx950: their SUM sets the occupancy) / scratch / waves per SIMD of every kernel in
# libmslice (compile only; no GPU needed)
mkdir -p /tmp/exp
C="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"/pyslice_amd/csrc
for f in "$C"/*.hip; do
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize "$@" --cuda-device-only -c -o /tmp/exp/$(basename $f .hip).o $f -Rpass-analysis=kernel-resource-usage > /tmp/exp/$(basename $f .hip).log 2>&1 &
done
wait
cat /tmp/exp/mslice.log /tmp/exp/tacaw_*.log \
 | grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ *\[-Rpass.*//' | paste - - - - - - \
 | sed -e 's/Function Name: //' -e 's/_ZN3msl[0-9]*//' | awk '{printf "%-58s sgpr %-4s vgpr %-4s agpr %-3s scratch %-4s waves/SIMD %s\n", $1, $3, $5, $7, $10, $13}'
