#!/bin/bash
# VGPRs / AGPRs (one unified file on gfx950: their SUM sets the occupancy) / scratch / waves per SIMD of every kernel in
# libmslice (compile only; no GPU needed)
mkdir -p /tmp/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize "$@" --cuda-device-only -c -o /tmp/exp/dev.o "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"/pyslice_amd/csrc/mslice.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ *\[-Rpass.*//' | paste - - - - - - \
 | sed -e 's/Function Name: //' -e 's/_ZN3msl[0-9]*//' | awk '{printf "%-58s sgpr %-4s vgpr %-4s agpr %-3s scratch %-4s waves/SIMD %s\n", $1, $3, $5, $7, $10, $13}'
