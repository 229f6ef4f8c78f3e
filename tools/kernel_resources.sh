#!/bin/bash
# VGPRs / scratch bytes / code size of every kernel in libmslice (compile only; no GPU needed)
mkdir -p /tmp/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize "$@" --cuda-device-only -c -o /tmp/exp/dev.o "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"/pyslice_amd/csrc/mslice.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize" | sed -e 's/.*remark: *//' -e 's/ *\[-Rpass.*//' | paste - - - | sed -e 's/Function Name: //' -e 's/_ZN3msl[0-9]*//' | awk '{printf "%-60s %s %s  scratch %s\n", $1, $2, $3, $6}'
