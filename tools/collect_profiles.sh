#!/bin/bash
# Collect the rocprofv3 evidence for the current build on the GPU box (run through gpurun):
#   kernel-trace stats of bench.py, and FETCH_SIZE / WRITE_SIZE in separate --pmc passes (the guide's recipe).
# Outputs land in gpurun_out/; tools/summarize_profiles.py copies the summaries into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.txt 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.txt 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_write.txt 2>&1 || exit 1
echo collected
