#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --grid 512 --slices 100 --probes 1 --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_c2.txt 2>&1
