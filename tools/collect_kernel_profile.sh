#!/bin/bash
# rocprofv3 evidence for one bench.py configuration (run through gpurun):
#   tools/collect_kernel_profile.sh <tag> <bench.py args...>
# five separate runs of the same command, as the guide prescribes (counters never together with tracing):
#   kernel-trace statistics, FETCH_SIZE, WRITE_SIZE, and two passes of SQ counters.
# Everything lands in gpurun_out/<tag>/ ; tools/summarize_kernel_profile.py copies the judged summary into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "bench.py $*" > $O/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py "$@" --no-cpu-baseline --no-tacaw > $O/stats.txt 2>&1 || { tail -5 $O/stats.txt; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py "$@" --no-cpu-baseline --no-tacaw > $O/fetch.txt 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py "$@" --no-cpu-baseline --no-tacaw > $O/write.txt 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/sq1 -- python3 $R/bench.py "$@" --no-cpu-baseline --no-tacaw > $O/sq1.txt 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/bench.py "$@" --no-cpu-baseline --no-tacaw > $O/sq2.txt 2>&1 || exit 1
cd $R && python3 tools/summarize_kernel_profile.py $tag
