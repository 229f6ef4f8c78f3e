#!/bin/bash
# rocprofv3 kernel-trace statistics of a bench.py run (through gpurun):  tools/prof_stats.sh <name> <bench.py args...>
# writes gpurun_out/<name>/ and prints the top kernels (calls, total ms, average us, share)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
name=$1; shift
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name/trace -- python3 $R/bench.py "$@" > $R/gpurun_out/$name/bench.txt 2>&1 || { tail -5 $R/gpurun_out/$name/bench.txt; exit 1; }
f=$(find $R/gpurun_out/$name/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/$name/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print(r["Name"][:78].ljust(78), r["Calls"].rjust(7), ("%.2f" % (float(r["TotalDurationNs"]) / 1e6)).rjust(9), "ms",
          ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", r["Percentage"].rjust(6))
PY
grep -h '^{' $R/gpurun_out/$name/bench.txt | tail -1 | cut -c1-600
