#!/bin/bash
# MFMA counters of the structure-factor kernels (one --pmc pass; the program directly behind `--`)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3_sfprof
rocprofv3 -L 2>/dev/null | grep -i "mfma" | head -20 > $R/gpurun_out/r3_sfprof/counters.txt
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3_sfprof/p1 -- python3 $R/bench.py --grid 512 --slices 100 --probes 1 --steps 32 --warmup 32 --no-cpu-baseline --no-tacaw > $R/gpurun_out/r3_sfprof/p1.txt 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/r3_sfprof/p1/**/*counter_collection.csv',recursive=True)
d=collections.defaultdict(lambda: collections.defaultdict(list))
if not f:
    raise SystemExit('no counter_collection.csv: the rocprofv3 pass failed, see p1.txt')
for r in csv.DictReader(open(f[0])):
    d[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    if 'structure' in k or 'rowT2' in k:
        print(k, {c: sorted(x)[len(x)//2] for c,x in v.items()})
PY
