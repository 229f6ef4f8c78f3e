#!/bin/bash
# rocprofv3 evidence for the TACAW time transform alone (through gpurun):  tools/collect_tacaw_profile.sh <tag> <tacaw_bench.py args>
# kernel-trace statistics and two --pmc passes of SQ counters, each its own run (counters never together with tracing)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "tools/tacaw_bench.py $*" > $O/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/tacaw_bench.py "$@" > $O/stats.txt 2>&1 || { tail -5 $O/stats.txt; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/sq1 -- python3 $R/tools/tacaw_bench.py "$@" > $O/sq1.txt 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/tools/tacaw_bench.py "$@" > $O/sq2.txt 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/tacaw_bench.py "$@" > $O/fetch.txt 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/tacaw_bench.py "$@" > $O/write.txt 2>&1 || exit 1
grep "^T=" $O/stats.txt
cd $R && python3 - $tag <<'PY'
import collections, csv, glob, json, os, sys
O = os.path.join("gpurun_out", sys.argv[1])
def one(p):
    g = glob.glob(os.path.join(O, p), recursive=True); return max(g, key=os.path.getmtime) if g else None
def med(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    f = one(sub + "/**/*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msl::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in d.items()}
stats = {}
for r in csv.DictReader(open(one("stats/**/*kernel_stats.csv"))):
    stats[r["Name"].split("(")[0].replace("void ", "").replace("msl::", "")] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
sq = {}
for part in (med("sq1"), med("sq2"), med("fetch"), med("write")):
    for k, v in part.items(): sq.setdefault(k, {}).update(v)
out = {"command": open(os.path.join(O, "command.txt")).read().strip(), "line": [l.strip() for l in open(os.path.join(O, "stats.txt")) if l.startswith("T=")], "kernels": {}}
for k, s in sq.items():
    if k in stats and ("time_" in k or "col_pass" in k or "line_fft" in k):
        e = dict(stats[k]); e["counters"] = s
        w = s.get("SQ_WAVE_CYCLES", 0)
        if w:
            e["per_wave_fraction_issuing"] = round(s.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)
            e["per_wave_fraction_waitcnt"] = round(s.get("SQ_WAIT_ANY", 0) / w, 3)
            e["per_wave_fraction_issue_stall"] = round(s.get("SQ_WAIT_INST_ANY", 0) / w, 3)
            e["lds_conflict_fraction"] = round(s.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, s.get("SQ_LDS_IDX_ACTIVE", 1.0)), 3)
        if "FETCH_SIZE" in s and "WRITE_SIZE" in s: e["hbm_bytes_per_launch"] = (2 * s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024
        out["kernels"][k] = e
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
