"""Copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/summarize_profiles.py r01_onepass
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else "current"
shutil.copy(newest("gpurun_out/prof_stats/runc/*_kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats.csv")


def med(pattern, counter):
    rows = list(csv.DictReader(open(newest(pattern))))
    d = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in d.items()}, {k: len(v) for k, v in d.items()}


f, nf = med("gpurun_out/pmc_fetch/runc/*counter_collection.csv", "FETCH_SIZE")
w, nw = med("gpurun_out/pmc_write/runc/*counter_collection.csv", "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
       "units": "KiB per launch (median over launches); FETCH_SIZE is doubled for hbm_bytes (gfx950 tallies 128-byte requests at "
                "64 B; calibrated on tools/membench copies, profiles/r01_fourstep_pmc_traffic.json)",
       "kernels": {}}
for k in sorted(f, key=lambda k: -f[k] * nf[k])[:6]:
    out["kernels"][k] = {"launches": nf[k], "FETCH_SIZE_raw_KiB": f[k], "WRITE_SIZE_KiB": w.get(k),
                         "hbm_bytes_per_launch": (2 * f[k] + w.get(k, 0.0)) * 1024}
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
dom = max((k for k in out["kernels"] if "rowT_pass" in k or "row_pass" in k or "col_pass" in k),
          key=lambda k: out["kernels"][k]["launches"])
cur = {"grid": 1024, "probes": 64, "passes_per_slice": 1 if "rowT" in dom else 2, "kernel": dom,
       "hbm_bytes_per_launch": out["kernels"][dom]["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc_traffic.json"}
json.dump(cur, open("profiles/pmc_traffic_current.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
print(cur)
