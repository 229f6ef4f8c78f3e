"""Summarise gpurun_out/<tag>/ (made by tools/collect_kernel_profile.sh) into gpurun_out/<tag>/summary.json: for the slice-loop
kernels of the run, calls and average duration from the kernel trace, HBM traffic per launch from the FETCH_SIZE /
WRITE_SIZE passes (FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md section HBM) and the SQ
counters per launch (medians).  Copy the file into profiles/ under a per-round name to have it judged.

    python tools/summarize_kernel_profile.py <tag>
"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
O = os.path.join("gpurun_out", tag)


def one(pattern):
    g = glob.glob(os.path.join(O, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None


def short(name):
    return name.split("(")[0].replace("void ", "").replace("msl::", "")


stats = {}
f = one("stats/**/*kernel_stats.csv")
for r in csv.DictReader(open(f)):
    stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                               "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "share_pct": round(float(r["Percentage"]), 2)}


def medians(sub):
    f = one(sub + "/**/*counter_collection.csv")
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in d.items()}


fetch, write, sq1, sq2 = medians("fetch"), medians("write"), medians("sq1"), medians("sq2")
bench = None
for line in open(os.path.join(O, "stats.txt")):
    if line.startswith("{"):
        bench = json.loads(line)
out = {"command": open(os.path.join(O, "command.txt")).read().strip() + " --no-cpu-baseline --no-tacaw (five rocprofv3 runs: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; two --pmc SQ passes)",
       "bench_line_of_the_traced_run": {k: bench[k] for k in ("value", "ms_per_step", "config", "roofline")} if bench else None,
       "units": "durations from the kernel trace; traffic in bytes per launch = 2 x FETCH_SIZE (KiB) + WRITE_SIZE (KiB), medians over launches; "
                "SQ_* cycle counters are quad-cycles summed over the waves of a launch",
       "kernels": {}}
top = sorted(stats, key=lambda k: -stats[k]["total_ms"])[:8]
for k in top:
    e = dict(stats[k])
    if k in fetch and k in write:
        fs, ws = fetch[k].get("FETCH_SIZE", 0.0), write[k].get("WRITE_SIZE", 0.0)
        e["FETCH_SIZE_raw_KiB"], e["WRITE_SIZE_KiB"] = fs, ws
        e["hbm_bytes_per_launch"] = (2 * fs + ws) * 1024
    if k in sq1 or k in sq2:
        e["sq"] = {**sq1.get(k, {}), **sq2.get(k, {})}
        s = e["sq"]
        if s.get("SQ_WAVE_CYCLES"):
            e["per_wave_fraction_issuing"] = round(s.get("SQ_ACTIVE_INST_ANY", 0) / s["SQ_WAVE_CYCLES"], 3)
            e["per_wave_fraction_waitcnt"] = round(s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"], 3)
            e["per_wave_fraction_issue_stall"] = round(s.get("SQ_WAIT_INST_ANY", 0) / s["SQ_WAVE_CYCLES"], 3)
    out["kernels"][k] = e
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
for k in top[:4]:
    e = out["kernels"][k]
    print(k, e["calls"], "calls", e["avg_us"], "us", "traffic/launch", e.get("hbm_bytes_per_launch"))
