"""Refresh profiles/pmc_traffic_current.json -- the HBM traffic figure bench.py puts into `roofline.traffic` -- from a counter
collection of tools/collect_kernel_profile.sh, and tie it to the kernel sources it was measured on:

    python tools/update_traffic.py <tag> <profiles/summary-name.json>

copies gpurun_out/<tag>/summary.json to the given tracked name and records the dominant slice-loop kernel, its traffic per
launch and sha256 over the sources the slice-loop passes are compiled from (LOOP_SOURCES: the pass kernels, the register FFTs, their
launchers; not the potential / TACAW / reduction kernels or the host code, which cannot change a pass's traffic).  bench.py recomputes
the hash at run time and reports traffic = null when those sources have changed since the counters were taken (a stale constant
must not ride along with a new kernel).
"""
import hashlib
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LOOP_SOURCES = ("fft_pow2.h", "fft_regs.h", "kernel_util.h", "rowt_pass.h", "slice_pass.hip")


def csrc_sha256():
    h = hashlib.sha256()
    d = os.path.join(REPO, "pyslice_amd", "csrc")
    for f in LOOP_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def main():
    tag, dest = sys.argv[1], sys.argv[2]
    src = os.path.join(REPO, "gpurun_out", tag, "summary.json")
    s = json.load(open(src))
    shutil.copy(src, os.path.join(REPO, dest))
    cfg = s["bench_line_of_the_traced_run"]["config"]
    loop = {k: v for k, v in s["kernels"].items() if k.startswith(("rowT", "row_pass", "col_pass", "line_fft")) and "hbm_bytes_per_launch" in v}
    name = max(loop, key=lambda k: loop[k]["total_ms"])
    out = {"grid": cfg["grid"], "probes": cfg["probes"], "frame_batch": cfg.get("frame_batch", 1), "passes_per_slice": 1,
           "kernel": name, "hbm_bytes_per_launch": loop[name]["hbm_bytes_per_launch"], "kernel_trace_avg_us": loop[name]["avg_us"],
           "source": dest, "csrc_sha256": csrc_sha256(),
           "note": "2 x FETCH_SIZE + WRITE_SIZE of separate --pmc passes (MI355X_MICROARCH.md, HBM section); valid only for these sources"}
    json.dump(out, open(os.path.join(REPO, "profiles", "pmc_traffic_current.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
