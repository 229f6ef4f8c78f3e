#!/usr/bin/env python
"""Benchmark of the multislice hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one MD frame through the hot path on one GPU: projected Kirkland potential ->
(P probes x nz slices) fused FFT / transmission / Fresnel slice loop -> exit-wave FFT epilogue,
all inputs already resident in HBM except the frame's atom positions (a few MB).  Workload at
N=1 is BASELINE.json configs[2]: 64-probe STEM grid, 1024^2 grid, 200 slices (the configuration
the metric is quoted on).  With N>1 every rank runs its own K frames (frame sharding, no
data-path collective): weak scaling, value = all ranks' slice-steps / max-over-ranks time.

Prints ONE JSON line on rank 0 with the throughput plus
  roofline     -- dominant slice-loop kernel: algorithmic bytes per launch / mean launch duration
                  (HIP events on the library's stream, taken inside the timed region)
  cpu_baseline -- the NumPy oracle (port of the reference's NumPy path) timed on this host on a
                  bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--slices", type=int, default=200)
    ap.add_argument("--probes", type=int, default=64)
    ap.add_argument("--aperture", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-slices", type=int, default=100,
                    help="slices of the bounded CPU sample (100 of 200 at 1024^2: about 12 s of single-thread work)")
    ap.add_argument("--tacaw", action="store_true", help="also time the TACAW time->frequency FFT over the K frames")
    ap.add_argument("--no-launch-timing", action="store_true",
                    help="run the library as production does (no per-launch HIP events, frames queue asynchronously); "
                         "the roofline block is then null")
    return ap.parse_args()


def cpu_baseline(grid, nz_full, probes, aperture, nz_sample):
    """Oracle (NumPy complex128, single thread) on a bounded sample: same grid, same atom density,
    `nz_sample` slices, 1 probe; extrapolated exactly linearly to `probes` probes sharing the potential."""
    from oracle import multislice_oracle as orc
    from pyslice_amd.synthetic import synthetic_trajectory
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        ctx = None
    tr = synthetic_trajectory(grid, nz_sample, 1, seed=0)
    xs, ys, zs, lx, ly, lz = orc.grid_from_box(tr.box_matrix)
    t0 = time.perf_counter()
    V = orc.potential(xs, ys, zs, tr.positions[0], tr.atom_types)
    t_pot = time.perf_counter() - t0
    pr = orc.batched_probes(orc.probe_array(xs, ys, aperture, 100e3), xs, ys, [(lx / 2, ly / 2)])
    t0 = time.perf_counter()
    ex = orc.propagate(pr, V, xs, ys, zs, 100e3)
    orc.diffraction(ex)
    t_prop = time.perf_counter() - t0
    if ctx is not None:
        ctx.restore_original_limits() if hasattr(ctx, "restore_original_limits") else None
    value = probes * nz_sample / (t_pot + probes * t_prop)
    return {"value": round(value, 3), "unit": "slice-steps/s", "cores": 1, "kind": "port",
            "sample": f"NumPy c128 oracle, {grid}^2 grid, {nz_sample} of {nz_full} slices at full atom density "
                      f"({tr.n_atoms} atoms), 1 frame, 1 probe measured (potential {t_pot:.2f}s + slice loop "
                      f"{t_prop:.2f}s) and extrapolated linearly to {probes} probes sharing the potential",
            "host_cpus": os.cpu_count()}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    backend = os.environ.get("MSL_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path on fewer GPUs than ranks
    n_dev = max(1, torch.cuda.device_count())
    if backend == "gloo":
        local_rank = local_rank % n_dev
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if world != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from pyslice_amd import _native
    from pyslice_amd.multislice import interaction_sigma, wavelength
    from pyslice_amd.potentials import gridFromTrajectory, loadKirkland, slice_edges
    from pyslice_amd.synthetic import stem_probe_grid, synthetic_trajectory

    n, nz, P = a.grid, a.slices, a.probes
    n_frames = a.steps + a.warmup
    tr = synthetic_trajectory(n, nz, n_frames, seed=100 * rank)        # every rank has its own frames
    xs, ys, zs, lx, ly, lz = gridFromTrajectory(tr)
    assert (len(xs), len(ys), len(zs)) == (n, n, nz), (len(xs), len(ys), len(zs))
    side = int(round(P ** 0.5))
    pp = stem_probe_grid(side) if side * side == P else np.random.default_rng(0).random((P, 2)) * [lx, ly]
    eng = _native.Engine(n, n, nz, xs[1] - xs[0], ys[1] - ys[0], zs[1] - zs[0] if nz > 1 else 0.5, wavelength(100e3),
                         interaction_sigma(100e3), n_probes=P, n_frames=n_frames, device=local_rank,
                         launch_timing=not a.no_launch_timing)
    eng.set_kirkland(loadKirkland())
    eng.set_slices(*slice_edges(zs))
    eng.set_probes(a.aperture, pp)
    Z = np.asarray(tr.atom_types, dtype=np.int32)

    def step(i):
        eng.build_potential(tr.positions[i], Z, 2)
        eng.propagate_frame(i)

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    eng.reset_counters()
    t0 = time.perf_counter()
    for i in range(a.warmup, n_frames):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ctr = eng.counters()

    tacaw_ms = None
    if a.tacaw and n_frames >= 2:
        eng.tacaw()
        tacaw_ms = eng.counters()["ms_tacaw"]

    if rank == 0:
        npix = n * n
        steps_total = world * a.steps * P * nz
        value = steps_total / dt
        # Slice-loop kernels.  Every launch streams the P wave functions of the frame once: 8 B read + 8 B write per
        # pixel = 16 B x nx x ny x P algorithmic bytes per launch.  The default one-pass loop needs ONE
        # such launch per slice (16 B/pixel/slice-step: the separable Fresnel propagator lets a pass finish one
        # propagation and start the next); the two-pass loop (MSL_SLICE_PATH=2) needs two (SURVEY 8d's 32 B).
        rows, cols = (ctr["row_launches"], ctr["ms_row"]), (ctr["col_launches"], ctr["ms_col"])
        name, (cnt, ms) = max((("pass_along_y", rows), ("pass_along_x", cols)), key=lambda kv: kv[1][1])
        bytes_per_launch = 16.0 * npix * P
        passes_per_slice = (rows[0] + cols[0]) / float(a.steps * nz) if nz else 0.0
        roof = None
        if cnt:
            avg_s = ms * 1e-3 / cnt
            ach = bytes_per_launch / avg_s / 1e9
            traffic = None
            tpath = os.path.join(REPO, "profiles", "pmc_traffic_current.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("grid") == n and tj.get("probes") == P and round(passes_per_slice) == tj.get("passes_per_slice"):
                    traffic = tj.get("hbm_bytes_per_launch")
            roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_us": round(avg_s * 1e6, 2), "launches": int(cnt),
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "passes_per_slice": round(passes_per_slice, 3),
                    "pass_along_y_GBps": round(bytes_per_launch * rows[0] / (rows[1] * 1e-3) / 1e9, 1) if rows[1] else None,
                    "pass_along_x_GBps": round(bytes_per_launch * cols[0] / (cols[1] * 1e-3) / 1e9, 1) if cols[1] else None,
                    "slice_loop_GBps_on_32B_per_slice_step_basis":
                        round(32.0 * npix * P * nz * a.steps / (ctr["ms_slice_kernels"] * 1e-3) / 1e9, 1)
                        if ctr["ms_slice_kernels"] else None}
        out = {
            "metric": "slice-steps/sec (probes x frames x slices / s), potential + slice loop + exit FFT",
            "value": round(value, 1), "unit": "slice-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "complex64 (f32)", "data": "synthetic",
            "config": {"workload": f"{P}-probe STEM grid, {n}x{n} grid, {nz} slices, {a.steps} MD frames per GPU "
                                   f"(BASELINE configs[2] per-frame work), {tr.n_atoms} atoms, 30 mrad, 100 keV",
                       "grid": n, "slices": nz, "probes": P, "frames_per_gpu": a.steps, "parallelism": f"frames x{world}"},
            "breakdown_ms_per_step": {"potential": round(ctr["ms_potential"] / a.steps, 3),
                                      "slice_loop_and_epilogue": round(ctr["ms_propagate"] / a.steps, 3)},
            "roofline": roof,
        }
        if tacaw_ms is not None:
            out["tacaw"] = {"ms": round(tacaw_ms, 3), "GBps": round(12.0 * P * n_frames * npix / (tacaw_ms * 1e-3) / 1e9, 1),
                            "frames": n_frames}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, nz, P, a.aperture, a.cpu_slices)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
