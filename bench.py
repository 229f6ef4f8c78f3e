#!/usr/bin/env python
"""Benchmark of the multislice hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one MD frame through the hot path on one GPU: projected Kirkland potential ->
(P probes x nz slices) fused FFT / transmission / Fresnel slice loop -> exit-wave FFT epilogue,
all inputs already resident in HBM except the frame's atom positions (a few MB).  Workload at
N=1 is BASELINE.json configs[2]: 64-probe STEM grid, 1024^2 grid, 200 slices (the configuration
the metric is quoted on).

N > 1: one process per GPU over RCCL.  Launched under torch.distributed.run the ranks come from
the environment; launched plainly (`python bench.py --gpus N`) the script starts the N rank
processes itself, before anything touches a GPU.  Default "weak" scaling: every rank runs its own
K frames (frame sharding, no data-path collective), value = all ranks' slice-steps / max-over-ranks
time.  `--scaling strong`: every step is one round of --frames-per-step frames sharded over the
ranks, so the total work is fixed as N grows.  After the timed region the end-of-run exchanges
of the sharded path are timed on the frames just computed and reported as `exchange_ms`:
gather of the frame shards on rank 0 (WFData), and all-to-all frames->probes + device time FFT +
gather of the intensities (TACAW).

Prints ONE JSON line on rank 0 with the throughput plus
  roofline     -- dominant slice-loop kernel: algorithmic bytes per launch / mean launch duration
                  (HIP events on the library's stream, taken inside the timed region)
  tacaw        -- (N=1) the time->frequency FFT over T=256 resident frame slots (BASELINE C3's T)
  cpu_baseline -- the NumPy oracle (port of the reference's NumPy path) timed on this host on a
                  bounded sample of the same workload (rank 0, N=1 only): single thread, and
                  `cpu_baseline_allcores` with scipy.fft workers + threaded BLAS.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
EXIT_EXCHANGE_STALLED = 3 # exit code when the end-of-run exchanges (or the final barrier) of an N > 1 run hung: line printed, run failed
TACAW_T = 256             # BASELINE C3's frame count; the four-step time-FFT kernel serves T = 256 and 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--slices", type=int, default=200)
    ap.add_argument("--probes", type=int, default=64)
    ap.add_argument("--aperture", type=float, default=30.0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--frames-per-step", type=int, default=8,
                    help="--scaling strong: MD frames per step, sharded over the ranks (fixed total work)")
    ap.add_argument("--frame-batch", type=int, default=0,
                    help="MD frames sharing every slice-loop launch (0 = the calculator's rule, calculators.default_frame_batch: about 256 images per launch)")
    ap.add_argument("--exchange-timeout", type=float, default=240.0,
                    help="N>1: seconds the end-of-run exchanges may take before the line is printed without them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-slices", type=int, default=100,
                    help="slices of the bounded CPU sample (100 of 200 at 1024^2: about 12 s of single-thread work)")
    ap.add_argument("--no-tacaw", action="store_true", help="skip the TACAW time->frequency FFT leg (N=1)")
    ap.add_argument("--tacaw-frames", type=int, default=TACAW_T,
                    help="frame slots of the TACAW leg (default 256 = BASELINE C3; 100 = the reference notebook's run)")
    ap.add_argument("--no-exchange", action="store_true", help="skip the end-of-run exchange timing (N>1)")
    ap.add_argument("--no-c3-full", action="store_true",
                    help="N=1: do not fill the remaining frame slots with REAL frames after the timed region (c3_full block)")
    ap.add_argument("--stream", action="store_true",
                    help="streaming TACAW (the only representable form of BASELINE C5): every frame goes through a ring of --stream-tile "
                         "frame slots (k-window --k-window, detector bin --k-bin) and is folded into the time->frequency transform "
                         "inside the timed region; N>1: each rank folds its own frames with global time indices, and after the timed "
                         "region the partial sums are reduce-scattered over probes, finished and gathered (stream_ms)")
    ap.add_argument("--stream-tile", type=int, default=8)
    ap.add_argument("--k-window", type=int, default=512)
    ap.add_argument("--k-bin", type=int, default=4)
    ap.add_argument("--no-launch-timing", action="store_true",
                    help="run the library as production does (no per-launch HIP events, frames queue asynchronously); "
                         "the roofline block is then null")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters, nothing in this
    process has touched a GPU or imported torch), wait for them, relay the first failure."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:           # a rank died: the others would wait in the rendezvous for ever
                    q.terminate()
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------------------------------------
def _sample(grid, nz_sample, seed=0):
    from oracle import multislice_oracle as orc
    from pyslice_amd.synthetic import synthetic_trajectory
    tr = synthetic_trajectory(grid, nz_sample, 1, seed=seed)
    return orc, tr, orc.grid_from_box(tr.box_matrix)


def cpu_baseline(grid, nz_full, probes, aperture, nz_sample):
    """Oracle (NumPy complex128, single thread) on a bounded sample: same grid, same atom density,
    `nz_sample` slices, 1 probe; extrapolated exactly linearly to `probes` probes sharing the potential."""
    from threadpoolctl import threadpool_limits
    orc, tr, (xs, ys, zs, lx, ly, lz) = _sample(grid, nz_sample)
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        V = orc.potential(xs, ys, zs, tr.positions[0], tr.atom_types)
        t_pot = time.perf_counter() - t0
        pr = orc.batched_probes(orc.probe_array(xs, ys, aperture, 100e3), xs, ys, [(lx / 2, ly / 2)])
        t0 = time.perf_counter()
        ex = orc.propagate(pr, V, xs, ys, zs, 100e3)
        orc.diffraction(ex)
        t_prop = time.perf_counter() - t0
    value = probes * nz_sample / (t_pot + probes * t_prop)
    return {"value": round(value, 3), "unit": "slice-steps/s", "cores": 1, "kind": "port",
            "sample": f"NumPy c128 oracle, {grid}^2 grid, {nz_sample} of {nz_full} slices at full atom density "
                      f"({tr.n_atoms} atoms), 1 frame, 1 probe measured (potential {t_pot:.2f}s + slice loop "
                      f"{t_prop:.2f}s) and extrapolated linearly to {probes} probes sharing the potential",
            "host_cpus": os.cpu_count()}, V


def usable_cores():
    """Cores this process may really use: the cgroup CPU quota when there is one (the GPU box gives one GPU's share of
    the host, 16 of 256 -- running 256 threads on it is slower than one), else the affinity mask; MSL_BENCH_CPU_WORKERS
    overrides."""
    if os.environ.get("MSL_BENCH_CPU_WORKERS"):
        return max(1, int(os.environ["MSL_BENCH_CPU_WORKERS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16) if n > 64 else n          # no quota visible on a 256-thread host: the documented share of one GPU


def cpu_baseline_allcores(grid, nz_full, probes, aperture, nz_sample, V):
    """The same oracle with every core this process may use: scipy.fft (pocketfft, `workers` threads, the measured probes
    batched in one call) for the slice loop, threaded BLAS for the structure-factor products of the potential
    (SURVEY 8d's second CPU leg).  Bounded sample: `nz_sample` slices, 4 probes measured, extrapolated to `probes`."""
    from threadpoolctl import threadpool_limits
    cores = usable_cores()
    orc, tr, (xs, ys, zs, lx, ly, lz) = _sample(grid, nz_sample)
    pm = min(4, probes)
    with threadpool_limits(limits=cores):
        t0 = time.perf_counter()
        V2 = orc.potential(xs, ys, zs, tr.positions[0], tr.atom_types)        # structure-factor products on `cores` BLAS threads
        t_pot = time.perf_counter() - t0
    pos = [(lx * (i + 1) / (pm + 1), ly / 2) for i in range(pm)]
    pr = orc.batched_probes(orc.probe_array(xs, ys, aperture, 100e3), xs, ys, pos)
    t0 = time.perf_counter()
    ex = orc.propagate(pr, V2 if V is None else V, xs, ys, zs, 100e3, workers=cores)
    orc.diffraction(ex, workers=cores)
    t_prop = (time.perf_counter() - t0) / pm
    value = probes * nz_sample / (t_pot + probes * t_prop)
    return {"value": round(value, 3), "unit": "slice-steps/s", "cores": cores, "kind": "port",
            "sample": f"NumPy c128 oracle with scipy.fft workers={cores} and threaded BLAS, {grid}^2 grid, {nz_sample} of "
                      f"{nz_full} slices, 1 frame, {pm} probes measured in one batch (potential {t_pot:.2f}s + slice loop "
                      f"{t_prop:.2f}s per probe) and extrapolated linearly to {probes} probes sharing the potential",
            "host_cpus": os.cpu_count()}


# ---------------------------------------------------------------------------------------------------------
def run(a):
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL across processes needs it)
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
        assert dist.get_world_size() == world
    if world != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from pyslice_amd import _native
    from pyslice_amd import distributed as D
    from pyslice_amd.multislice import interaction_sigma, wavelength
    from pyslice_amd.potentials import gridFromTrajectory, loadKirkland, slice_edges
    from pyslice_amd.synthetic import stem_probe_grid, synthetic_trajectory

    n, nz, P = a.grid, a.slices, a.probes
    npix = n * n
    strong = a.scaling == "strong"
    if strong:
        # one trajectory for the whole job; step s = frames [s*F, (s+1)*F), each rank takes its contiguous part
        F = a.frames_per_step
        tr = synthetic_trajectory(n, nz, (a.steps + a.warmup) * F, seed=0)
        lo, hi = D.shard_bounds(F, world, rank)
        my_frames = [[s * F + f for f in range(lo, hi)] for s in range(a.steps + a.warmup)]
    else:
        tr = synthetic_trajectory(n, nz, a.steps + a.warmup, seed=100 * rank)        # every rank has its own frames
        my_frames = [[s] for s in range(a.steps + a.warmup)]
    n_local = sum(len(f) for f in my_frames)
    xs, ys, zs, lx, ly, lz = gridFromTrajectory(tr)
    assert (len(xs), len(ys), len(zs)) == (n, n, nz), (len(xs), len(ys), len(zs))
    side = int(round(P ** 0.5))
    pp = stem_probe_grid(side) if side * side == P else np.random.default_rng(0).random((P, 2)) * [lx, ly]

    # N=1: hold T = 256 frame slots like the full C3 run (137 GB of spectra + 69 GB of intensities at 64 probes x 1024^2)
    # when the device has the room, so that the TACAW leg runs the kernel a C3 run uses
    slots = max(1, n_local)
    tacaw_T = None
    if world == 1 and not a.no_tacaw and n_local >= 2:
        free_b, _ = torch.cuda.mem_get_info(local_rank)
        want = max(a.tacaw_frames, n_local)
        if 12.0 * P * want * npix + 24e9 < free_b and n_local <= a.tacaw_frames:
            slots, tacaw_T = want, want
        else:
            tacaw_T = n_local
    # BASELINE C3 as configured: after the timed region the remaining slots are filled with REAL frames of the same trajectory
    # (the synthetic trajectory does not depend on its length: frame f is frame f) and the time FFT runs on them
    c3_full = (world == 1 and not a.no_c3_full and not a.stream and not strong and tacaw_T is not None and tacaw_T == slots and tacaw_T > n_local)
    if c3_full:
        tr = synthetic_trajectory(n, nz, tacaw_T, seed=0)
    from pyslice_amd.calculators import default_frame_batch
    fb = a.frame_batch if a.frame_batch > 0 else default_frame_batch(P, nz, n, n)      # the calculator's own default
    fb = max(1, min(fb, a.steps))
    stream = a.stream
    if stream:
        if strong:
            raise SystemExit("--stream runs with weak scaling (every rank folds its own block of frames)")
        slots = max(1, min(a.stream_tile, n_local))
        fb = max(1, min(fb, slots))
        tacaw_T = None
        kw = min(a.k_window, n)
    eng = _native.Engine(n, n, nz, xs[1] - xs[0], ys[1] - ys[0], zs[1] - zs[0] if nz > 1 else 0.5, wavelength(100e3),
                         interaction_sigma(100e3), n_probes=P, n_frames=slots, device=local_rank,
                         launch_timing=not a.no_launch_timing, frame_batch=fb,
                         window=(kw, kw) if stream else None, k_bin=(a.k_bin, a.k_bin) if stream and a.k_bin > 1 else None)
    eng.set_kirkland(loadKirkland())
    eng.set_slices(*slice_edges(zs))
    eng.set_probes(a.aperture, pp)
    Z = np.asarray(tr.atom_types, dtype=np.int32)
    slot_of = {}
    for fs in my_frames:
        for f in fs:
            slot_of[f] = len(slot_of)

    def step(s):
        for f in my_frames[s]:
            eng.build_potential(tr.positions[f], Z, 2)
            eng.propagate_frame(slot_of[f])

    def steps(s0, s1):
        """steps [s0, s1): frame by frame, or -- frame batching -- in groups of B frames per sequence of launches"""
        if eng.frame_batch == 1:
            for s in range(s0, s1):
                step(s)
            return
        todo = [f for s in range(s0, s1) for f in my_frames[s]]
        for i in range(0, len(todo), eng.frame_batch):
            chunk = todo[i:i + eng.frame_batch]
            consecutive = chunk[-1] - chunk[0] + 1 == len(chunk)
            eng.build_potentials(tr.positions[chunk[0]:chunk[-1] + 1] if consecutive else tr.positions[chunk], Z, 2)
            eng.propagate_frames(slot_of[chunk[0]], len(chunk))

    # streaming TACAW: rank r's frame i carries the global time index r * n_local + i; all T = world * n_local frames (warm-up
    # included: the fold is linear, and a transform needs every frame) go into one stream, all T frequency bins are kept
    stream_state = {"ref": False}

    def stream_steps(s0, s1):
        todo = [f for s in range(s0, s1) for f in my_frames[s]]
        for t0 in range(0, len(todo), slots):
            tile = todo[t0:t0 + slots]
            for i in range(0, len(tile), eng.frame_batch):
                chunk = tile[i:i + eng.frame_batch]
                if eng.frame_batch > 1:
                    eng.build_potentials(tr.positions[chunk[0]:chunk[-1] + 1], Z, 2)
                    eng.propagate_frames(i, len(chunk))
                else:
                    eng.build_potential(tr.positions[chunk[0]], Z, 2)
                    eng.propagate_frame(i)
            if not stream_state["ref"]:
                # the run's first frame is every rank's reference pattern (rank 0's slot 0, broadcast once)
                if world == 1:
                    eng.tacaw_stream_set_reference(slot=0)
                else:
                    K = eng.wx * eng.wy
                    if rank == 0:
                        eng.tacaw_stream_set_reference(slot=0)
                        eng.synchronize()
                        ref = torch.as_tensor(_native.DeviceArray(eng.device_ptr(_native.BUF_STREAM_REF), (P, K), "<c8", owner=eng), device=dev)
                        D.broadcast_from(ref, src=0)
                    else:
                        ref = torch.empty((P, K), dtype=torch.complex64, device=dev)
                        D.broadcast_from(ref, src=0)
                        torch.cuda.synchronize(dev)
                        eng.tacaw_stream_set_reference(ref_ptr=ref.data_ptr())
                        eng.synchronize()
                stream_state["ref"] = True
            eng.tacaw_stream_push(0, len(tile), rank * n_local + slot_of[tile[0]])

    if stream:
        dev = torch.device("cuda", local_rank)
        eng.tacaw_stream_begin(world * n_local, None)
        slot_of = {f: i for i, f in enumerate(f for fs in my_frames for f in fs)}
        steps = stream_steps

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tw0 = time.perf_counter()
    steps(0, a.warmup)
    fence()
    dt_warm = time.perf_counter() - tw0
    eng.reset_counters()
    t0 = time.perf_counter()
    steps(a.warmup, a.warmup + a.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ctr = eng.counters()
    frames_timed_local = sum(len(my_frames[s]) for s in range(a.warmup, a.warmup + a.steps))
    frames_timed = a.steps * a.frames_per_step if strong else world * a.steps

    dev = torch.device("cuda", local_rank)
    stream_ms = None
    stream_done = None
    if stream and world > 1:
        # the end-of-stream collectives below run over a backend this build could not exercise on multi-GPU hardware: if they
        # stall, the measurement (already complete) is printed with the failure in place of their timings and the rank exits 3
        import threading
        stream_done = threading.Event()

        def stream_watchdog():
            if not stream_done.wait(a.exchange_timeout):
                if rank == 0:
                    line = {"metric": "slice-steps/sec (probes x frames x slices / s), potential + slice loop + exit FFT",
                            "value": round(frames_timed * P * nz / dt, 1), "unit": "slice-steps/s", "n_gpus": world, "steps": a.steps,
                            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": a.scaling,
                            "vs_baseline": None, "dtype": "complex64 (f32)", "data": "synthetic",
                            "config": {"workload": f"{P}-probe STEM grid, {n}x{n} grid, {nz} slices, streaming TACAW", "grid": n, "slices": nz,
                                       "probes": P, "frames_timed": frames_timed, "parallelism": f"frames x{world}"},
                            "stream_ms": {"error": f"end-of-stream reduce / gather did not finish within {a.exchange_timeout} s"}}
                    print(json.dumps(line), flush=True)
                else:
                    time.sleep(2.0)
                os._exit(EXIT_EXCHANGE_STALLED)
        threading.Thread(target=stream_watchdog, daemon=True).start()
    if stream:
        # end of the stream (after the timed region): sum of the ranks' partial sums (reduce-scatter over probes), |.|^2 of this
        # rank's probes, gather of the intensities on rank 0
        F, K = world * n_local, eng.wx * eng.wy

        def view(what, shape, typestr):
            return torch.as_tensor(_native.DeviceArray(eng.device_ptr(what), shape, typestr, owner=eng), device=dev)
        fence()
        t1 = time.perf_counter()
        p0, p1 = D.reduce_probes(view(_native.BUF_STREAM_ACC, (P, F, K), "<c8"), P)
        D.reduce_probes(view(_native.BUF_STREAM_S1, (P, K, 2), "<f8"), P)
        D.reduce_probes(view(_native.BUF_STREAM_S2, (P, K), "<f8"), P)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        mine = torch.empty((p1 - p0, F, K), dtype=torch.float32, device=dev)
        eng.tacaw_stream_finish_range(p0, p1 - p0, mine.data_ptr(), True)
        t3 = time.perf_counter()
        full = D.gather_probes(mine, P, dst=0)
        fence()
        t4 = time.perf_counter()
        stream_ms = {"frames": F, "bins": F, "stored_pixels": K, "ring": slots, "reduce_scatter": round((t2 - t1) * 1e3, 3),
                     "finish": round((t3 - t2) * 1e3, 3), "gather_probes": round((t4 - t3) * 1e3, 3),
                     "accumulator_bytes_per_rank": 8.0 * P * F * K,
                     "intensity_sum": float(full.double().sum()) if full is not None else None}
        del mine, full
        if stream_done is not None:
            stream_done.set()
    wf_view = torch.as_tensor(eng.result_view(_native.BUF_WAVEFUNCTION, "<c8"), device=dev).reshape(P, slots, eng.wx * eng.wy)

    # ---- BASELINE C3 as configured (N=1, after the timed region): every frame slot holds a REAL frame, the time FFT runs on them
    c3 = None
    if c3_full:
        todo = list(range(n_local, tacaw_T))
        tx0 = time.perf_counter()
        for i in range(0, len(todo), eng.frame_batch):
            chunk = todo[i:i + eng.frame_batch]
            if eng.frame_batch > 1:
                eng.build_potentials(tr.positions[chunk[0]:chunk[-1] + 1], Z, 2)
                eng.propagate_frames(chunk[0], len(chunk))
            else:
                eng.build_potential(tr.positions[chunk[0]], Z, 2)
                eng.propagate_frame(chunk[0])
        fence()
        dt_extra = time.perf_counter() - tx0
        c3 = {"frames": tacaw_T, "probes": P, "grid": n, "slices": nz,
              "frames_in_warmup_and_timed_region": n_local, "frames_after_timed_region": len(todo),
              "seconds_warmup": round(dt_warm, 3), "seconds_timed_region": round(dt, 3), "seconds_remaining_frames": round(dt_extra, 3),
              "seconds_propagate": round(dt_warm + dt + dt_extra, 3)}

    # ---- TACAW leg (N=1): time FFT over the resident frame slots; without c3_full the slots beyond the computed frames hold copies
    tacaw = None
    if tacaw_T is not None:
        if not c3_full:
            for j in range(n_local, tacaw_T):
                wf_view[:, j].copy_(wf_view[:, j % n_local])
        torch.cuda.synchronize()
        before = eng.counters()["ms_tacaw"]
        eng.tacaw()                                  # first call: first touch of the freshly allocated intensity buffer, tables
        ms_first = eng.counters()["ms_tacaw"] - before
        before = eng.counters()["ms_tacaw"]
        eng.tacaw()
        ms = eng.counters()["ms_tacaw"] - before
        tacaw = {"ms": round(ms, 3), "GBps": round(12.0 * P * tacaw_T * npix / (ms * 1e-3) / 1e9, 1), "frames": tacaw_T,
                 "probes": P, "kernel": _tacaw_kernel_name(tacaw_T, npix),
                 "algorithmic_bytes": 12.0 * P * tacaw_T * npix, "frac_of_hbm_peak": round(12.0 * P * tacaw_T * npix / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "first_call_ms": round(ms_first, 3),
                 "note": ("all slots hold real frames (c3_full)" if c3_full else
                          f"{n_local} computed frames, remaining slots filled with copies (timing is data-independent)")}
        if c3_full:
            # Parseval over the frequency axis, per probe: sum_w I[p,w,k] = T sum_t |Psi|^2 - |sum_t Psi|^2 (the mean subtraction of
            # tacaw_data.py:94 removes exactly the u = 0 bin); float64 sums over the resident arrays, one probe at a time
            inten = torch.as_tensor(eng.result_view(_native.BUF_INTENSITY, "<f4"), device=dev).reshape(P, tacaw_T, npix)
            worst = 0.0
            for pi in range(P):
                w = wf_view[pi]
                s2, s1 = 0.0, torch.zeros(npix, dtype=torch.complex128, device=dev)
                for t0_ in range(0, tacaw_T, 32):
                    blk = w[t0_:t0_ + 32]
                    s2 += float((blk.real.double() ** 2).sum() + (blk.imag.double() ** 2).sum())
                    s1 += blk.sum(dim=0, dtype=torch.complex128)
                rhs = tacaw_T * s2 - float((s1.real ** 2 + s1.imag ** 2).sum())
                lhs = float(sum(inten[pi, t0_:t0_ + 32].sum(dtype=torch.float64) for t0_ in range(0, tacaw_T, 32)))
                worst = max(worst, abs(lhs - rhs) / abs(rhs))
                del s1
            # value check at full size: 96 time series (probe, pixel) drawn at random, |fftshift fft(x - mean)|^2 in float64 on the host
            # against the device's intensity (tacaw_data.py:94-104); per-series relative L2 error
            rng_c = np.random.default_rng(7)
            sel = [(int(rng_c.integers(P)), int(rng_c.integers(npix))) for _ in range(96)]
            series = np.stack([wf_view[pi, :, kk].cpu().numpy() for pi, kk in sel]).astype(np.complex128)
            got_i = np.stack([inten[pi, :, kk].cpu().numpy() for pi, kk in sel]).astype(np.float64)
            want_i = np.abs(np.fft.fftshift(np.fft.fft(series - series.mean(axis=1, keepdims=True), axis=1), axes=1)) ** 2
            series_err = float(np.max(np.linalg.norm(got_i - want_i, axis=1) / np.linalg.norm(want_i, axis=1)))
            # known answer: the synthetic trajectory's atoms oscillate at 10, 25 and 40 THz (pyslice_amd/synthetic.py)
            spec = np.asarray(eng.tacaw_spectrum()).reshape(P, tacaw_T).sum(axis=0)
            freqs = np.fft.fftshift(np.fft.fftfreq(tacaw_T, tr.timestep))
            pos = freqs > 0
            fpos, spos = freqs[pos], spec[pos]
            loc = [i for i in range(1, len(spos) - 1) if spos[i] > spos[i - 1] and spos[i] >= spos[i + 1]]
            loc.sort(key=lambda i: -spos[i])
            c3.update({"tacaw_ms": round(ms, 3), "tacaw_first_call_ms": round(ms_first, 3),
                       "seconds_total": round(c3["seconds_propagate"] + ms_first * 1e-3, 3),
                       "parseval_rel": float(worst), "time_fft_worst_series_rel_l2": series_err, "time_fft_series_checked": len(sel),
                       "spectrum_peak_THz": [round(float(fpos[i]), 2) for i in sorted(loc[:3])],
                       "frequency_resolution_THz": round(1.0 / (tacaw_T * tr.timestep), 4),
                       "expected_peaks_THz": [10.0, 25.0, 40.0],
                       "resident_GB": round((8.0 + 4.0) * P * tacaw_T * npix / 1e9, 1),
                       "baseline_md_target_seconds": 27.5})
            del inten

    # ---- end-of-run exchanges of the sharded path (N>1), on the frames just computed (run AFTER the bench line is assembled,
    # under a watchdog: see below)
    def run_exchange():
        if os.environ.get("MSL_BENCH_TEST_STALL") == str(rank):      # test hook: this rank never reaches the exchange
            time.sleep(3600)

        def sync():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

        def timed(fn):
            sync()
            t1 = time.perf_counter()
            out = fn()
            sync()
            return out, (time.perf_counter() - t1) * 1e3

        # equal shards of the timed frames; a total the ranks can hold: at most ~100 GB gathered on rank 0
        per_rank = max(1, min(frames_timed_local, int(100e9 / (8.0 * P * npix * world))))
        T_x = per_rank * world
        first = n_local - frames_timed_local
        local = wf_view[:, first:first + per_rank]
        full, ms_gather = timed(lambda: D.gather_frames(local, T_x, dst=0))
        del full
        mine, ms_a2a = timed(lambda: D.frames_to_probes(local, T_x))
        out = torch.empty(mine.shape, dtype=torch.float32, device=dev)

        def do_tacaw():
            if mine.shape[0] > 0 and T_x >= 2:
                eng.tacaw(mine.data_ptr(), out.data_ptr(), mine.shape[0], T_x, npix)
                eng.synchronize()
        _, ms_t = timed(do_tacaw)
        inten, ms_gp = timed(lambda: D.gather_probes(out, P, dst=0))
        del inten, mine, out
        shard_b = 8.0 * P * per_rank * npix
        return {"backend": backend, "frames": T_x, "frames_per_rank": per_rank,
                "gather_frames": round(ms_gather, 3), "frames_to_probes": round(ms_a2a, 3), "tacaw": round(ms_t, 3),
                "gather_probes": round(ms_gp, 3),
                "gather_frames_GBps_into_rank0": round(shard_b * (world - 1) / (ms_gather * 1e-3) / 1e9, 1),
                "frames_to_probes_GBps_out_per_rank": round(shard_b * (world - 1) / world / (ms_a2a * 1e-3) / 1e9, 1),
                "bytes_per_rank_shard": shard_b}

    if rank == 0:
        steps_total = frames_timed * P * nz
        value = steps_total / dt
        # Slice-loop kernels.  Every launch streams the P wave functions of the frame once: 8 B read + 8 B write per
        # pixel = 16 B x nx x ny x P algorithmic bytes per launch.  The default one-pass loop needs ONE
        # such launch per slice (16 B/pixel/slice-step: the separable Fresnel propagator lets a pass finish one
        # propagation and start the next); the two-pass loop (MSL_SLICE_PATH=2) needs two (SURVEY 8d's 32 B).
        rows, cols = (ctr["row_launches"], ctr["ms_row"]), (ctr["col_launches"], ctr["ms_col"])
        name, (cnt, ms) = max((("pass_along_y", rows), ("pass_along_x", cols)), key=lambda kv: kv[1][1])
        bytes_per_launch = 16.0 * npix * P * eng.frame_batch
        passes_per_slice = (rows[0] + cols[0]) * eng.frame_batch / float(frames_timed_local * nz) if nz and frames_timed_local else 0.0
        roof = None
        if cnt:
            avg_s = ms * 1e-3 / cnt
            ach = bytes_per_launch / avg_s / 1e9
            traffic, tsrc = None, None
            tpath = os.path.join(REPO, "profiles", "pmc_traffic_current.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                sys.path.insert(0, os.path.join(REPO, "tools"))
                from update_traffic import csrc_sha256
                if (tj.get("grid") == n and tj.get("probes") == P and tj.get("frame_batch", 1) == eng.frame_batch
                        and round(passes_per_slice) == tj.get("passes_per_slice")):
                    if tj.get("csrc_sha256") == csrc_sha256():
                        traffic = tj.get("hbm_bytes_per_launch")
                        tsrc = (f"profiles/pmc_traffic_current.json: rocprofv3 --pmc passes of this command on these very kernel sources "
                                f"(sha256 {tj['csrc_sha256'][:12]}, kernel {tj.get('kernel')}); committed, not re-measured in this run")
                    else:
                        tsrc = "profiles/pmc_traffic_current.json is stale (kernel sources changed since the counters were taken): traffic withheld"
            loop_32 = (32.0 * npix * P * nz * frames_timed_local / (ctr["ms_slice_kernels"] * 1e-3) / 1e9) if ctr["ms_slice_kernels"] else None
            roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                    "avg_launch_us": round(avg_s * 1e6, 2), "launches": int(cnt),
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "passes_per_slice": round(passes_per_slice, 3),
                    "pass_along_y_GBps": round(bytes_per_launch * rows[0] / (rows[1] * 1e-3) / 1e9, 1) if rows[1] else None,
                    "pass_along_x_GBps": round(bytes_per_launch * cols[0] / (cols[1] * 1e-3) / 1e9, 1) if cols[1] else None,
                    "slice_loop_GBps_on_32B_per_slice_step_basis": round(loop_32, 1) if loop_32 else None,
                    "frac_on_32B_basis": round(loop_32 / HBM_PEAK_GBS, 4) if loop_32 else None}
        per_gpu = f"{a.steps} MD frames per GPU" if not strong else f"{a.steps} steps of {a.frames_per_step} MD frames sharded over the ranks"
        out = {
            "metric": "slice-steps/sec (probes x frames x slices / s), potential + slice loop + exit FFT",
            "value": round(value, 1), "unit": "slice-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "complex64 (f32)", "data": "synthetic",
            "config": {"workload": f"{P}-probe STEM grid, {n}x{n} grid, {nz} slices, {per_gpu} "
                                   f"(BASELINE configs[2] per-frame work), {tr.n_atoms} atoms, 30 mrad, 100 keV",
                       "grid": n, "slices": nz, "probes": P, "frames_timed": frames_timed,
                       "parallelism": f"frames x{world}", "backend": backend if world > 1 else None,
                       "world_size_checked": world, "frame_slots": slots, "frame_batch": eng.frame_batch},
            "breakdown_ms_per_step": {"potential": round(ctr["ms_potential"] / a.steps, 3),
                                      "slice_loop_and_epilogue": round(ctr["ms_propagate"] / a.steps, 3)},
            "roofline": roof,
        }
        if tacaw is not None:
            out["tacaw"] = tacaw
        if c3 is not None:
            out["c3_full"] = c3
        if stream_ms is not None:
            out["stream_ms"] = stream_ms
            out["config"]["streaming_tacaw"] = {"ring": slots, "k_window": kw, "k_bin": a.k_bin, "frames_folded": world * n_local}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], V = cpu_baseline(n, nz, P, a.aperture, a.cpu_slices)
            try:
                out["cpu_baseline_allcores"] = cpu_baseline_allcores(n, nz, P, a.aperture, a.cpu_slices, V)
            except Exception as exc:  # pragma: no cover  (scipy missing: the single-thread leg stands alone)
                out["cpu_baseline_allcores"] = {"error": repr(exc)}
    else:
        out = None
    if world > 1 and not a.no_exchange and not stream:
        # The timed region is over and its line is complete.  The exchanges move tens of GB between the ranks over a backend
        # this build could not exercise on real multi-GPU hardware: if they stall, every rank's watchdog ends its process after
        # `--exchange-timeout` seconds and rank 0 still prints the line, with the failure recorded instead of the timings.
        import threading
        finished = threading.Event()
        emit = threading.Lock()             # the line is printed once, by whoever gets here first

        def watchdog():
            if not finished.wait(a.exchange_timeout):
                with emit:
                    if finished.is_set():                # the exchanges came in at the last moment: the main thread has the line
                        return
                    if rank == 0:
                        out["exchange_ms"] = {"error": f"end-of-run exchanges did not finish within {a.exchange_timeout} s"}
                        print(json.dumps(out), flush=True)
                    else:
                        time.sleep(2.0)                  # rank 0 prints first: the launcher ends the job at the first failing rank
                    os._exit(EXIT_EXCHANGE_STALLED)      # the measurement is out, but the run did NOT end well: callers must see it
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            ex = run_exchange()
        except Exception as exc:                     # a failing exchange must not cost the measured line
            ex = {"error": repr(exc)}
        with emit:
            finished.set()
            if rank == 0:
                out["exchange_ms"] = ex
                print(json.dumps(out), flush=True)
    elif rank == 0:
        print(json.dumps(out), flush=True)
    del wf_view
    eng.close()
    if world > 1:
        import threading
        t = threading.Timer(60.0, lambda: os._exit(EXIT_EXCHANGE_STALLED))      # the line is out: a peer that died must not keep this rank in the barrier
        t.daemon = True
        t.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
        t.cancel()


def _tacaw_kernel_name(T, npix):
    """the kernel msl_tacaw picks for T frames (mslice.hip: msl_tacaw)"""
    n = T
    for p in (2, 3, 5):
        while n % p == 0:
            n //= p
    smooth = n == 1
    if smooth and 16 <= T <= 128:
        return "per-lane mixed-radix register FFT (time_direct_kernel)"
    if smooth and 128 < T <= 512 and 65 * npix * 8 < 2 ** 32 and any(T % L == 0 and T // L <= 128 and (L <= 4 or T // L <= 100) for L in (2, 3, 4, 5, 6)):
        return "mixed-radix register FFT split over the waves of a workgroup (time_split_kernel)"
    if smooth and 512 < T <= 1024:
        for L in (8, 6):
            if T % L == 0 and T // L <= 128:
                if (T // L + (T // L + 1) // 2) * npix * 8 < 2 ** 32:
                    return "mixed-radix register FFT split over the waves of a workgroup, two blocks per wave (time_split_kernel)"
                break
    if T == 1024 and npix % 16 == 0 and npix >= 32:
        return "four-step time FFT (32 x 32 lanes x registers)"
    return "chirp-z on the register FFTs" if T <= 512 else "generic LDS kernel"


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    run(a)


if __name__ == "__main__":
    main()
