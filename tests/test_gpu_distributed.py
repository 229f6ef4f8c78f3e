"""Two ranks on one MI355X (gloo rehearsal of the one-process-per-GPU path): frames are sharded over the ranks, each
rank propagates its shard on the device, WFData is gathered on rank 0, and TACAWData re-shards frames -> probes with
the all-to-all, runs the device time FFT per rank and gathers the intensities.  RCCL itself needs one GPU per rank and is
exercised by `bench.py --gpus N`; the exchange logic, shard arithmetic and device plumbing are the same code, and the last test
runs the grouped send / receive on RCCL with the rank as its own peer."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pyslice_amd as ps
        from pyslice_amd.synthetic import synthetic_trajectory
        tr = synthetic_trajectory(256, 5, 5, density=0.05, seed=41)          # 5 frames -> shards of 3 and 2
        pp = [(12.0, 12.0), (3.0, 20.0), (17.5, 6.25)]
        out = {}
        calc = ps.MultisliceCalculator(device=0, progress=False, gather="rank0", output="device")
        calc.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
        torch.cuda.set_device(0)
        torch.zeros(1, device="cuda")                      # create the context before the allocator statistics are reset
        torch.cuda.reset_peak_memory_stats(0)
        base = torch.cuda.memory_allocated(0)
        wf = calc.run()
        # torch-side device memory of the gather: the (P,T,nx,ny) result on rank 0 and nothing else (the shard itself
        # lives in the engine's own buffer); the old implementation held a padded copy, per-rank buffers and a cat
        full_bytes, shard_bytes = 3 * 5 * 256 * 256 * 8, 3 * 3 * 256 * 256 * 8
        peak = torch.cuda.max_memory_allocated(0) - base
        assert peak <= (full_bytes if rank == 0 else 0) + shard_bytes + (1 << 20), (rank, peak)
        if rank == 0:
            out["wf"] = wf.wavefunction_data.cpu().numpy()
        else:
            assert wf.wavefunction_data is None
        calc2 = ps.MultisliceCalculator(device=0, progress=False, gather="none")
        calc2.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
        tac = ps.TACAWData(calc2.run())
        if rank == 0:
            out["intensity"] = tac.intensity.cpu().numpy()
            out["frequencies"] = tac.frequencies
        else:
            assert tac.intensity is None
        # the same with a k-window: shards, all-to-all and gather carry (.., 64, 96) spectra; reductions on rank 0
        calc3 = ps.MultisliceCalculator(device=0, progress=False, gather="none", k_window=(64, 96))
        calc3.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
        tac3 = ps.TACAWData(calc3.run())
        if rank == 0:
            out["intensity_win"] = tac3.intensity.cpu().numpy()
            out["spectrum_win"] = tac3.spectrum(None)
            out["diffraction_win"] = tac3.diffraction(1)
        else:
            assert tac3.intensity is None
        # an odd pixel count (21 x 19 = 399 pixels at a pitch of 416): the shards are STRIDED views of the library's buffers
        for tag, gather in (("odd", "rank0"), ("odd_none", "none")):
            calc4 = ps.MultisliceCalculator(device=0, progress=False, gather=gather, output="device", k_window=(21, 19))
            calc4.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
            wf4 = calc4.run()
            assert calc4._engine.result_pitch() == 416
            if gather == "rank0":
                if rank == 0:
                    out["wf_odd"] = wf4.wavefunction_data.cpu().numpy()
                continue
            tac4 = ps.TACAWData(wf4)
            if rank == 0:
                out["intensity_odd"] = tac4.intensity.cpu().numpy()
                out["spectrum_odd"] = tac4.spectrum(2)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_frame_sharding_gather_and_tacaw():
    import torch.multiprocessing as mp
    from conftest import rel_l2
    from oracle import multislice_oracle as orc
    from pyslice_amd.synthetic import synthetic_trajectory
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    res = {}
    while len(res) < 2:
        try:
            r, out = q.get(timeout=2)
            res[r] = out
        except queue.Empty:        # a rank that died will never report: fail now instead of waiting out the timeout
            assert all(p.exitcode in (None, 0) for p in procs), [p.exitcode for p in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    tr = synthetic_trajectory(256, 5, 5, density=0.05, seed=41)
    pp = [(12.0, 12.0), (3.0, 20.0), (17.5, 6.25)]
    want = orc.run_frames(tr.box_matrix, tr.positions, tr.atom_types, 30.0, 100e3, pp)["wavefunction_data"]
    got = res[0]["wf"]
    assert got.shape == want.shape
    assert rel_l2(got, want) < 1e-4
    f, inten = orc.tacaw(want, np.arange(5) * tr.timestep)
    assert np.allclose(res[0]["frequencies"], f)
    assert rel_l2(res[0]["intensity"], inten) < 2e-4
    win = inten[:, :, 128 - 32:128 + 32, 128 - 48:128 + 48]
    assert res[0]["intensity_win"].shape == win.shape
    assert np.linalg.norm(res[0]["intensity_win"] - win) / np.linalg.norm(win) < 2e-4
    assert rel_l2(res[0]["spectrum_win"], win.sum(axis=(2, 3)).mean(axis=0)) < 2e-4
    assert rel_l2(res[0]["diffraction_win"], win[1].sum(axis=0)) < 2e-4
    odd = (slice(128 - 10, 128 - 10 + 21), slice(128 - 9, 128 - 9 + 19))
    assert rel_l2(res[0]["wf_odd"], want[:, :, odd[0], odd[1]]) < 1e-4
    assert rel_l2(res[0]["intensity_odd"], inten[:, :, odd[0], odd[1]]) < 2e-4
    assert rel_l2(res[0]["spectrum_odd"], inten[2][:, odd[0], odd[1]].sum(axis=(1, 2))) < 2e-4


def _stream_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pyslice_amd as ps
        from pyslice_amd.synthetic import synthetic_trajectory
        torch.cuda.set_device(0)
        out = {}
        # 11 frames -> shards of 4, 4, 3 (world 3) or 6, 5 (world 2); ring of 2 frame slots: several tiles per rank
        tr = synthetic_trajectory(128, 3, 11, density=0.08, seed=52)
        pp = [(6.0, 6.0), (2.0, 10.0), (9.5, 3.25)]
        for tag, kw, args in (("all", dict(), dict()), ("win", dict(k_window=(64, 32), k_bin=(2, 2), frame_batch=2), dict(freq_window=(0.0, 60.0)))):
            calc = ps.MultisliceCalculator(device=0, progress=False, stream_tile=2, gather="rank0", **kw)
            calc.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
            tac = calc.run_streaming_tacaw(**args)
            if rank == 0:
                out[tag] = dict(intensity=tac.intensity.cpu().numpy(), total=tac.total_diffraction, frequencies=tac.frequencies,
                                bins=tac.frequency_bins, spectrum=tac.spectrum(1), diffraction=tac.diffraction(None))
            else:
                assert tac.intensity is None and tac.total_diffraction is None
        # every rank keeps its probe shard (gather="none"): P = 3 over the ranks
        calc = ps.MultisliceCalculator(device=0, progress=False, stream_tile=3, gather="none")
        calc.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
        tac = calc.run_streaming_tacaw()
        # the shard's TACAWData indexes ITS probes: spectrum of every local probe, spectrum image, probe-mean diffraction
        n_loc = tac.probe_range[1] - tac.probe_range[0]
        assert len(tac.probe_positions) == n_loc
        out["shard"] = (tac.probe_range, tac.intensity.cpu().numpy(),
                        [tac.spectrum(b) for b in range(n_loc)], tac.diffraction(None) if n_loc else None)
        if n_loc:
            with pytest.raises(ValueError):
                tac.spectrum(n_loc)                       # one past the shard
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_frame_sharded_streaming_tacaw(world):
    """BASELINE C5's path in small: the (P,T,nx,ny) array exists nowhere -- every rank folds its own frames (global time
    indices, common reference pattern) through a ring of frame slots, the partial sums are reduce-scattered over probes, each
    rank finishes its probes and rank 0 gathers the intensities.  Against the oracle's TACAW of the full array
    (tacaw_data.py:89-104)."""
    import torch.multiprocessing as mp
    from conftest import rel_l2
    from oracle import multislice_oracle as orc
    from pyslice_amd.distributed import shard_bounds
    from pyslice_amd.synthetic import synthetic_trajectory
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    res = {}
    while len(res) < world:
        try:
            r, out = q.get(timeout=2)
            res[r] = out
        except queue.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), [p.exitcode for p in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    tr = synthetic_trajectory(128, 3, 11, density=0.08, seed=52)
    pp = [(6.0, 6.0), (2.0, 10.0), (9.5, 3.25)]
    full = orc.run_frames(tr.box_matrix, tr.positions, tr.atom_types, 30.0, 100e3, pp)["wavefunction_data"]
    f, inten = orc.tacaw(full, np.arange(11) * tr.timestep)
    a = res[0]["all"]
    assert np.allclose(a["frequencies"], f) and a["intensity"].shape == inten.shape
    assert rel_l2(a["intensity"], inten) < 2e-4
    assert rel_l2(a["total"], inten.sum(axis=1)) < 2e-4
    assert rel_l2(a["spectrum"], inten[1].sum(axis=(1, 2))) < 2e-4
    assert rel_l2(a["diffraction"], inten.sum(axis=1).mean(axis=0)) < 2e-4
    # k-window 64 x 32 binned 2 x 2, two frames per launch, a frequency window
    win = full[:, :, 64 - 32:64 + 32, 64 - 16:64 + 16, 0].reshape(3, 11, 32, 2, 16, 2).sum(axis=(3, 5))
    fw, iw = orc.tacaw(win[..., None], np.arange(11) * tr.timestep)
    sel = np.nonzero((fw >= 0.0) & (fw <= 60.0))[0]
    w = res[0]["win"]
    assert np.array_equal(w["bins"], sel) and np.allclose(w["frequencies"], fw[sel])
    assert rel_l2(w["intensity"], iw[:, sel]) < 2e-4
    assert rel_l2(w["total"], iw.sum(axis=1)) < 2e-4
    for r in range(world):
        (p0, p1), shard, spectra, diff = res[r]["shard"]
        assert (p0, p1) == shard_bounds(3, world, r)
        if p1 > p0:
            assert rel_l2(shard, inten[p0:p1]) < 2e-4
            for b, sp in enumerate(spectra):
                assert rel_l2(sp, inten[p0 + b].sum(axis=(1, 2))) < 2e-4
            assert rel_l2(diff, inten[p0:p1].sum(axis=1).mean(axis=0)) < 2e-4


def test_bench_two_ranks_gloo_rehearsal(tmp_path):
    """`python bench.py --gpus 2` starts its own two ranks (both on this one GPU, gloo instead of RCCL), reports
    n_gpus = 2 and times the end-of-run exchanges."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MSL_BENCH_BACKEND"] = "gloo"
    for extra in ([], ["--scaling", "strong", "--frames-per-step", "4"]):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--grid", "256", "--slices", "6",
                            "--probes", "4", "--steps", "2", "--warmup", "1"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, r.stdout
        j = json.loads(line[0])
        assert j["n_gpus"] == 2 and j["config"]["world_size_checked"] == 2 and j["scaling"] == ("strong" if extra else "weak")
        ex = j["exchange_ms"]
        assert ex["backend"] == "gloo" and ex["gather_frames"] > 0 and ex["frames_to_probes"] > 0 and ex["gather_probes"] > 0
        assert j["config"]["frames_timed"] == (8 if extra else 4)
        assert j["value"] > 0 and j["roofline"]["launches"] > 0


def test_bench_streaming_tacaw_one_and_two_ranks():
    """`bench.py --stream`: every frame is folded into the streaming TACAW inside the timed region; with two ranks (gloo rehearsal
    on the one GPU) each folds its own frames with global time indices and the partial sums are reduce-scattered, finished and
    gathered after the timed region.  Frequency-integrated intensity: positive, and equal for the same frames split differently
    is not testable here (every rank synthesises its own frames) -- the physics is covered by test_ranks_frame_sharded_streaming_tacaw."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MSL_BENCH_BACKEND"] = "gloo"
    for gpus in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(gpus), "--grid", "256", "--slices", "6", "--probes", "3",
                            "--steps", "5", "--warmup", "2", "--stream", "--stream-tile", "3", "--k-window", "64", "--k-bin", "2",
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, r.stdout
        j = json.loads(line[0])
        sm = j["stream_ms"]
        assert j["n_gpus"] == gpus and sm["frames"] == 7 * gpus and sm["stored_pixels"] == 32 * 32 and sm["ring"] == 3
        assert sm["intensity_sum"] > 0 and j["value"] > 0 and j["roofline"]["launches"] > 0
        assert j["config"]["streaming_tacaw"]["frames_folded"] == 7 * gpus


def test_bench_line_survives_a_stalled_exchange():
    """the end-of-run exchanges run after the bench line is assembled, under a watchdog: a rank that never arrives costs the
    exchange timings, not the measurement -- the line is printed once, with the failure in it -- and every rank exits, with
    the non-zero code that tells the caller the run did not end well"""
    import json
    import subprocess
    import sys
    import time
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MSL_BENCH_BACKEND"] = "gloo"
    env["MSL_BENCH_TEST_STALL"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--grid", "256", "--slices", "6", "--probes", "4",
                        "--steps", "2", "--warmup", "1", "--exchange-timeout", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 200
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "did not finish" in j["exchange_ms"]["error"]


def test_bench_single_gpu_line_and_c3_full_block():
    """`python bench.py` at N = 1 in small: one JSON line with the roofline and cpu_baseline objects, and the c3_full block --
    after the timed region the remaining frame slots are filled with REAL frames, the time FFT runs on them, Parseval over the
    frequency axis holds, sampled time series agree with a float64 FFT and the trajectory's phonon peaks are found (timestep
    0.005 ps, 64 frames: 3.1 THz per bin)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--grid", "256", "--slices", "8", "--probes", "4", "--steps", "8",
                        "--warmup", "4", "--tacaw-frames", "64", "--cpu-slices", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    j = json.loads(line[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] == 1
    c3 = j["c3_full"]
    assert c3["frames"] == 64 and c3["frames_in_warmup_and_timed_region"] == 12 and c3["frames_after_timed_region"] == 52
    assert c3["parseval_rel"] < 1e-4 and c3["time_fft_worst_series_rel_l2"] < 2e-4
    assert j["tacaw"]["frames"] == 64 and "real frames" in j["tacaw"]["note"]
    assert len(c3["spectrum_peak_THz"]) == 3 and min(abs(p - 25.0) for p in c3["spectrum_peak_THz"]) <= c3["frequency_resolution_THz"]


_RCCL_LOOPBACK = r'''
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from pyslice_amd import _native, distributed as D
# a library-owned result buffer (hipMalloc of libmslice, not torch's allocator) on an odd pixel count: strided zero-copy view
P, T, nx, ny = 3, 4, 9, 7
eng = _native.Engine(nx, ny, 1, 0.1, 0.1, 0.5, 0.037, 1e-3, n_probes=P, n_frames=T, device=0)
rng = np.random.default_rng(5)
frames = (rng.standard_normal((P, T, nx, ny)) + 1j * rng.standard_normal((P, T, nx, ny))).astype(np.complex64)
for t in range(T):
    eng.upload_frame(t, frames[:, t])
view = torch.as_tensor(eng.result_view(_native.BUF_WAVEFUNCTION, "<c8"), device="cuda")
assert not view.is_contiguous()
# the grouped point-to-point launch of every exchange plan (distributed._run: batch_isend_irecv = ncclGroupStart/End), here with
# the rank as its own peer: probe 1's frames [1, 3) out of the library buffer, into a torch tensor
dst = torch.zeros((2, nx, ny, 2), dtype=torch.float32, device="cuda")
src_real, cplx, dev = D._as_real(view)
ops = [D.Op("recv", 0, 0, dst.numel(), 0, ("dst",)), D.Op("send", 0, 0, dst.numel(), 0, ("src",))]
D._run(ops, lambda w: dst if w[0] == "dst" else D._chunk(src_real[1, 1:3]))
torch.cuda.synchronize()
got = torch.view_as_complex(dst).cpu().numpy()
assert np.array_equal(got, frames[1, 1:3]), "RCCL self-exchange out of the library buffer"
# ... and INTO the library buffer (receives land in the destination, no staging): frame slot 0 of probe 2, one dense image row at a time
new = torch.view_as_real(torch.from_numpy(frames[0, 3]).cuda().contiguous())
rows = torch.view_as_real(view)[2, 0]
assert rows.is_contiguous()                     # one image = nx*ny pixels at the start of its pitch
ops = [D.Op("recv", 0, 0, rows.numel(), 0, ("into",)), D.Op("send", 0, 0, rows.numel(), 0, ("from",))]
D._run(ops, lambda w: rows if w[0] == "into" else new)
torch.cuda.synchronize()
assert np.array_equal(eng.frame(0)[2], frames[0, 3]) and np.array_equal(eng.frame(0)[1], frames[1, 0])
# the collectives bench.py's timing protocol uses
t = torch.tensor([3.5], device="cuda")
dist.barrier()
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 3.5
print("rccl-loopback-ok", torch.cuda.nccl.version())
dist.destroy_process_group()
'''


def test_rccl_executes_the_grouped_exchange_on_library_buffers():
    """RCCL needs one GPU per rank, so the multi-rank tests above run on gloo.  What one GPU can show of the real transport: a
    world-size-1 "nccl" group runs distributed._run's grouped send/receive (the rank as its own peer) out of and into the library's
    own hipMalloc'ed result buffer through the strided zero-copy view, plus the barrier / max-reduce of bench.py's timing."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), PYTHONPATH=repo)
    r = subprocess.run([sys.executable, "-c", _RCCL_LOOPBACK], env=env, capture_output=True, text=True, timeout=240, cwd=repo)
    assert r.returncode == 0 and "rccl-loopback-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
