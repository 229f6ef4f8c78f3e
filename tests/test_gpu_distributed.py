"""Two ranks on one MI355X (gloo rehearsal of the one-process-per-GPU path): frames are sharded over the ranks, each
rank propagates its shard on the device, WFData is gathered on rank 0, and TACAWData re-shards frames -> probes with
the all-to-all, runs the device time FFT per rank and gathers the intensities.  RCCL itself needs >= 2 GPUs and is
exercised by `bench.py --gpus N`; the exchange logic, shard arithmetic and device plumbing are the same code."""
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


def test_bench_line_survives_a_stalled_exchange():
    """the end-of-run exchanges run after the bench line is assembled, under a watchdog: a rank that never arrives costs the
    exchange timings, not the measurement, and every rank still exits"""
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
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 200
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "did not finish" in j["exchange_ms"]["error"]
