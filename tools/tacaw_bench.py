"""Time msl_tacaw alone: (P, T, npix) complex64 random spectra resident on the device -> (P, T, npix) float32 intensity.

    python tools/tacaw_bench.py --frames 100 [--probes 64] [--pixels 1048576] [--reps 5]

Prints ms per call and the rate on the 12 B per (probe, frame, pixel) the transform has to move.  Used under rocprofv3 for the
counters of time_cz_kernel (tools/collect_tacaw_profile.sh)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--probes", type=int, default=64)
    ap.add_argument("--pixels", type=int, default=1024 * 1024)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--grid", type=int, nargs=2, default=None, metavar=("NX", "NY"),
                    help="time the library-owned result buffers of an NX x NY grid (images at msl_result_pitch) next to a dense "
                         "caller-held array of the same shape")
    a = ap.parse_args()
    import torch
    from pyslice_amd import _native
    if a.grid:
        a.pixels = a.grid[0] * a.grid[1]
    P, T, K = a.probes, a.frames, a.pixels
    dev = torch.device("cuda", 0)
    src = torch.view_as_complex(torch.randn((P, T, K, 2), dtype=torch.float32, device=dev))
    dst = torch.empty((P, T, K), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    eng = _native.Engine(2, 2, 1, 1.0, 1.0, 1.0, 1.0, 0.0, n_probes=1, n_frames=0, device=0)
    eng.tacaw(src.data_ptr(), dst.data_ptr(), P, T, K)          # tables, warm-up
    ms = []
    for _ in range(a.reps):
        before = eng.counters()["ms_tacaw"]
        eng.tacaw(src.data_ptr(), dst.data_ptr(), P, T, K)
        ms.append(eng.counters()["ms_tacaw"] - before)
    best = min(ms)
    print(f"T={T} P={P} npix={K}: {best:.3f} ms (min of {a.reps}; all: {' '.join('%.2f' % m for m in ms)}) = "
          f"{12.0 * P * T * K / best / 1e6:.0f} GB/s = {12.0 * P * T * K / best / 1e6 / 8000:.3f} of the HBM peak")
    eng.close()
    if a.grid:
        nx, ny = a.grid
        own = _native.Engine(nx, ny, 1, 0.1, 0.1, 0.5, 0.037, 1e-3, n_probes=P, n_frames=T, device=0)
        v = torch.as_tensor(own.result_view(_native.BUF_WAVEFUNCTION, "<c8"), device=dev)
        v.copy_(src.reshape(P, T, nx, ny))
        torch.cuda.synchronize()
        own.tacaw()
        ms = []
        for _ in range(a.reps):
            before = own.counters()["ms_tacaw"]
            own.tacaw()
            ms.append(own.counters()["ms_tacaw"] - before)
        best = min(ms)
        got = torch.as_tensor(own.result_view(_native.BUF_INTENSITY, "<f4"), device=dev)
        same = bool(torch.equal(got.reshape(P, T, K), dst))
        print(f"  library-owned buffers, pixel pitch {own.result_pitch()} (+{own.result_pitch() - K}): {best:.3f} ms = "
              f"{12.0 * P * T * K / best / 1e6:.0f} GB/s = {12.0 * P * T * K / best / 1e6 / 8000:.3f} of the HBM peak; "
              f"intensity equal to the dense run's: {same}")
        own.close()


if __name__ == "__main__":
    main()
