"""Cost of the streaming TACAW fold (msl_tacaw_stream_push, O(T F) per stored pixel) at BASELINE C5's numbers: T = 1024 frames, all
1024 frequency bins, 128 x 128 stored pixels (k-window 512 x 512 binned 4 x 4), 16 probes per launch sequence (one GPU's
frame), for several ring lengths.  The ring is filled with random spectra (the fold's time does not depend on the data) and
pushed T / ring times; wall time around the pushes with the stream drained before and after.

    python tools/fold_cost.py [--probes 16] [--frames 1024] [--pixels 128] [--rings 8,32,64]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--probes", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--pixels", type=int, default=128)
    ap.add_argument("--rings", default="8,32,64")
    ap.add_argument("--bins", type=int, default=0, help="frequency bins kept (0 = all)")
    a = ap.parse_args()
    from pyslice_amd import _native
    n, P, T = a.pixels, a.probes, a.frames
    rng = np.random.default_rng(0)
    for ring in [int(r) for r in a.rings.split(",")]:
        eng = _native.Engine(n, n, 1, 0.1, 0.1, 0.5, 0.037, 1e-3, n_probes=P, n_frames=ring)
        for i in range(ring):
            eng.upload_frame(i, (rng.standard_normal((P, n, n)) + 1j * rng.standard_normal((P, n, n))).astype(np.complex64))
        F = a.bins if a.bins else T
        eng.tacaw_stream_begin(T, None if not a.bins else np.arange(F))
        eng.tacaw_stream_set_reference(slot=0)
        eng.tacaw_stream_push(0, ring, 0)          # warm-up tile
        eng.synchronize()
        t0 = time.perf_counter()
        tiles = 0
        for s in range(ring, T, ring):
            eng.tacaw_stream_push(0, min(ring, T - s), s)
            tiles += 1
        eng.synchronize()
        dt = time.perf_counter() - t0
        eng.tacaw_stream_finish(False)
        eng.close()
        frames = tiles * ring
        macs = P * n * n * F * frames
        acc_bytes = 16.0 * P * n * n * F * tiles + 8.0 * P * n * n * frames * ((F + 15) // 16)
        print(f"ring {ring:3d}: {dt / tiles * 1e3:8.3f} ms per tile of {ring} frames = {dt / frames * 1e3:7.4f} ms per frame "
              f"({P} probes x {n}x{n} px x {F} bins of {T}); {8 * macs / dt / 1e12:6.2f} TFLOP/s, "
              f"accumulator + tile traffic {acc_bytes / dt / 1e12:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
