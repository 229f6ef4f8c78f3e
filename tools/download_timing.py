"""Time msl_download of the (P,T,wx,wy) result into fresh pageable host memory: complex64 and complex128, grids with and without a pixel pitch.
    python tools/download_timing.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyslice_amd import _native
for nx, ny in ((501, 491), (512, 480), (997, 1001), (1024, 1024)):
    P, T = 2, 50
    eng = _native.Engine(nx, ny, 1, 0.1, 0.1, 0.5, 0.037, 1e-3, n_probes=P, n_frames=T)
    rng = np.random.default_rng(1)
    fr = (rng.standard_normal((P, nx, ny)) + 1j * rng.standard_normal((P, nx, ny))).astype(np.complex64)
    for t in range(T):
        eng.upload_frame(t, fr * (t + 1))
    for name, fn in (("c64", eng.wavefunction), ("c128", eng.wavefunction_c128)):
        fn(); t0 = time.perf_counter(); a = fn(); dt = time.perf_counter() - t0
        ok = np.array_equal(a[:, 7], (fr * 8).astype(a.dtype)) and np.array_equal(a[:, T - 1], (fr * T).astype(a.dtype))
        print(f"{nx}x{ny} pitch {eng.result_pitch()} (+{eng.result_pitch() - nx * ny}) {name}: {a.nbytes / 1e6:.0f} MB in {dt * 1e3:.1f} ms = {a.nbytes / dt / 1e9:.2f} GB/s  values ok: {ok}")
    eng.close()
