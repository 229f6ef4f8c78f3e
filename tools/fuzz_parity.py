"""Randomised parity sweep through the C ABI: grid shapes that mix every kernel family (register kernels for 256/1024,
2R^2 kernels for 512/2048, generic radix-2..13 lengths, Bluestein lengths), slice counts, probe counts and k-windows,
against the complex128 oracle.  Potentials are uploaded (synthetic, atom-like), so a case costs the oracle only its
slice loop.  usage: python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import multislice_oracle as orc
from pyslice_amd import _native

LENGTHS = [256, 512, 1024, 2048, 64, 96, 128, 160, 200, 243, 250, 320, 343, 360, 384, 448, 500, 101, 97, 127, 330, 1000, 26, 39]


def one(rng, max_pix):
    while True:
        nx, ny = int(rng.choice(LENGTHS)), int(rng.choice(LENGTHS))
        if nx * ny <= max_pix:
            break
    nz = int(rng.integers(1, 13))
    P = int(rng.integers(1, 5))
    T = int(rng.integers(1, 3))
    dx, dy, dz = 0.1, float(rng.choice([0.1, 0.09])), float(rng.choice([0.5, 0.7]))
    eV = float(rng.choice([60e3, 100e3, 300e3]))
    mrad = float(rng.choice([0.0, 15.0, 30.0]))
    window = None
    if rng.random() < 0.4:
        window = (int(rng.integers(1, nx + 1)), int(rng.integers(1, ny + 1)))
        if rng.random() < 0.5:
            window = (max(32, window[0] // 32 * 32) if nx >= 32 else window[0], max(32, window[1] // 32 * 32) if ny >= 32 else window[1])
            window = (min(window[0], nx), min(window[1], ny))
    xs, ys, zs = np.arange(nx) * dx, np.arange(ny) * dy, np.arange(nz) * dz
    V = (rng.random((T, nx, ny, nz)) ** 10 * 3000.0).astype(np.float32)
    pp = [(float(rng.random() * xs[-1]), float(rng.random() * ys[-1])) for _ in range(P)]
    eng = _native.Engine(nx, ny, nz, dx, dy, dz if nz > 1 else 0.5, orc.wavelength(eV), orc.interaction_sigma(eV), n_probes=P, n_frames=T,
                         window=window)
    eng.set_probes(mrad, pp)
    for t in range(T):
        eng.upload_potential(np.moveaxis(V[t], 2, 0))
        eng.propagate_frame(t)
    got = eng.wavefunction()
    eng.upload_potential(np.moveaxis(V[0], 2, 0))
    eng.propagate()
    got_exit = eng.exit_waves()
    eng.close()
    probes = orc.batched_probes(orc.probe_array(xs, ys, mrad, eV), xs, ys, pp)
    errs = []
    for t in range(T):
        ex = orc.propagate(probes, V[t].astype(np.float64), xs, ys, zs, eV)
        if t == 0:
            errs.append(np.linalg.norm(got_exit - ex) / np.linalg.norm(ex))
        want = orc.diffraction(ex)
        scale = np.linalg.norm(want)
        if window:
            x0, y0 = nx // 2 - window[0] // 2, ny // 2 - window[1] // 2
            want = want[:, x0:x0 + window[0], y0:y0 + window[1]]
            scale *= np.sqrt(window[0] * window[1] / (nx * ny))
        errs.append(np.linalg.norm(got[:, t] - want) / max(np.linalg.norm(want), scale))
    return (nx, ny, nz, P, T, window, mrad, eV), max(errs)


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst, bad, t0 = 0.0, 0, time.time()
    for c in range(n_cases):
        cfg, err = one(rng, max_pix=2 ** 21 if c % 10 else 2 ** 22)
        worst = max(worst, err)
        flag = "" if err < 1e-4 else "   <-- FAIL"
        bad += err >= 1e-4
        print(f"{c:3d} nx={cfg[0]:4d} ny={cfg[1]:4d} nz={cfg[2]:2d} P={cfg[3]} T={cfg[4]} window={cfg[5]} mrad={cfg[6]:g} eV={cfg[7]:g}: rel-L2 {err:.2e}{flag}", flush=True)
    print(f"{n_cases} cases, worst rel-L2 {worst:.2e}, {bad} above 1e-4, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
