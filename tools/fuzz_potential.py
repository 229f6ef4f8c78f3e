"""Randomised parity sweep of msl_build_potential against the oracle: grid shapes over every structure-factor /
inverse-FFT path (Hermitian + MFMA tiles, Hermitian VALU, full grid, register / 2R^2-length / generic / Bluestein FFTs),
1-3 species, atoms outside the box and outside every slice, both engine modes (one-pass: transmission functions with
every second slice stored transposed; keep_potential: V itself).  usage: python tools/fuzz_potential.py [n_cases] [seed]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import multislice_oracle as orc
from pyslice_amd import _native
from pyslice_amd.potentials import loadKirkland

LENGTHS = [256, 512, 64, 96, 128, 160, 200, 243, 250, 320, 101, 97, 127, 26, 39, 33, 1024, 600, 997, 513, 1100]
KIRK = None


def one(rng, max_pix):
    global KIRK
    if KIRK is None:
        KIRK = loadKirkland()
    while True:
        nx, ny = int(rng.choice(LENGTHS)), int(rng.choice(LENGTHS))
        if nx * ny <= max_pix:
            break
    nz = int(rng.integers(1, 9))
    dx, dy, dz = 0.1, float(rng.choice([0.1, 0.09])), 0.5
    xs, ys, zs = np.arange(nx) * dx, np.arange(ny) * dy, np.arange(nz) * dz
    lx, ly, lz = nx * dx, ny * dy, nz * dz
    n = int(rng.integers(1, 400))
    species = rng.choice([1, 5, 6, 7, 8, 14, 31, 42, 79], size=int(rng.integers(1, 4)), replace=False)
    Z = rng.choice(species, size=n).astype(np.int32)
    pos = rng.random((n, 3)) * [lx, ly, lz]
    pos[rng.random(n) < 0.1] += [lx * 0.7, -ly * 0.4, lz * 0.9]          # some atoms outside the box / above the last slice
    pos[rng.random(n) < 0.05, 2] = -0.3                                   # below the first slice: dropped
    eV = 100e3
    sig = orc.interaction_sigma(eV)
    V = orc.potential(xs, ys, zs, pos, Z)
    Vmax = max(np.abs(V).max(), 1e-30)
    errs = []
    for keep in (False, True):
        eng = _native.Engine(nx, ny, nz, dx, dy, dz, orc.wavelength(eV), sig, n_probes=1, n_frames=0, keep_potential=keep)
        eng.set_kirkland(KIRK)
        eng.set_slices(*orc.slice_edges(zs))
        eng.build_potential(pos, Z, 2)
        t = eng.transmission()
        errs.append(np.abs(t - np.exp(1j * sig * np.moveaxis(V, 2, 0))).max() / (sig * Vmax))   # |dt| ~ sigma |dV|
        if keep:
            errs.append(np.abs(eng.potential() - np.moveaxis(V, 2, 0)).max() / Vmax)
        eng.close()
    return (nx, ny, nz, n, list(species)), max(errs)


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst, bad, t0 = 0.0, 0, time.time()
    for c in range(n_cases):
        cfg, err = one(rng, max_pix=2 ** 18 if c % 8 else 2 ** 20)
        worst = max(worst, err)
        bad += err >= 1e-5
        print(f"{c:3d} nx={cfg[0]:4d} ny={cfg[1]:4d} nz={cfg[2]} atoms={cfg[3]:3d} Z={cfg[4]}: max|dV|/max|V| {err:.2e}{'' if err < 1e-5 else '   <-- FAIL'}", flush=True)
    print(f"{n_cases} cases, worst {worst:.2e}, {bad} above 1e-5, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
