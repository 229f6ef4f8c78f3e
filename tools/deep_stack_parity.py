"""Error growth with slice count: exit-wave rel-L2 of the device path against the complex128 oracle on deep stacks
(BASELINE C5 uses 400 slices).  Uploaded synthetic potentials (the potential build has its own parity tests), so the
oracle side is only the slice loop.  Run on the GPU box: python tools/deep_stack_parity.py"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from oracle import multislice_oracle as orc
from pyslice_amd import _native

def case(n, nz, seed=0, distinct=None):
    """distinct: number of different potential slices to synthesise (cycled along z); None = all nz different"""
    rng = np.random.default_rng(seed)
    dx = 0.1; dz = 0.5
    xs = np.arange(n) * dx; zs = np.arange(nz) * dz
    # atom-like projected potential: sparse Gaussian peaks of a few hundred V.A per slice
    V = np.zeros((n, n, nz), dtype=np.float32)
    k = np.fft.fftfreq(n, dx)
    g = np.exp(-(np.pi * 0.35) ** 2 * (k[:, None] ** 2 + k[None, :] ** 2))
    nd = nz if distinct is None else min(nz, distinct)
    for z in range(nd):
        img = np.zeros((n, n))
        idx = rng.integers(0, n, size=(max(4, n * n // 2000), 2))
        img[idx[:, 0], idx[:, 1]] = 2500.0
        V[:, :, z] = np.fft.ifft2(np.fft.fft2(img) * g).real.astype(np.float32)
    for z in range(nd, nz):
        V[:, :, z] = V[:, :, z % nd]
    eng = _native.Engine(n, n, nz, dx, dx, dz, orc.wavelength(100e3), orc.interaction_sigma(100e3), n_probes=1, n_frames=0)
    eng.upload_potential(np.moveaxis(V, 2, 0))
    eng.set_probes(30.0, [(xs[-1] / 2, xs[-1] / 2)])
    eng.propagate()
    got = eng.exit_waves()
    eng.close()
    t0 = time.time()
    probes = orc.batched_probes(orc.probe_array(xs, xs, 30.0, 100e3), xs, xs, [(xs[-1] / 2, xs[-1] / 2)])
    want = orc.propagate(probes, V.astype(np.float64), xs, xs, zs, 100e3, workers=orc.usable_cores())
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    print(f"{n}x{n} x {nz} slices: exit-wave rel-L2 {err:.2e}   (max |V| {V.max():.0f} V.A, oracle {time.time() - t0:.0f} s)", flush=True)
    return err

if __name__ == "__main__":
    for n, nz in [(256, 400), (512, 400), (1024, 400), (2048, 100), (500, 400)]:
        case(n, nz)
