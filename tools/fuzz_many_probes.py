"""Parity with many probes per launch (several work items per workgroup, probe chunks, t_k reuse) against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import multislice_oracle as orc
from pyslice_amd import _native

def case(nx, ny, nz, P, seed=0):
    rng = np.random.default_rng(seed)
    dx = dy = 0.1; dz = 0.5; eV = 100e3
    xs, ys, zs = np.arange(nx) * dx, np.arange(ny) * dy, np.arange(nz) * dz
    V = (rng.random((nx, ny, nz)) ** 10 * 3000.0).astype(np.float32)
    pp = [(float(rng.random() * xs[-1]), float(rng.random() * ys[-1])) for _ in range(P)]
    eng = _native.Engine(nx, ny, nz, dx, dy, dz, orc.wavelength(eV), orc.interaction_sigma(eV), n_probes=P, n_frames=1)
    eng.set_probes(30.0, pp)
    eng.upload_potential(np.moveaxis(V, 2, 0))
    eng.propagate_frame(0)
    got = eng.wavefunction()[:, 0]
    eng.close()
    t0 = time.time()
    worst = 0.0
    probes = orc.batched_probes(orc.probe_array(xs, ys, 30.0, eV), xs, ys, pp)
    for p0 in range(0, P, 8):                       # oracle in chunks of 8 probes (memory)
        want = orc.diffraction(orc.propagate(probes[p0:p0 + 8], V.astype(np.float64), xs, ys, zs, eV))
        g = got[p0:p0 + 8]
        worst = max(worst, max(np.linalg.norm(g[i] - want[i]) / np.linalg.norm(want[i]) for i in range(len(g))))
    print(f"{nx}x{ny} x {nz} slices x {P} probes: worst per-probe rel-L2 {worst:.2e}  (oracle {time.time() - t0:.0f} s)", flush=True)
    return worst

if __name__ == "__main__":
    bad = 0
    for cfg in [(256, 256, 3, 700), (256, 256, 4, 97), (512, 512, 4, 70), (512, 256, 3, 130), (1024, 1024, 3, 70), (1024, 1024, 2, 9),
                (2048, 2048, 5, 6), (2048, 512, 4, 20), (500, 360, 3, 40), (1024, 512, 4, 33), (256, 1024, 5, 65)]:
        bad += case(*cfg) >= 1e-4
    sys.exit(1 if bad else 0)
