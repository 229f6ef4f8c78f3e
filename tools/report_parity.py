"""Print the measured parity of the HIP path against the oracle / reference goldens (run on the GPU box):
    python tools/report_parity.py > gpurun_out/parity_report.txt"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pyslice_amd as ps  # noqa: E402
from oracle import multislice_oracle as orc  # noqa: E402
from pyslice_amd.synthetic import synthetic_trajectory  # noqa: E402


def npy(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def rel(a, b):
    return float(np.linalg.norm((np.asarray(a) - b).ravel()) / np.linalg.norm(np.asarray(b).ravel()))


def resid(a, b):
    F, D = np.abs(a), np.abs(b)
    return float(((F - D) ** 2).sum() / (F ** 2).sum())


print("case                                             wave rel-L2   ref residual   potential max|dV|/max|V|   TACAW rel-L2")
for name in ("g7_calculator_64", "g8_tacaw_32"):
    g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    pos = g["positions"]
    tr = ps.Trajectory(g["Z"], pos, np.zeros_like(pos), g["box"], 0.005)
    calc = ps.MultisliceCalculator(progress=False)
    calc.setup(tr, aperture=float(g["aperture"]), voltage_eV=float(g["eV"]), probe_positions=[tuple(p) for p in g["probe_positions"]])
    wf = calc.run()
    d = npy(wf.wavefunction_data)
    t = ""
    if "intensity" in g.files:
        t = f"{rel(npy(ps.TACAWData(wf).intensity), g['intensity']):.2e}"
    print(f"reference golden {name:30s}  {rel(d, g['wavefunction_data']):.2e}      {resid(d, g['wavefunction_data']):.2e}            -                    {t}")

for n, ny, nz, P, T, dens in ((256, 256, 50, 2, 1, 0.102), (256, 1024, 6, 3, 1, 0.03), (512, 512, 100, 1, 1, 0.102),
                               (1024, 1024, 24, 2, 1, 0.01), (101, 97, 8, 2, 3, 0.15)):
    tr = synthetic_trajectory(n, nz, T, ny=ny, density=dens, seed=5)
    xs, ys, zs, lx, ly, lz = ps.gridFromTrajectory(tr)
    pp = [(lx / 2, ly / 2)] + [tuple(v) for v in np.random.default_rng(1).random((P - 1, 2)) * [lx, ly]]
    calc = ps.MultisliceCalculator(progress=False)
    calc.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
    wf = calc.run()
    d = npy(wf.wavefunction_data)
    want = orc.run_frames(tr.box_matrix, tr.positions, tr.atom_types, 30.0, 100e3, pp)["wavefunction_data"]
    V = orc.potential(xs, ys, zs, tr.positions[0], tr.atom_types)
    Vg = npy(ps.Potential(xs, ys, zs, tr.positions[0], list(tr.atom_types)).array)
    tac = ""
    if T >= 2:
        f, inten = orc.tacaw(want, wf.time)
        tac = f"{rel(npy(ps.TACAWData(wf).intensity), inten):.2e}"
    reg = lambda m: m in (256, 512, 1024, 2048)
    path = "one-pass register kernels" if reg(n) and reg(ny) else "one-pass generic / Bluestein"
    print(f"oracle {n}x{ny}x{nz} P={P} T={T} ({path:19s})  {rel(d, want):.2e}      {resid(d, want):.2e}            {np.abs(V - Vg).max() / np.abs(V).max():.2e}             {tac}")
