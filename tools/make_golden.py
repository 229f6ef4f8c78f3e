"""Generate tests/golden/*.npz by running the imported reference (build container only).

The reference (/root/reference, pure Python) is importable here; it cannot travel to the GPU
box, so its outputs on seeded inputs are committed as data fixtures.  For every fixture the
oracle (oracle/multislice_oracle.py) is evaluated on the same inputs and the relative error is
recorded in tests/golden/MANIFEST.json -- that is the pin of the oracle.

    cd /root/repo && python tools/make_golden.py

Fixtures: G1 grid, G2 wavelength/sigma, G3 form factors, G4 potentials
(incl. edge atoms), G5 probes, G6 propagate, G7 calculator end-to-end, G8 TACAW, G9 HAADF.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

scratch = tempfile.mkdtemp(prefix="refrun_")
os.chdir(scratch)

import torch  # noqa: E402

from src.multislice.potentials import gridFromTrajectory, Potential, kirkland  # noqa: E402
from src.multislice.multislice import (Probe, Propagate, create_batched_probes, wavelength,  # noqa: E402
                                       m_electron, c_light, q_electron)
from src.multislice.trajectory import Trajectory as RefTrajectory  # noqa: E402
from src.multislice.calculators import MultisliceCalculator  # noqa: E402
from src.postprocessing.tacaw_data import TACAWData  # noqa: E402
from src.postprocessing.haadf_data import HAADFData  # noqa: E402

from oracle import multislice_oracle as orc  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
manifest = {}


def npy(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return float(d / n) if n > 0 else float(d)


def save(name, errs, **arrays):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    manifest[name] = {"oracle_vs_reference_rel_l2": errs,
                      "bytes": os.path.getsize(os.path.join(OUT, name + ".npz"))}
    print(name, errs)


def make_traj(box, n_atoms, n_frames, seed, species=(5, 7), extra=None, amp=0.03):
    rng = np.random.default_rng(seed)
    L = np.array([box[0, 0], box[1, 1], box[2, 2]])
    pos0 = rng.random((n_atoms, 3)) * L
    if extra is not None:
        pos0 = np.concatenate([pos0, extra], axis=0)
    types = np.asarray([species[i % len(species)] for i in range(len(pos0))], dtype=np.int64)
    ph = np.random.default_rng(seed + 1).random((len(pos0), 3)) * 2 * np.pi
    pos = np.stack([pos0 + amp * np.sin(2 * np.pi * 25.0 * t * 0.005 + ph) for t in range(n_frames)])
    return types, pos


# ---------------- G1 grid ----------------
boxes = [np.diag([6.35, 6.35, 2.75]), np.diag([6.45, 6.4, 3.0]), np.diag([12.75, 12.75, 4.75]),
         np.diag([25.55, 25.55, 24.75]), np.diag([9.55, 7.95, 2.25]), np.diag([10.0, 5.0, 1.0])]
g1 = []
for b in boxes:
    tr = RefTrajectory(np.array([5, 7]), np.zeros((1, 2, 3)), np.zeros((1, 2, 3)), b, 0.005)
    xs, ys, zs, lx, ly, lz = gridFromTrajectory(tr, sampling=0.1, slice_thickness=0.5)
    oxs, oys, ozs, *_ = orc.grid_from_box(b, 0.1, 0.5)
    assert len(xs) == len(oxs) and len(ys) == len(oys) and len(zs) == len(ozs)
    g1.append([len(xs), len(ys), len(zs), xs[1] - xs[0], ys[1] - ys[0], zs[1] - zs[0] if len(zs) > 1 else 0.5])
    assert np.array_equal(xs, oxs) and np.array_equal(zs, ozs)
save("g1_grid", 0.0, boxes=np.asarray(boxes), expect=np.asarray(g1))

# ---------------- G2 wavelength / sigma ----------------
evs = np.array([60e3, 100e3, 200e3, 300e3])
lam = np.array([wavelength(e) for e in evs])
E0 = m_electron * c_light ** 2 / q_electron
sig = np.array([(2 * np.pi) / (wavelength(e) * e) * (E0 + e) / (2 * E0 + e) for e in evs])
save("g2_wavelength", {"lambda": rel([orc.wavelength(e) for e in evs], lam),
                       "sigma": rel([orc.interaction_sigma(e) for e in evs], sig)},
     eV=evs, wavelength=lam, sigma=sig)

# ---------------- G3 form factors ----------------
k = np.fft.fftfreq(32, d=0.1)
qsq = k[:, None] ** 2 + k[None, :] ** 2
Zs = np.array([1, 5, 6, 7, 14, 31, 79])
ff = np.stack([npy(kirkland(torch.tensor(qsq, dtype=torch.float64), int(Z))) for Z in Zs])
off = np.stack([orc.form_factor(qsq, int(Z)) for Z in Zs])
save("g3_formfactor", rel(off, ff), qsq=qsq, Z=Zs, f=ff)

# ---------------- G4 potentials ----------------
def pot_case(name, box, n_atoms, seed, slice_axis=2):
    tr0 = RefTrajectory(np.array([5, 7]), np.zeros((1, 2, 3)), np.zeros((1, 2, 3)), box, 0.005)
    xs, ys, zs, lx, ly, lz = gridFromTrajectory(tr0, 0.1, 0.5)
    dz = zs[1] - zs[0]
    extra = np.array([
        [-0.3, 1.0, 0.2],                 # x outside box (wraps implicitly)
        [lx + 0.4, ly + 0.2, 0.9],        # x,y beyond box
        [1.0, 2.0, -0.1],                 # z < 0 (dropped)
        [2.0, 1.0, lz + 0.3],             # z >= zs[-1]+dz?  (dropped or last slice)
        [3.0, 3.0, zs[-1] + 0.9 * dz],    # inside extended last slice
        [1.5, 2.5, zs[1] - dz / 2],       # exactly on a slice edge
        [2.5, 0.5, 0.0],                  # z == 0
        [0.0, 0.0, zs[-1] - dz / 2],      # exactly on last slice lower edge
    ])
    types, pos = make_traj(box, n_atoms, 1, seed, extra=extra)
    P = Potential(xs, ys, zs, pos[0], list(types), kind="kirkland", device="cpu", slice_axis=slice_axis)
    V = npy(P.array)
    oV = orc.potential(xs, ys, zs, pos[0], types, slice_axis)
    save(name, rel(oV, V), box=box, positions=pos[0], Z=types, V=V, slice_axis=np.array(slice_axis))
    return xs, ys, zs, pos[0], types, V


c64 = pot_case("g4_potential_64", np.diag([6.35, 6.35, 2.75]), 12, 11)
c96 = pot_case("g4_potential_96x80", np.diag([9.55, 7.95, 2.25]), 16, 12)
# slice axis 0 (quirk Q17: the two in-plane axes are painted onto the kx(xs)/ky(ys) grids)
cax = pot_case("g4_potential_axis0", np.diag([3.15, 3.15, 2.75]), 10, 13, slice_axis=0)
# element names instead of ints take the string branch (potentials.py:287-291): same numbers
xs, ys, zs, pos, types, V = c64
names = ["B" if z == 5 else "N" for z in types]
Vn = npy(Potential(xs, ys, zs, pos, names, kind="kirkland", device="cpu").array)
assert rel(Vn, V) < 1e-14

# ---------------- G5 probes ----------------
xs, ys, zs = c64[0], c64[1], c64[2]
pp = [(3.175, 3.175), (1.0, 2.0), (0.0, 0.0), (5.9, 0.35)]
arrs = {}
errs = {}
for mrad in (0, 5, 30):
    pr = Probe(xs, ys, mrad, 100e3, device="cpu")
    arrs[f"base_{mrad}"] = npy(pr.array)
    errs[f"base_{mrad}"] = rel(orc.probe_array(xs, ys, mrad, 100e3), arrs[f"base_{mrad}"])
    bp = create_batched_probes(pr, pp)
    arrs[f"batch_{mrad}"] = npy(bp.array)
    errs[f"batch_{mrad}"] = rel(orc.batched_probes(orc.probe_array(xs, ys, mrad, 100e3), xs, ys, pp), arrs[f"batch_{mrad}"])
save("g5_probes", errs, xs=xs, ys=ys, positions=np.asarray(pp), eV=np.array(100e3), **arrs)

# non-square probe
xs2, ys2 = c96[0], c96[1]
pr = Probe(xs2, ys2, 30, 100e3, device="cpu")
pp2 = [(4.0, 4.0), (0.7, 6.1)]
bp = create_batched_probes(pr, pp2)
save("g5_probes_96x80", rel(orc.batched_probes(orc.probe_array(xs2, ys2, 30, 100e3), xs2, ys2, pp2), npy(bp.array)),
     xs=xs2, ys=ys2, positions=np.asarray(pp2), eV=np.array(100e3), batch_30=npy(bp.array))

# ---------------- G6 propagate ----------------
def prop_case(name, case, mrad, pp, eV=100e3):
    xs, ys, zs, pos, types, V = case
    P = Potential(xs, ys, zs, pos, list(types), kind="kirkland", device="cpu")
    pr = Probe(xs, ys, mrad, eV, device="cpu")
    bp = create_batched_probes(pr, pp)
    ex = npy(Propagate(bp, P, device=torch.device("cpu")))
    if ex.ndim == 2:
        ex = ex[None]
    oex = orc.propagate(orc.batched_probes(orc.probe_array(xs, ys, mrad, eV), xs, ys, pp), V, xs, ys, zs, eV)
    save(name, rel(oex, ex), xs=xs, ys=ys, zs=zs, V=V, positions=np.asarray(pp), mrad=np.array(mrad),
         eV=np.array(eV), exit=ex)


prop_case("g6_propagate_64_single", c64, 5, [(3.175, 3.175)])
prop_case("g6_propagate_64_batch", c64, 30, pp)
prop_case("g6_propagate_64_plane", c64, 0, [(3.175, 3.175)])
prop_case("g6_propagate_96x80_batch", c96, 30, pp2)

# ---------------- G7 calculator end-to-end ----------------
def calc_case(name, box, n_atoms, T, pp, aperture, seed, eV=100e3):
    types, pos = make_traj(box, n_atoms, T, seed)
    tr = RefTrajectory(types, pos, np.zeros_like(pos), box, 0.005)
    if os.path.exists("psi_data"):
        shutil.rmtree("psi_data")
    calc = MultisliceCalculator(force_cpu=True)
    calc.setup(tr, aperture=aperture, voltage_eV=eV, sampling=0.1, slice_thickness=0.5, probe_positions=pp)
    wf = calc.run()
    data = npy(wf.wavefunction_data)
    o = orc.run_frames(box, pos, types, aperture, eV, pp)
    okx, oky, ot = orc.wf_axes(len(o["xs"]), len(o["ys"]), 0.1, T, 0.005)
    errs = {"wavefunction_data": rel(o["wavefunction_data"], data), "kxs": rel(okx, npy(wf.kxs)),
            "time": rel(ot, wf.time)}
    assert npy(wf.kxs).dtype == np.float32
    return tr, wf, data, errs, dict(box=box, positions=pos, Z=types, aperture=np.array(aperture), eV=np.array(eV),
                                    probe_positions=np.asarray(wf.probe_positions, dtype=np.float64),
                                    wavefunction_data=data, kxs=npy(wf.kxs), kys=npy(wf.kys), time=wf.time)


tr, wf, data, errs, arrays = calc_case("g7", np.diag([6.35, 6.35, 2.75]), 14, 4, [(3.0, 3.0), (1.2, 4.4)], 30.0, 21)
save("g7_calculator_64", errs, **arrays)
tr, wf, data, errs, arrays = calc_case("g7d", np.diag([3.15, 3.15, 1.75]), 6, 3, None, 0.0, 22)
save("g7_calculator_32_default_probe", errs, **arrays)

# ---------------- G8 TACAW ----------------
tr, wf, data, errs, arrays = calc_case("g8", np.diag([3.15, 3.15, 1.75]), 8, 8, [(1.5, 1.5), (0.4, 2.2)], 30.0, 23)
tac = TACAWData(wf)
inten = npy(tac.intensity)
of, oi = orc.tacaw(data, wf.time)
e8 = {"frequencies": rel(of, tac.frequencies), "intensity": rel(oi, inten),
      "spectrum0": rel(orc.tacaw_spectrum(oi, 0), tac.spectrum(0)),
      "spectrum_all": rel(orc.tacaw_spectrum(oi, None), tac.spectrum(None)),
      "diffraction0": rel(orc.tacaw_diffraction(oi, 0), tac.diffraction(0)),
      "diffraction_all": rel(orc.tacaw_diffraction(oi, None), tac.diffraction(None))}
e8.update(errs)
arrays.update(frequencies=tac.frequencies, intensity=inten, spectrum0=tac.spectrum(0), spectrum_all=tac.spectrum(None),
              diffraction0=tac.diffraction(0), diffraction_all=tac.diffraction(None),
              spectrum_image_25=tac.spectrum_image(25.0), spectral_diffraction_25=tac.spectral_diffraction(25.0, 1))
save("g8_tacaw_32", e8, **arrays)

# ---------------- G9 HAADF (next-row consumer) ----------------
gx, gy = np.meshgrid(np.linspace(1.0, 2.0, 2), np.linspace(1.0, 2.5, 3))
pp9 = np.reshape([gx, gy], (2, gx.size)).T
tr, wf, data, errs, arrays = calc_case("g9", np.diag([3.15, 3.15, 1.75]), 8, 2, pp9, 30.0, 24)
wf.probe_positions = np.asarray(wf.probe_positions)
wf.wavefunction_data = npy(wf.wavefunction_data) if not hasattr(wf.wavefunction_data, "dim") else wf.wavefunction_data
h = HAADFData(wf)
try:
    adf = npy(h.calculateADF(collection_angle=45, preview=False))
    ogx, ogy, oadf = orc.haadf(data, arrays["kxs"], arrays["kys"], pp9, 100e3, 45.0)
    errs["adf"] = rel(oadf, adf)
    arrays.update(adf=adf, collection_angle=np.array(45.0))
    save("g9_haadf_32", errs, **arrays)
except Exception as exc:  # the reference's HAADF path is fragile across backends; record, don't fail
    print("HAADF reference failed:", repr(exc))
    manifest["g9_haadf_32"] = {"skipped": repr(exc)}

with open(os.path.join(OUT, "MANIFEST.json"), "w") as fh:
    json.dump({"generator": "tools/make_golden.py", "reference": "h-walk/PySlice @ 2025-09-19 (torch-CPU path, complex128)",
               "numpy": np.__version__, "torch": torch.__version__, "fixtures": manifest}, fh, indent=1, default=float)
shutil.rmtree(scratch, ignore_errors=True)
print("done")
