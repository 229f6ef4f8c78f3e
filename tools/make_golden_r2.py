"""Round-2 additions to tests/golden/ (same rules as tools/make_golden.py: the imported reference is run
here, in the build container, and only its inputs and outputs are stored).

    python tools/make_golden_r2.py

G10 Probe.defocus for dz > 0 and dz < 0 (reference multislice.py:183-190; 07_defocus.py uses dz = +1000).
G11 the reference's cache directory names and cache file format for two calculator runs
    (calculators.py:78-94, 139, 173, 311).
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
from src.multislice.multislice import Probe  # noqa: E402
from src.multislice.trajectory import Trajectory as RefTrajectory  # noqa: E402
from src.multislice.calculators import MultisliceCalculator  # noqa: E402
from oracle import multislice_oracle as orc  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
man_path = os.path.join(OUT, "MANIFEST.json")
manifest = json.load(open(man_path))


def npy(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))


def save(name, errs, **arrays):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    manifest["fixtures"][name] = {"oracle_vs_reference_rel_l2": errs, "bytes": os.path.getsize(os.path.join(OUT, name + ".npz")),
                                  "generator": "tools/make_golden_r2.py"}
    print(name, errs)


# ---------------- G10 defocus ----------------
arrs, errs = {}, {}
for tag, (nx, ny, lx, ly) in {"64": (64, 64, 6.35, 6.35), "96x80": (96, 80, 9.55, 7.95)}.items():
    xs = np.linspace(0, lx, nx, endpoint=False)
    ys = np.linspace(0, ly, ny, endpoint=False)
    arrs[f"xs_{tag}"], arrs[f"ys_{tag}"] = xs, ys
    for dz in (100.0, -100.0, 1000.0, -37.5):
        pr = Probe(xs, ys, 30, 100e3, device="cpu")
        base = npy(pr.array).copy()
        pr.defocus(dz)
        key = f"defocus_{tag}_{dz:g}"
        arrs[key] = npy(pr.array)
        errs[key] = rel(orc.defocus(base, xs, ys, 100e3, dz), arrs[key])
save("g10_defocus", errs, dz=np.array([100.0, -100.0, 1000.0, -37.5]), mrad=np.array(30.0), eV=np.array(100e3), **arrs)

# ---------------- G11 cache directory name + file format ----------------
def cache_case(box, n_atoms, T, pp, aperture, seed, eV=100e3):
    rng = np.random.default_rng(seed)
    L = np.array([box[0, 0], box[1, 1], box[2, 2]])
    pos0 = rng.random((n_atoms, 3)) * L
    types = np.asarray([(5, 7)[i % 2] for i in range(n_atoms)], dtype=np.int64)
    pos = np.stack([pos0 + 0.03 * np.sin(0.7 * t + pos0) for t in range(T)])
    tr = RefTrajectory(types, pos, np.zeros_like(pos), box, 0.005)
    if os.path.exists("psi_data"):
        shutil.rmtree("psi_data")
    calc = MultisliceCalculator(force_cpu=True)
    calc.setup(tr, aperture=aperture, voltage_eV=eV, sampling=0.1, slice_thickness=0.5, probe_positions=pp)
    wf = calc.run()
    name = calc.output_dir.name
    files = sorted(os.listdir(calc.output_dir))
    f0 = np.load(calc.output_dir / "frame_0.npy")
    oname = orc.cache_dir_name(T, n_atoms, box, types, aperture, eV, 0.5, 0.1, pp)
    assert oname == name, (oname, name)
    return dict(box=box, positions=pos, Z=types, aperture=np.array(aperture), eV=np.array(eV),
                probe_positions=np.asarray(pp if pp is not None else [], dtype=np.float64), has_positions=np.array(pp is not None),
                dir_name=np.array(name), files=np.array(files), frame0=f0, frame0_dtype=np.array(str(f0.dtype)),
                wavefunction_frame0=npy(wf.wavefunction_data)[:, 0])


a = cache_case(np.diag([3.15, 3.15, 1.75]), 6, 2, [(1.5, 1.5), (0.4, 2.2)], 30.0, 31)
b = cache_case(np.diag([3.15, 3.15, 1.25]), 5, 2, None, 0.0, 32)
save("g11_cache", {"dir_name_a": 0.0, "dir_name_b": 0.0}, **{k + "_a": v for k, v in a.items()}, **{k + "_b": v for k, v in b.items()})

json.dump(manifest, open(man_path, "w"), indent=1, default=float)
shutil.rmtree(scratch, ignore_errors=True)
print("done")
