"""Grid helper and Potential -- host mirror of src/multislice/potentials.py of the reference.

`Potential(xs, ys, zs, positions, atomTypes, kind="kirkland", device=None, slice_axis=2)` keeps
the reference signature (potentials.py:188) but the projected potential is rasterised by the
HIP library (msl_build_potential); `.array` is fetched lazily in the reference's layout
(nx, ny, nz) / float64.  There is no CPU path.
"""
from __future__ import annotations

import os

import numpy as np

from . import _native

try:  # torch is plumbing only: the reference hands back torch tensors when torch is importable
    import torch
    TORCH_AVAILABLE = True
except ImportError:  # pragma: no cover
    torch = None
    TORCH_AVAILABLE = False

_ELEMENTS = ["H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P", "S", "Cl", "Ar",
             "K", "Ca", "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr",
             "Rb", "Sr", "Y", "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I", "Xe",
             "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb",
             "Lu", "Hf", "Ta", "W", "Re", "Os", "Ir", "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn",
             "Fr", "Ra", "Ac", "Th", "Pa", "U", "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es", "Fm", "Md", "No", "Lr"]

_TABLE = None


def getZfromElementName(element: str) -> int:
    """Atomic number from element symbol (reference potentials.py:98-111; its 'Ti'-for-Tl typo is not kept)."""
    return _ELEMENTS.index(element) + 1


def loadKirkland() -> np.ndarray:
    """(103,3,4) float64 Kirkland a,b,c,d table (reference potentials.py:134-185 reads kirkland.txt)."""
    global _TABLE
    if _TABLE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "kirkland_abcd.npy")
        if not os.path.exists(path):
            raise FileNotFoundError("Could not find kirkland_abcd.npy")
        _TABLE = np.load(path)
    return _TABLE


def gridFromTrajectory(trajectory, sampling=0.1, slice_thickness=0.5):
    """xs, ys, zs, lx, ly, lz from the box diagonal (reference potentials.py:113-131, quirk Q1)."""
    box = trajectory.box_matrix
    lx, ly, lz = box[0, 0], box[1, 1], box[2, 2]
    nx = int(lx / sampling) + 1
    ny = int(ly / sampling) + 1
    nz = int(lz / slice_thickness) + 1
    xs = np.linspace(0, lx, nx, endpoint=False)
    ys = np.linspace(0, ly, ny, endpoint=False)
    zs = np.linspace(0, lz, nz, endpoint=False)
    return xs, ys, zs, lx, ly, lz


def _line_cost(n: int) -> float:
    """modelled cost per point of a slice-loop pass along lines of n points = 1 / (fraction of the HBM peak measured for that
    kernel family): power-of-two register kernels 0.39-0.54, direct mixed-radix passes 0.27-0.33, any other
    length a zero-padded convolution on the next register transform M >= 2 n - 1, i.e. that transform's fraction times n / M"""
    from . import _native
    cls = _native.line_kernel_class(n)
    if cls == 2:
        return 1.0 / {256: 0.50, 512: 0.49, 1024: 0.54, 2048: 0.386}[n]
    if cls == 1:
        return 1.0 / (0.33 if n < 972 else 0.27)
    for m, base in ((256, 0.50), (1024, 0.56), (2048, 0.386), (4096, 0.27)):
        if 2 * n - 1 <= m or (m == 4096 and n <= 2047):
            return m / (base * n)
    return 20.0                                               # generic LDS kernel


def suggest_sampling(trajectory, sampling=0.1, max_refine=0.15, min_gain=1.15):
    """A finer `sampling` whose grid is modelled at least `min_gain` times cheaper in the slice loop, or None.

    The grid is int(L / sampling) + 1 points per axis (reference potentials.py:123-125), so the line lengths are whatever the box
    gives; a length without a kernel of its own (msl_line_kernel_class == 0) runs as a zero-padded convolution on the next
    power-of-two transform, which costs little just below a power of two (501 on 1024) and up to 4 x just above one (520 on 2048).
    Returns (sampling', nx', ny') with sampling (1 - max_refine) <= sampling' <= sampling and both axes on direct kernels: the
    candidate with the lowest modelled cost nx ny (cost(nx) + cost(ny)), if that beats the current grid by min_gain."""
    from . import _native
    box = trajectory.box_matrix
    lx, ly = float(box[0, 0]), float(box[1, 1])
    n_of = lambda L, s: int(L / s) + 1
    total = lambda a, b: a * b * (_line_cost(a) + _line_cost(b))
    now = total(n_of(lx, sampling), n_of(ly, sampling))
    best = None
    for L, other in ((lx, ly), (ly, lx)):
        n0 = n_of(L, sampling)
        for n in range(n0, int(n0 / (1.0 - max_refine)) + 2):
            if _native.line_kernel_class(n) == 0:
                continue
            s = L / (n - 0.5)                                   # the middle of the interval of samplings that give n points
            if not (sampling * (1.0 - max_refine) <= s <= sampling) or n_of(L, s) != n:
                continue
            if _native.line_kernel_class(n_of(other, s)) > 0:
                c = total(n_of(lx, s), n_of(ly, s))
                if best is None or c < best[0]:
                    best = (c, s, n_of(lx, s), n_of(ly, s))
    if best is None or best[0] * min_gain > now:
        return None
    return best[1:]


def slice_edges(coords: np.ndarray):
    """[lo, hi) per slice, the reference's masks (potentials.py:302-307), evaluated in float64."""
    n = len(coords)
    sp = coords[1] - coords[0] if n > 1 else 0.5
    lo = np.array([coords[s] - sp / 2 if s > 0 else 0.0 for s in range(n)], dtype=np.float64)
    hi = np.array([coords[s] + sp / 2 if s < n - 1 else coords[-1] + sp for s in range(n)], dtype=np.float64)
    return lo, hi


def atomic_numbers_of(atomTypes) -> np.ndarray:
    """Element names or ints -> int32 Z (reference potentials.py:261-266)."""
    out = np.empty(len(atomTypes), dtype=np.int32)
    for i, at in enumerate(atomTypes):
        out[i] = getZfromElementName(at) if isinstance(at, str) else int(at)
    return out


def _device_index(device) -> int:
    if device is None:
        return int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("MSL_USE_LOCAL_RANK") else 0
    if isinstance(device, int):
        return device
    s = str(device)
    if s in ("cuda", "hip"):
        return 0
    if s.startswith("cuda:") or s.startswith("hip:"):
        return int(s.split(":")[1])
    if s == "cpu":
        raise NotImplementedError("pyslice_amd has no CPU path; use the reference implementation for CPU runs")
    raise ValueError(f"unknown device {device!r}")


def _as_tensor(a):
    return torch.from_numpy(np.ascontiguousarray(a)) if TORCH_AVAILABLE else a


class Potential:
    def __init__(self, xs, ys, zs, positions, atomTypes, kind="kirkland", device=None, slice_axis=2):
        if kind != "kirkland":
            raise NotImplementedError("only kind='kirkland' is implemented (the reference's 'gauss' branch is dead code)")
        xs = np.asarray(xs, dtype=np.float64)
        ys = np.asarray(ys, dtype=np.float64)
        zs = np.asarray(zs, dtype=np.float64)
        nx, ny, nz = len(xs), len(ys), len(zs)
        dx = xs[1] - xs[0]
        dy = ys[1] - ys[0]
        dz = zs[1] - zs[0] if nz > 1 else 0.5
        self.device = device
        self.use_torch = TORCH_AVAILABLE
        self.xs, self.ys, self.zs = _as_tensor(xs), _as_tensor(ys), _as_tensor(zs)
        self.slice_axis = slice_axis
        axes = [0, 1, 2]
        axes.remove(slice_axis)
        self.inplane_axis1, self.inplane_axis2 = axes
        self.slice_coords = [xs, ys, zs][slice_axis]
        self.slice_spacing = [dx, dy, dz][slice_axis]
        self.n_slices = len(self.slice_coords)
        self.kxs = _as_tensor(np.fft.fftfreq(nx, d=dx))
        self.kys = _as_tensor(np.fft.fftfreq(ny, d=dy))
        self._nx, self._ny, self._dz = nx, ny, dz
        # The beam is unknown here (the reference's Potential takes no energy): sigma = 0 now,
        # Propagate() calls set_beam() and the library re-derives exp(i sigma V) from the kept V.
        self._engine = _native.Engine(nx, ny, self.n_slices, dx, dy, dz, wavelength=1.0, sigma=0.0, n_probes=1,
                                      n_frames=0, device=_device_index(device), keep_potential=True)
        self._engine.set_kirkland(loadKirkland())
        lo, hi = slice_edges(np.asarray(self.slice_coords, dtype=np.float64))
        self._engine.set_slices(lo, hi)
        self._engine.build_potential(np.asarray(positions, dtype=np.float64), atomic_numbers_of(atomTypes), slice_axis)
        self._array = None

    @property
    def array(self):
        """(nx, ny, n_slices) float64, z fastest -- the reference layout (potentials.py:334-348)."""
        if self._array is None:
            v = self._engine.potential()                       # (nz, nx, ny) float32 on the device
            a = np.ascontiguousarray(np.moveaxis(v, 0, 2)).astype(np.float64)
            self._array = _as_tensor(a)
        return self._array

    @array.setter
    def array(self, value):
        """Assigning a caller-made potential uploads it (Propagate() accepts any Potential, multislice.py:237)."""
        a = value.detach().cpu().numpy() if hasattr(value, "detach") else np.asarray(value)
        self._engine.upload_potential(np.ascontiguousarray(np.moveaxis(a, 2, 0), dtype=np.float32))
        self._array = _as_tensor(np.asarray(a, dtype=np.float64))

    def to_cpu(self):
        a = self.array
        return a.cpu().numpy() if hasattr(a, "cpu") else a

    def to_device(self, device):
        self.device = device
        return self
