"""CPU oracle for the multislice hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy complex128/float64 restatement of the reference algorithm for the path
Potential -> probes -> Propagate -> exit-wave FFT -> TACAW time-FFT.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only
as the checker / reported CPU baseline.  The product package (pyslice_amd/) never imports
it and has no CPU fallback.

Parity pin: every function below is checked against the imported reference
(/root/reference, importable in the build container) by tools/make_golden.py, which also
writes the fixtures in tests/golden/*.npz that tests/test_oracle.py replays on any machine.
Agreement with the reference: <= 1e-12 relative (see tests/golden/MANIFEST.json).

Each function cites the reference lines it restates.  The code is written from the closed
forms, not transcribed: e.g. the per-slice structure factor is one
matrix product per (element, slice), slices are found by explicit bin edges, probes are
built directly in reciprocal space.
"""
from __future__ import annotations

import os
import numpy as np

# Physical constants exactly as the reference states them (src/multislice/multislice.py:31-34).
M_ELECTRON = 9.109383e-31
Q_ELECTRON = 1.602177e-19
C_LIGHT = 299792458.0
H_PLANCK = 6.62607015e-34

_TABLE = None


def kirkland_table() -> np.ndarray:
    """(103,3,4) float64 table, last axis (a,b,c,d); reference: potentials.py:161-172."""
    global _TABLE
    if _TABLE is None:
        here = os.path.dirname(os.path.abspath(__file__))
        _TABLE = np.load(os.path.join(here, "..", "pyslice_amd", "data", "kirkland_abcd.npy"))
    return _TABLE


def wavelength(eV: float) -> float:
    """Relativistic electron wavelength in Angstrom; reference: multislice.py:41-42."""
    e = eV * Q_ELECTRON
    return H_PLANCK * C_LIGHT / np.sqrt(e * e + 2.0 * e * M_ELECTRON * C_LIGHT ** 2) * 1e10


def interaction_sigma(eV: float) -> float:
    """Interaction parameter sigma (Kirkland eq. 5.6); reference: multislice.py:258-260."""
    e0 = M_ELECTRON * C_LIGHT ** 2 / Q_ELECTRON
    return (2.0 * np.pi) / (wavelength(eV) * eV) * (e0 + eV) / (2.0 * e0 + eV)


def grid_from_box(box_matrix, sampling=0.1, slice_thickness=0.5):
    """xs, ys, zs, lx, ly, lz; reference: potentials.py:113-131 (n = int(L/step)+1, endpoint=False)."""
    lx, ly, lz = box_matrix[0, 0], box_matrix[1, 1], box_matrix[2, 2]
    nx = int(lx / sampling) + 1
    ny = int(ly / sampling) + 1
    nz = int(lz / slice_thickness) + 1
    xs = np.linspace(0, lx, nx, endpoint=False)
    ys = np.linspace(0, ly, ny, endpoint=False)
    zs = np.linspace(0, lz, nz, endpoint=False)
    return xs, ys, zs, lx, ly, lz


def form_factor(qsq: np.ndarray, Z: int) -> np.ndarray:
    """Kirkland f_Z(q^2) = sum_i a_i/(q^2+b_i) + sum_i c_i exp(-d_i q^2); reference: potentials.py:79-96."""
    p = kirkland_table()[Z - 1]
    out = np.zeros_like(qsq, dtype=np.float64)
    for a, b, c, d in p:
        out = out + a / (qsq + b)
    acc = np.zeros_like(qsq, dtype=np.float64)
    for a, b, c, d in p:
        acc = acc + c * np.exp(-d * qsq)
    return out + acc


def slice_edges(zs: np.ndarray):
    """Per-slice [lo, hi) bin edges along the beam axis; reference: potentials.py:302-307.

    slice 0 starts at 0, the last slice ends at zs[-1]+dz, inner edges sit at zs[s] -/+ dz/2.
    Atoms outside [0, zs[-1]+dz) belong to no slice (silently dropped by the reference).
    """
    nz = len(zs)
    dz = zs[1] - zs[0] if nz > 1 else 0.5
    lo = np.array([zs[s] - dz / 2 if s > 0 else 0.0 for s in range(nz)], dtype=np.float64)
    hi = np.array([zs[s] + dz / 2 if s < nz - 1 else zs[-1] + dz for s in range(nz)], dtype=np.float64)
    return lo, hi


def potential(xs, ys, zs, positions, atomic_numbers, slice_axis=2) -> np.ndarray:
    """Projected potential V[nx,ny,nz] (float64); reference: potentials.py:188-348.

    V_s = Re ifft2( sum_Z f_Z(q^2) * sum_{a in Z, slice s} exp(-2 pi i (kx x_a + ky y_a)) ) / (dx^2 dy^2).
    `atomic_numbers` is an int array (element names are mapped to Z by the caller).
    """
    xs = np.asarray(xs, np.float64); ys = np.asarray(ys, np.float64); zs = np.asarray(zs, np.float64)
    positions = np.asarray(positions, np.float64)
    atomic_numbers = np.asarray(atomic_numbers)
    nx, ny, nz = len(xs), len(ys), len(zs)
    dx = xs[1] - xs[0]
    dy = ys[1] - ys[0]
    axes = [0, 1, 2]
    axes.remove(slice_axis)
    ax1, ax2 = axes
    coords = [xs, ys, zs][slice_axis]            # slice coordinates (potentials.py:241-245)
    lo, hi = slice_edges(coords)
    n_slices = len(coords)
    kxs = np.fft.fftfreq(nx, d=dx)
    kys = np.fft.fftfreq(ny, d=dy)
    qsq = kxs[:, None] ** 2 + kys[None, :] ** 2
    recip = np.zeros((n_slices, nx, ny), dtype=np.complex128)
    for Z in np.unique(atomic_numbers):
        sel = positions[atomic_numbers == Z]
        if len(sel) == 0:
            continue
        fz = form_factor(qsq, int(Z))
        zc = sel[:, slice_axis]
        for s in range(n_slices):
            m = (zc >= lo[s]) & (zc < hi[s])
            if not m.any():
                continue
            px = sel[m, ax1]
            py = sel[m, ax2]
            ex = np.exp(-2j * np.pi * np.outer(kxs, px))     # (nx, a)
            ey = np.exp(-2j * np.pi * np.outer(py, kys))     # (a, ny)
            recip[s] += (ex @ ey) * fz
    v = np.fft.ifft2(recip, axes=(1, 2)).real / (dx ** 2 * dy ** 2)
    return np.ascontiguousarray(np.moveaxis(v, 0, 2))          # (nx, ny, nz) like the reference


def probe_array(xs, ys, mrad, eV) -> np.ndarray:
    """Base probe; reference: multislice.py:112-124.

    mrad == 0 -> real ones (plane wave, float64).  Otherwise ifftshift(ifft2(mask)) with the
    strict aperture mask |k| < mrad*1e-3/lambda.
    """
    nx, ny = len(xs), len(ys)
    if mrad == 0:
        return np.ones((nx, ny), dtype=np.float64)
    dx = xs[1] - xs[0]
    dy = ys[1] - ys[0]
    kxs = np.fft.fftfreq(nx, d=dx)
    kys = np.fft.fftfreq(ny, d=dy)
    kr = np.sqrt(kxs[:, None] ** 2 + kys[None, :] ** 2)
    mask = (kr < (mrad * 1e-3) / wavelength(eV)).astype(np.float64)
    return np.fft.ifftshift(np.fft.ifft2(mask))


def batched_probes(base: np.ndarray, xs, ys, positions_xy) -> np.ndarray:
    """(P,nx,ny) shifted probes; reference: multislice.py:216-233 (ramp exp(+2 pi i k p))."""
    nx, ny = len(xs), len(ys)
    kxs = np.fft.fftfreq(nx, d=xs[1] - xs[0])
    kys = np.fft.fftfreq(ny, d=ys[1] - ys[0])
    bk = np.fft.fft2(base)
    out = np.empty((len(positions_xy), nx, ny), dtype=np.complex128)
    for i, (px, py) in enumerate(positions_xy):
        ramp = np.exp(2j * np.pi * kxs * px)[:, None] * np.exp(2j * np.pi * kys * py)[None, :]
        out[i] = np.fft.ifft2(bk * ramp)
    return out


def defocus(array: np.ndarray, xs, ys, eV, dz) -> np.ndarray:
    """Probe.defocus; reference: multislice.py:183-190.

    P = exp(-i pi lambda dz k^2); dz > 0 multiplies the spectrum by P, dz < 0 DIVIDES by it -- and since P(dz<0) is
    exp(+i pi lambda |dz| k^2), the division applies exp(-i pi lambda |dz| k^2) again: both signs give the same
    result (a reference quirk the product reproduces); dz == 0 leaves the array alone.
    """
    if dz == 0:
        return np.asarray(array)
    kxs = np.fft.fftfreq(len(xs), d=xs[1] - xs[0])
    kys = np.fft.fftfreq(len(ys), d=ys[1] - ys[0])
    ph = np.exp(-1j * np.pi * wavelength(eV) * abs(dz) * (kxs[:, None] ** 2 + kys[None, :] ** 2))
    return np.fft.ifft2(ph * np.fft.fft2(array))


def cache_dir_name(n_frames, n_atoms, box_matrix, atom_types, aperture, voltage_eV, slice_thickness, sampling,
                   probe_positions, backend="pytorch") -> str:
    """Name of the reference's per-run cache directory below psi_data/; reference: calculators.py:78-94, 139."""
    import hashlib
    params = {'n_frames': n_frames, 'n_atoms': n_atoms, 'box_matrix': np.asarray(box_matrix).tolist(),
              'atom_types': np.asarray(atom_types).tolist(), 'aperture': aperture, 'voltage_eV': voltage_eV,
              'slice_thickness': slice_thickness, 'sampling': sampling, 'probe_positions': probe_positions,
              'backend': backend}
    return "torch_" + hashlib.md5(str(sorted(params.items())).encode()).hexdigest()[:12]


def _fft_pair(workers):
    """(fft2, ifft2) over the last two axes: numpy.fft (the reference's NumPy path) or, for the all-core CPU baseline
    of bench.py, scipy.fft with `workers` threads (same pocketfft kernels, batched over the probe axis)."""
    if not workers:
        return (lambda a: np.fft.fft2(a, axes=(-2, -1))), (lambda a: np.fft.ifft2(a, axes=(-2, -1)))
    import scipy.fft as sfft
    return (lambda a: sfft.fft2(a, axes=(-2, -1), workers=workers)), (lambda a: sfft.ifft2(a, axes=(-2, -1), workers=workers))


def propagate(probes: np.ndarray, V: np.ndarray, xs, ys, zs, eV, workers=None) -> np.ndarray:
    """Multislice loop; reference: multislice.py:254-299.

    probes (P,nx,ny) -> exit waves (P,nx,ny): nz transmissions, nz-1 Fresnel propagations.
    (The reference squeezes P==1; the oracle always keeps the probe axis.)
    """
    psi = np.array(probes, dtype=np.complex128)
    if psi.ndim == 2:
        psi = psi[None]
    nx, ny, nz = len(xs), len(ys), len(zs)
    lam = wavelength(eV)
    sig = interaction_sigma(eV)
    dz = zs[1] - zs[0] if nz > 1 else 0.5
    kxs = np.fft.fftfreq(nx, d=xs[1] - xs[0])
    kys = np.fft.fftfreq(ny, d=ys[1] - ys[0])
    prop = np.exp(-1j * np.pi * lam * dz * (kxs[:, None] ** 2 + kys[None, :] ** 2))
    fft2, ifft2 = _fft_pair(workers)
    for z in range(nz):
        psi = np.exp(1j * sig * V[:, :, z])[None] * psi
        if z < nz - 1:
            psi = ifft2(prop[None] * fft2(psi))
    return psi


def diffraction(exit_waves: np.ndarray, workers=None) -> np.ndarray:
    """fftshift(fft2(exit)) over the last two axes; reference: calculators.py:285-287."""
    return np.fft.fftshift(_fft_pair(workers)[0](exit_waves), axes=(-2, -1))


def usable_cores():
    """CPU cores this process may use (cgroup quota, affinity): the `workers` of the threaded FFT path"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def run_frames(box_matrix, positions_t, atomic_numbers, aperture, eV, probe_positions=None,
               sampling=0.1, slice_thickness=0.5, slice_axis=2, frames=None, workers=None):
    """Calculator-level oracle; reference: calculators.py:144-161, 172-186, 256-290.
    workers: threads of the FFTs (scipy.fft: the same pocketfft transforms as numpy.fft, tests/test_oracle.py checks both paths agree);
    the long GPU parity tests pass usable_cores().

    Returns dict(wavefunction_data (P,T,nx,ny,1) c128, xs, ys, zs, probe_positions).
    """
    xs, ys, zs, lx, ly, lz = grid_from_box(box_matrix, sampling, slice_thickness)
    if probe_positions is None:
        probe_positions = [(lx / 2, ly / 2)]
    base = probe_array(xs, ys, aperture, eV)
    frames = range(positions_t.shape[0]) if frames is None else frames
    frames = list(frames)
    out = np.zeros((len(probe_positions), len(frames), len(xs), len(ys), 1), dtype=np.complex128)
    pr = batched_probes(base, xs, ys, probe_positions)
    for ti, t in enumerate(frames):
        V = potential(xs, ys, zs, positions_t[t], atomic_numbers, slice_axis)
        ex = propagate(pr, V, xs, ys, zs, eV, workers=workers)
        out[:, ti, :, :, 0] = diffraction(ex, workers=workers)
    return dict(wavefunction_data=out, xs=xs, ys=ys, zs=zs, probe_positions=probe_positions)


def wf_axes(nx, ny, sampling, n_frames, timestep):
    """kxs, kys (float32 like torch's default fftfreq), time; reference: calculators.py:218-220."""
    kxs = np.fft.fftshift(np.fft.fftfreq(nx, sampling)).astype(np.float32)
    kys = np.fft.fftshift(np.fft.fftfreq(ny, sampling)).astype(np.float32)
    return kxs, kys, np.arange(n_frames) * timestep


def tacaw(wavefunction_data: np.ndarray, time: np.ndarray, layer_index=None):
    """frequencies, intensity (P,T,nx,ny); reference: tacaw_data.py:74-104."""
    n_layers = wavefunction_data.shape[4]
    if layer_index is None:
        layer_index = n_layers - 1
    if layer_index < 0 or layer_index >= n_layers:
        raise ValueError(f"layer_index {layer_index} out of range [0, {n_layers - 1}]")
    T = len(time)
    freqs = np.fft.fftshift(np.fft.fftfreq(T, d=time[1] - time[0]))
    wf = wavefunction_data[:, :, :, :, layer_index]
    wf = wf - wf.mean(axis=1, keepdims=True)
    spec = np.fft.fftshift(np.fft.fft(wf, axis=1), axes=1)
    return freqs, np.abs(spec) ** 2


# ---- reductions used by "next" rows -------------------------------------

def tacaw_spectrum(intensity, probe_index=None):
    """reference: tacaw_data.py:109-143."""
    s = intensity.sum(axis=(2, 3))
    return s.mean(axis=0) if probe_index is None else s[probe_index]


def tacaw_diffraction(intensity, probe_index=None):
    """reference: tacaw_data.py:183-217."""
    d = intensity.sum(axis=1)
    return d.mean(axis=0) if probe_index is None else d[probe_index]


def haadf(wavefunction_data, kxs, kys, probe_positions, eV, collection_angle=45.0):
    """ADF image; reference: haadf_data.py:44-68 (mask q>radius, mean_t sum_k |Psi*mask|)."""
    pp = np.asarray(probe_positions, dtype=np.float64)
    gx = np.asarray(sorted(set(pp[:, 0])))
    gy = np.asarray(sorted(set(pp[:, 1])))
    q = np.sqrt(np.asarray(kxs, np.float64)[:, None] ** 2 + np.asarray(kys, np.float64)[None, :] ** 2)
    mask = (q > (collection_angle * 1e-3) / wavelength(eV)).astype(np.float64)
    adf = np.zeros((len(gx), len(gy)))
    for i, x in enumerate(gx):
        for j, y in enumerate(gy):
            p = int(np.argmin(np.sqrt(((pp - np.array([x, y])[None]) ** 2).sum(axis=1))))
            ex = wavefunction_data[p, :, :, :, -1]
            adf[i, j] = np.mean(np.sum(np.abs(ex * mask[None]), axis=(1, 2)))
    return gx, gy, adf
