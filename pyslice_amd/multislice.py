"""Probe / create_batched_probes / Propagate -- host mirror of src/multislice/multislice.py.

Same names, arguments and return conventions as the reference; the arrays are produced by the
HIP library.  Complex results come back as complex128 (torch tensors when torch is importable,
like the reference) after a float32 computation on the device.
"""
from __future__ import annotations

import numpy as np

from . import _native
from .potentials import TORCH_AVAILABLE, _as_tensor, _device_index

if TORCH_AVAILABLE:
    import torch

# constants exactly as the reference states them (multislice.py:31-34)
m_electron = 9.109383e-31
q_electron = 1.602177e-19
c_light = 299792458.0
h_planck = 6.62607015e-34


def m_effective(eV):
    """reference multislice.py:37-39"""
    return m_electron + eV * q_electron / c_light ** 2


def wavelength(eV):
    """Relativistic wavelength in Angstrom (reference multislice.py:41-42)."""
    return h_planck * c_light / ((eV * q_electron) ** 2 + 2 * eV * q_electron * m_electron * c_light ** 2) ** 0.5 * 1e10


def interaction_sigma(eV):
    """Kirkland eq. 5.6 as the reference evaluates it (multislice.py:258-260)."""
    E0_eV = m_electron * c_light ** 2 / q_electron
    return (2 * np.pi) / (wavelength(eV) * eV) * (E0_eV + eV) / (2 * E0_eV + eV)


def probe_grid(xlims, ylims, n, m):
    """reference multislice.py:193-195"""
    x, y = np.meshgrid(np.linspace(*xlims, n), np.linspace(*ylims, m))
    return np.reshape([x, y], (2, len(x.flat))).T


def _to_numpy(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


class Probe:
    """Aperture-limited probe (reference multislice.py:44-124).

    `array` is computed on the device on first access: ifftshift(ifft2(mask)) for mrad > 0,
    real ones for mrad == 0 (quirk Q5).  A probe made by create_batched_probes() remembers its
    recipe (mrad, positions) so Propagate() can rebuild it on the device without a host copy.
    """

    def __init__(self, xs, ys, mrad, eV, array=None, device=None):
        self.device = device
        self.use_torch = TORCH_AVAILABLE
        self.xs = xs
        self.ys = ys
        self.mrad = mrad
        self.eV = eV
        self.wavelength = wavelength(eV)
        xs_np, ys_np = np.asarray(xs, dtype=np.float64), np.asarray(ys, dtype=np.float64)
        self._nx, self._ny = len(xs_np), len(ys_np)
        self._dx, self._dy = xs_np[1] - xs_np[0], ys_np[1] - ys_np[0]
        self.kxs = _as_tensor(np.fft.fftfreq(self._nx, d=self._dx))
        self.kys = _as_tensor(np.fft.fftfreq(self._ny, d=self._dy))
        self._positions = None          # recipe: shifted copies of the analytic probe
        self._custom = array is not None
        self._array = None
        if array is not None:
            a = _to_numpy(array)
            self._array = _as_tensor(a.astype(np.complex128) if np.iscomplexobj(a) else a)

    # -- device recipe -----------------------------------------------------------------------
    def _engine_for(self, n_probes):
        return _native.Engine(self._nx, self._ny, 1, self._dx, self._dy, 0.5, self.wavelength, 0.0,
                              n_probes=n_probes, n_frames=0, device=_device_index(self.device))

    def _materialise(self):
        if self.mrad == 0 and self._positions is None:
            a = np.ones((self._nx, self._ny), dtype=np.float64)     # reference: real ones (multislice.py:112-113)
            return _as_tensor(a)
        pos = [(0.0, 0.0)] if self._positions is None else self._positions
        eng = self._engine_for(len(pos))
        try:
            eng.set_probes(self.mrad, pos)
            a = eng.probes().astype(np.complex128)
        finally:
            eng.close()
        return _as_tensor(a[0] if self._positions is None else a)

    @property
    def array(self):
        if self._array is None:
            self._array = self._materialise()
        return self._array

    @array.setter
    def array(self, value):
        self._array = value
        self._custom = True
        self._positions = None

    def to_cpu(self):
        return _to_numpy(self.array)

    def to_device(self, device):
        self.device = device
        return self

    def defocus(self, dz):
        """reference multislice.py:183-190: psi <- ifft2(P fft2(psi)) with P = exp(-i pi lambda dz k^2) for dz > 0 and
        psi <- ifft2(fft2(psi) / P) for dz < 0.  Dividing by P(dz<0) = exp(+i pi lambda |dz| k^2) applies
        exp(-i pi lambda |dz| k^2) again, so in the reference BOTH signs defocus by +|dz| (quirk Q19, pinned by
        tests/golden/g10_defocus.npz); the same here.  dz == 0 leaves the array alone.

        Evaluated on the device as one Fresnel step of the slice loop through vacuum (two empty slices).
        """
        if dz == 0:
            return
        base = _to_numpy(self.array).astype(np.complex64)
        if base.ndim != 2:
            raise ValueError("defocus() applies to a single (nx,ny) probe")
        eng = _native.Engine(self._nx, self._ny, 2, self._dx, self._dy, abs(float(dz)), self.wavelength, 0.0,
                             n_probes=1, n_frames=0, device=_device_index(self.device))
        try:
            eng.upload_potential(np.zeros((2, self._nx, self._ny), dtype=np.float32))
            eng.upload_probes(base)
            eng.propagate()
            self.array = _as_tensor(eng.exit_waves()[0].astype(np.complex128))
        finally:
            eng.close()

    def _recipe(self):
        """('analytic', positions) when the device can rebuild the probe, else ('array', ndarray (P,nx,ny))."""
        if not self._custom:
            return "analytic", ([(0.0, 0.0)] if self._positions is None else self._positions)
        a = _to_numpy(self._array)
        if a.ndim == 2:
            a = a[None]
            self._array = _as_tensor(a)             # Q13: the reference mutates probe.array to 3-D
        return "array", a


def create_batched_probes(base_probe, probe_positions, device=None):
    """(P,nx,ny) shifted probes (reference multislice.py:198-235; shift ramp exp(+2 pi i k p), quirk Q3)."""
    pos = [(float(px), float(py)) for px, py in probe_positions]
    out = Probe(base_probe.xs, base_probe.ys, base_probe.mrad, base_probe.eV, device=base_probe.device)
    if not base_probe._custom:
        out._positions = pos
        return out
    base = _to_numpy(base_probe.array)
    if base.ndim != 2:
        raise ValueError(f"base probe array must be 2-D, got {base.shape}")
    eng = out._engine_for(len(pos))
    try:
        eng.shift_probes(base.astype(np.complex64), pos)
        out._array = _as_tensor(eng.probes().astype(np.complex128))
        out._custom = True
    finally:
        eng.close()
    return out


def Propagate(probe, potential, device=None):
    """Multislice propagation (reference multislice.py:237-299).

    Returns the exit wave(s) as complex128, squeezed to (nx,ny) for a single probe, and -- like the
    reference -- leaves `probe.array` 3-D afterwards (quirk Q13).
    """
    eng = potential._engine
    kind, payload = probe._recipe()
    zs = _to_numpy(potential.zs)
    dz = zs[1] - zs[0] if len(zs) > 1 else 0.5
    eng.set_beam(probe.wavelength, interaction_sigma(probe.eV), dz)
    eng.resize_probes(len(payload))
    if kind == "analytic":
        eng.set_probes(probe.mrad, payload)
    else:
        if payload.shape[1:] != (eng.nx, eng.ny):
            raise ValueError(f"probe array {payload.shape} does not match the potential grid ({eng.nx},{eng.ny})")
        eng.upload_probes(payload.astype(np.complex64))
    eng.propagate()
    ex = eng.exit_waves().astype(np.complex128)
    if ex.shape[0] == 1:
        ex = ex[0]
    return _as_tensor(ex)
