"""Seeded synthetic MD trajectories for the benchmark / parity workloads.

Box diag((N-1/2)*sampling, (N-1/2)*sampling, (nz-1/2)*slice_thickness) gives exactly an
N x N x nz grid through the reference's `n = int(L/step)+1` rule (potentials.py:123-125).
Atoms: uniform random at hBN number density (0.102 atoms/A^3), Z alternating 5/7.  Frames:
every atom oscillates as sum_m A sin(2 pi f_m t dt + phi_{a,m}) with f in {10,25,40} THz so the
TACAW spectrum has known peaks.
"""
from __future__ import annotations

import numpy as np

from .trajectory import Trajectory

HBN_DENSITY = 0.102          # atoms / A^3
PHONON_THZ = (10.0, 25.0, 40.0)


def box_for_grid(n: int, nz: int, sampling: float = 0.1, slice_thickness: float = 0.5,
                 ny: int | None = None) -> np.ndarray:
    ny = n if ny is None else ny
    return np.diag([(n - 0.5) * sampling, (ny - 0.5) * sampling, (nz - 0.5) * slice_thickness])


def synthetic_trajectory(n: int, nz: int, n_frames: int, *, ny: int | None = None,
                         sampling: float = 0.1, slice_thickness: float = 0.5,
                         density: float = HBN_DENSITY, amplitude: float = 0.03,
                         timestep: float = 0.005, seed: int = 0,
                         species=(5, 7)) -> Trajectory:
    box = box_for_grid(n, nz, sampling, slice_thickness, ny)
    lx, ly, lz = box[0, 0], box[1, 1], box[2, 2]
    n_atoms = max(2, int(round(density * lx * ly * lz)))
    rng = np.random.default_rng(seed)
    pos0 = rng.random((n_atoms, 3)) * np.array([lx, ly, lz])
    types = np.asarray([species[i % len(species)] for i in range(n_atoms)], dtype=np.int64)
    prng = np.random.default_rng(seed + 1)
    phases = prng.random((len(PHONON_THZ), n_atoms, 3)) * 2 * np.pi
    positions = np.empty((n_frames, n_atoms, 3), dtype=np.float64)
    for t in range(n_frames):
        disp = np.zeros((n_atoms, 3))
        for m, f in enumerate(PHONON_THZ):
            disp += amplitude * np.sin(2 * np.pi * f * t * timestep + phases[m])
        p = pos0 + disp
        p[:, 2] = np.clip(p[:, 2], 0.0, np.nextafter(lz, 0.0))
        positions[t] = p
    return Trajectory(atom_types=types, positions=positions,
                      velocities=np.zeros_like(positions), box_matrix=box, timestep=timestep)


def stem_probe_grid(n_side: int = 8, a: float = 2.4908, b: float = 2.1571) -> np.ndarray:
    """n_side x n_side probe raster over [a,3a]x[b,3b] (reference 03_manyprobes.py:16,24-25)."""
    x, y = np.meshgrid(np.linspace(a, 3 * a, n_side), np.linspace(b, 3 * b, n_side))
    return np.reshape([x, y], (2, x.size)).T
