"""Trajectory container -- the input boundary type of the hot path.

Mirrors the reference dataclass (src/multislice/trajectory.py:8-50): same field names, same
shape validation and the same ValueError messages, so scripts that build a Trajectory from
arrays work unchanged.  The tile/slice/displace helpers of the reference are pure NumPy
slicing off the timed path and are not restated here.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Trajectory:
    atom_types: np.ndarray      # (n_atoms,)
    positions: np.ndarray       # (n_frames, n_atoms, 3) Angstrom
    velocities: np.ndarray      # (n_frames, n_atoms, 3)
    box_matrix: np.ndarray      # (3, 3)
    timestep: float             # picoseconds

    def __post_init__(self):
        self._validate_shapes()

    def _validate_shapes(self):
        # reference: trajectory.py:20-40
        if self.positions.ndim != 3 or self.positions.shape[2] != 3:
            raise ValueError(f"positions must be (frames, atoms, 3), got {self.positions.shape}")
        if self.velocities.ndim != 3 or self.velocities.shape[2] != 3:
            raise ValueError(f"velocities must be (frames, atoms, 3), got {self.velocities.shape}")
        if self.atom_types.ndim != 1:
            raise ValueError(f"atom_types must be 1D, got {self.atom_types.ndim}D")
        if self.box_matrix.shape != (3, 3):
            raise ValueError(f"box_matrix must be (3, 3), got {self.box_matrix.shape}")
        nf_p, na_p = self.positions.shape[:2]
        nf_v, na_v = self.velocities.shape[:2]
        na_t = len(self.atom_types)
        if nf_p != nf_v:
            raise ValueError(f"Frame count mismatch: {nf_p} vs {nf_v}")
        if not (na_p == na_v == na_t):
            raise ValueError(f"Atom count mismatch: {na_p}, {na_v}, {na_t}")

    @property
    def n_frames(self) -> int:
        return self.positions.shape[0]

    @property
    def n_atoms(self) -> int:
        return len(self.atom_types)

    @property
    def box_tilts(self) -> np.ndarray:
        return np.array([self.box_matrix[0, 1], self.box_matrix[0, 2], self.box_matrix[1, 2]])
