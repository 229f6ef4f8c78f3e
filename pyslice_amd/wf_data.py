"""WFData -- the output boundary type (reference src/postprocessing/wf_data.py:9-28): same fields."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Tuple

import numpy as np


@dataclass
class WFData:
    """Wave-function data: probe_positions, time, kxs, kys, layer, wavefunction_data (P,T,kx,ky,layer), probe."""
    probe_positions: List[Tuple[float, float]]
    time: np.ndarray
    kxs: Any
    kys: Any
    layer: np.ndarray
    wavefunction_data: Any
    probe: Any
