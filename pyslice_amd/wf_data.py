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

    def release(self):
        """Not in the reference: drop this result's hold on the device.  A WFData / TACAWData that came from
        MultisliceCalculator.run() keeps the engine alive -- the (P,T,kx,ky) spectra stay resident so that TACAWData and the
        reductions work on the device copy without a host round trip -- and with it every device buffer of the run (hundreds
        of GB at BASELINE C3).  Once the host arrays are all that is needed, release() returns that memory; device-resident
        fields (output="device" views, the resident intensity) must not be used afterwards."""
        for key in ("_engine", "_intensity_src", "_reduce_engine", "_resident", "_frame_shard"):
            self.__dict__.pop(key, None)
