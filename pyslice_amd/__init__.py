"""pyslice_amd -- MI355X-native multislice engine behind the PySlice calculator API.

Host code is Python; all arithmetic of the hot path (projected Kirkland potential, probes, the
FFT / transmission / Fresnel slice loop, exit-wave FFT, TACAW time FFT) runs in the C-ABI HIP
library `libmslice.so` (include/mslice.h, pyslice_amd/csrc).  There is no CPU fallback.
"""
from .trajectory import Trajectory
from .wf_data import WFData
from .potentials import Potential, gridFromTrajectory, getZfromElementName, loadKirkland
from .multislice import Probe, Propagate, create_batched_probes, probe_grid, wavelength, m_effective
from .calculators import MultisliceCalculator
from .tacaw_data import TACAWData
from .haadf_data import HAADFData

__all__ = ["Trajectory", "WFData", "Potential", "gridFromTrajectory", "getZfromElementName", "loadKirkland",
           "Probe", "Propagate", "create_batched_probes", "probe_grid", "wavelength", "m_effective",
           "MultisliceCalculator", "TACAWData", "HAADFData"]
__version__ = "0.1.0"
