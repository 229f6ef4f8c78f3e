"""HAADFData -- host mirror of src/postprocessing/haadf_data.py (a consumer of WFData).

ADF image: for every probe position, mean over frames of sum_k |Psi(k)| over the annulus
q > collection_angle*1e-3/lambda  (haadf_data.py:44-68).  The masked |.| reduction runs in the HIP
library (msl_adf): on the resident (P,T,nx,ny) buffer when the WFData came from run(), else on a
complex64 copy staged through torch device memory.
"""
from __future__ import annotations

import numpy as np

from . import _native
from .potentials import TORCH_AVAILABLE
from .wf_data import WFData

if TORCH_AVAILABLE:
    import torch


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


class HAADFData(WFData):
    def __init__(self, WFData):
        self.__dict__ = WFData.__dict__            # reference haadf_data.py:37-41 (aliases the source object)

    def calculateADF(self, collection_angle: float = 45, preview: bool = False) -> np.ndarray:
        pp = np.asarray(self.probe_positions, dtype=np.float64)
        self.xs = np.asarray(sorted(set(pp[:, 0])))
        self.ys = np.asarray(sorted(set(pp[:, 1])))
        kxs, kys = _np(self.kxs).astype(np.float64), _np(self.kys).astype(np.float64)
        q = np.sqrt(kxs[:, None] ** 2 + kys[None, :] ** 2)
        radius = (collection_angle * 1e-3) / self.probe.wavelength
        mask = (q > radius)
        eng = self.__dict__.get("_engine")
        if eng is not None and self.__dict__.get("_resident", False) and len(self.layer) == 1:
            per_probe = eng.adf(mask)                                  # resident exit waves of run()
        else:
            if not TORCH_AVAILABLE or not torch.cuda.is_available():
                raise RuntimeError("HAADFData needs the HIP device (no CPU path in pyslice_amd)")
            wf = self.wavefunction_data
            wf = wf if hasattr(wf, "dim") else torch.from_numpy(np.ascontiguousarray(wf))
            dev = wf.device if wf.is_cuda else torch.device("cuda", torch.cuda.current_device())
            src = wf[:, :, :, :, -1].to(device=dev, dtype=torch.complex64).contiguous()
            P, T, nx, ny = src.shape
            torch.cuda.synchronize(dev)
            helper = eng if (eng is not None and eng.device == dev.index) else \
                _native.Engine(2, 2, 1, 1.0, 1.0, 1.0, 1.0, 0.0, n_probes=1, n_frames=0, device=dev.index)
            try:
                per_probe = helper.adf(mask, src=(src.data_ptr(), P, T, nx * ny))
            finally:
                if helper is not eng:
                    helper.close()
        self.adf = np.zeros((len(self.xs), len(self.ys)))
        for i, x in enumerate(self.xs):
            for j, y in enumerate(self.ys):
                p = int(np.argmin(np.sqrt(((pp - np.array([x, y])[None, :]) ** 2).sum(axis=1))))
                self.adf[i, j] = per_probe[p]
        return self.adf
