"""HAADFData -- host mirror of src/postprocessing/haadf_data.py (a consumer of WFData).

ADF image: for every probe position, mean over frames of sum_k |Psi(k)| over the annulus
q > collection_angle*1e-3/lambda  (haadf_data.py:44-68).  The masked |.| reduction runs on the
device through torch (plumbing) when the wave data is device resident, else on the host array
that run() already returned.
"""
from __future__ import annotations

import numpy as np

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
        wf = self.wavefunction_data
        if TORCH_AVAILABLE and hasattr(wf, "dim"):
            m = torch.as_tensor(mask, device=wf.device)
            # (P,T,kx,ky): sum over the annulus, mean over frames
            per_probe = (wf[:, :, :, :, -1].abs() * m[None, None]).sum(dim=(2, 3)).mean(dim=1)
            per_probe = _np(per_probe).astype(np.float64)
        else:
            per_probe = (np.abs(wf[:, :, :, :, -1]) * mask[None, None]).sum(axis=(2, 3)).mean(axis=1)
        self.adf = np.zeros((len(self.xs), len(self.ys)))
        for i, x in enumerate(self.xs):
            for j, y in enumerate(self.ys):
                p = int(np.argmin(np.sqrt(((pp - np.array([x, y])[None, :]) ** 2).sum(axis=1))))
                self.adf[i, j] = per_probe[p]
        return self.adf
