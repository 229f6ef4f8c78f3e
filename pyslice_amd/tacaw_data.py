"""TACAWData -- host mirror of src/postprocessing/tacaw_data.py.

`TACAWData(wfdata, layer_index=None)` keeps the reference behaviour (re-classes itself and aliases
`wfdata.__dict__`, quirk Q16) but the time->frequency transform
    intensity[p,w,kx,ky] = | fftshift_t fft_t( Psi - <Psi>_t ) |^2          (tacaw_data.py:89-104)
runs in the HIP library (msl_tacaw).  When the WFData came from MultisliceCalculator.run() the
exit waves are still resident on the device and are transformed in place there; otherwise the
array is staged through torch device memory.  The reductions below (spectrum, diffraction, ...;
tacaw_data.py:109-353) stream the device copy of the intensity once through the library's
reduction kernels (msl_tacaw_spectrum / _diffraction / _dispersion) and return small host arrays.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import _native
from .potentials import TORCH_AVAILABLE, _as_tensor
from .wf_data import WFData

if TORCH_AVAILABLE:
    import torch


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


class TACAWData(WFData):
    def __init__(self, WFData, layer_index: int = None):
        # reference tacaw_data.py:39-42: alias the WFData's dict (in-place mutation of the source object)
        self.__dict__ = WFData.__dict__
        self.fft_from_wf_data(layer_index)

    def fft_from_wf_data(self, layer_index: int = None):
        if layer_index is None:
            layer_index = len(self.layer) - 1
        if layer_index < 0 or layer_index >= len(self.layer):
            raise ValueError(f"layer_index {layer_index} out of range [0, {len(self.layer)-1}]")
        n_freq = len(self.time)
        dt = self.time[1] - self.time[0]
        self.frequencies = np.fft.fftshift(np.fft.fftfreq(n_freq, d=dt))

        eng = self.__dict__.get("_engine")
        resident = eng is not None and self.__dict__.get("_resident", False) and len(self.layer) == 1
        shard = self.__dict__.get("_frame_shard")
        if shard is not None and eng is not None and len(self.layer) == 1:
            # multi-process run with frame-sharded exit waves still on the devices (gather="none"):
            # all-to-all to probe shards, local time FFT, gather of the intensities on rank 0
            self.intensity = self._tacaw_sharded(eng, shard)
            return
        if resident:
            eng.tacaw()
            self._intensity_src = (eng, None)          # reductions read the library's own buffer
            if self.__dict__.get("_output") == "device":
                self.intensity = torch.as_tensor(eng.result_view(_native.BUF_INTENSITY, "<f4"), device=f"cuda:{eng.device}")
            else:
                self.intensity = _as_tensor(eng.intensity().astype(np.float64))
            return
        # WFData assembled elsewhere (gathered shards, user arrays): stage through torch device memory
        if not TORCH_AVAILABLE or not torch.cuda.is_available():
            raise RuntimeError("TACAWData needs the HIP device (no CPU path in pyslice_amd)")
        wf = self.wavefunction_data
        wf = wf if hasattr(wf, "dim") else torch.from_numpy(np.ascontiguousarray(wf))
        layer = wf[:, :, :, :, layer_index]
        P, T, nx, ny = layer.shape
        dev = torch.device("cuda", torch.cuda.current_device())
        src = layer.to(device=dev, dtype=torch.complex64).contiguous()
        dst = torch.empty((P, T, nx, ny), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        helper = _native.Engine(2, 2, 1, 1.0, 1.0, 1.0, 1.0, 0.0, n_probes=1, n_frames=0, device=dev.index)
        try:
            helper.tacaw(src.data_ptr(), dst.data_ptr(), P, T, nx * ny)
        finally:
            helper.close()
        self._intensity_src = (None, dst)               # device copy kept for the reductions
        self.intensity = dst.to(torch.float64).cpu()

    def _tacaw_sharded(self, eng, shard):
        """Frame-sharded (P, T_r, nx, ny) on every rank -> intensity (P, T, nx, ny) on rank 0 (None elsewhere).

        One RCCL all-to-all re-shards frames -> probes, msl_tacaw runs on the rank's
        complete time series, one gather assembles the result."""
        from . import distributed as D
        n_frames, t_local = shard
        P, nx, ny = eng.n_probes, eng.wx, eng.wy
        dev = torch.device("cuda", eng.device)
        local = torch.as_tensor(eng.result_view(_native.BUF_WAVEFUNCTION, "<c8"), device=dev)[:, :t_local]
        mine = D.frames_to_probes(local.reshape(P, t_local, nx * ny), n_frames)         # (P_r, T, npix)
        out = torch.empty(mine.shape, dtype=torch.float32, device=dev)
        if mine.shape[0] > 0:
            torch.cuda.synchronize(dev)
            eng.tacaw(mine.data_ptr(), out.data_ptr(), mine.shape[0], n_frames, nx * ny)
        full = D.gather_probes(out, P, dst=0)
        if full is None:
            return None
        full = full.reshape(P, n_frames, nx, ny)
        self._intensity_src = (eng, full)
        return full if self.__dict__.get("_output") == "device" else full.to(torch.float64).cpu()

    # ---- reductions over intensity(P, F, kx, ky): device kernels behind the reference's method signatures -------
    def _source(self):
        """(engine, src, (B, F, K), ptr, ld) for the reduction entry points; src = None (library buffer) or (ptr, B, F, K);
        ld = elements between the rows of K pixels behind ptr (the library's buffer keeps its images at a line-aligned pitch)."""
        d = self.__dict__
        eng, dev = d.get("_intensity_src", (None, None))
        if dev is None and eng is not None:
            return (eng, None, (eng.n_probes, eng.intensity_F, eng.wx * eng.wy), eng.device_ptr(_native.BUF_INTENSITY),
                    eng.result_pitch(_native.BUF_INTENSITY))
        if dev is None:
            # intensity assembled elsewhere: stage a float32 copy on the device once
            if not TORCH_AVAILABLE or not torch.cuda.is_available():
                raise RuntimeError("TACAWData reductions need the HIP device (no CPU path in pyslice_amd)")
            I = self.intensity
            I = I if hasattr(I, "dim") else torch.from_numpy(np.ascontiguousarray(I))
            dev = I.to(device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float32).contiguous()
        if eng is None:
            eng = d.get("_reduce_engine")
            if eng is None:
                eng = _native.Engine(2, 2, 1, 1.0, 1.0, 1.0, 1.0, 0.0, n_probes=1, n_frames=0, device=dev.device.index)
                d["_reduce_engine"] = eng
        d["_intensity_src"] = (eng, dev)
        torch.cuda.synchronize(dev.device)
        B, F = int(dev.shape[0]), int(dev.shape[1])
        K = int(np.prod(dev.shape[2:]))
        return eng, (dev.data_ptr(), B, F, K), (B, F, K), dev.data_ptr(), K

    def _rows(self, b, f0, f1, mask=None):
        """sum over k of I[b, f0:f1] -> (f1-f0,) float64"""
        eng, _, (B, F, K), ptr, ld = self._source()
        return eng.tacaw_spectrum(mask, src=(ptr + 4 * (b * F + f0) * ld, 1, f1 - f0, K, ld))[0]

    def _check_probe(self, probe_index):
        if probe_index >= len(self.probe_positions):
            raise ValueError(f"Probe index {probe_index} out of range")

    def spectrum(self, probe_index: int = None) -> np.ndarray:
        """reference tacaw_data.py:109-143"""
        eng, src, (B, F, K), _, _ = self._source()
        if probe_index is None:
            return eng.tacaw_spectrum(src=src)[:len(self.probe_positions)].mean(axis=0)
        self._check_probe(probe_index)
        return self._rows(probe_index, 0, F)

    def spectrum_image(self, frequency: float, probe_indices: Optional[List[int]] = None) -> np.ndarray:
        """reference tacaw_data.py:145-179"""
        fi = int(np.argmin(np.abs(self.frequencies - frequency)))
        if probe_indices is None:
            probe_indices = list(range(len(self.probe_positions)))
        return np.array([self._rows(p, fi, fi + 1)[0] for p in probe_indices])

    def diffraction(self, probe_index: int = None) -> np.ndarray:
        """reference tacaw_data.py:183-217"""
        eng, src, (B, F, K), _, _ = self._source()
        shape = (len(self.kxs), len(self.kys))
        if probe_index is None:
            n = len(self.probe_positions)
            return eng.tacaw_diffraction(probes=(0, n), scale=1.0 / n, src=src).reshape(shape)
        self._check_probe(probe_index)
        return eng.tacaw_diffraction(probes=(probe_index, probe_index + 1), src=src).reshape(shape)

    def spectral_diffraction(self, frequency: float, probe_index: int = None) -> np.ndarray:
        """reference tacaw_data.py:219-254"""
        fi = int(np.argmin(np.abs(self.frequencies - frequency)))
        eng, src, (B, F, K), _, _ = self._source()
        shape = (len(self.kxs), len(self.kys))
        if probe_index is None:
            n = len(self.probe_positions)
            return eng.tacaw_diffraction(probes=(0, n), freqs=(fi, fi + 1), scale=1.0 / n, src=src).reshape(shape)
        self._check_probe(probe_index)
        return eng.tacaw_diffraction(probes=(probe_index, probe_index + 1), freqs=(fi, fi + 1), src=src).reshape(shape)

    def masked_spectrum(self, mask: np.ndarray, probe_index: int = None) -> np.ndarray:
        """reference tacaw_data.py:256-300 (its self.kx/self.ky lookup is broken, Q18; kxs/kys used here).
        As in the reference the mask multiplies the intensity, so a non-boolean mask weights it: a 0/1 mask takes the byte-mask
        kernel, any other mask the float64-weighted sum (msl_tacaw_spectrum_weighted)."""
        if mask.shape != (len(self.kxs), len(self.kys)):
            raise ValueError(f"Mask shape {mask.shape} doesn't match k-space shape ({len(self.kxs)}, {len(self.kys)})")
        eng, src, (B, F, K), ptr, ld = self._source()
        mask = np.asarray(mask)
        binary = mask.dtype == np.bool_ or bool(np.all((mask == 0) | (mask == 1)))

        def masked(b0, nb):
            s = (ptr + 4 * b0 * F * ld, nb, F, K, ld)
            return eng.tacaw_spectrum(mask != 0, src=s) if binary else eng.tacaw_spectrum_weighted(mask, src=s)

        if probe_index is None:
            return masked(0, len(self.probe_positions)).mean(axis=0)
        self._check_probe(probe_index)
        return masked(probe_index, 1)[0]

    def dispersion(self, kx_path: np.ndarray, ky_path: np.ndarray, probe_index: int = None) -> np.ndarray:
        """reference tacaw_data.py:302-353"""
        kxs, kys = _np(self.kxs), _np(self.kys)
        ix = np.array([int(np.argmin(np.abs(kxs - v))) for v in kx_path], dtype=np.int64)
        iy = np.array([int(np.argmin(np.abs(kys - v))) for v in ky_path], dtype=np.int64)
        n = min(len(ix), len(iy))
        eng, src, (B, F, K), _, _ = self._source()
        if probe_index is not None:
            self._check_probe(probe_index)
        out = np.zeros((len(self.frequencies), len(ix)))          # zip() semantics: columns past the shorter path stay 0
        if n == 0:
            return out
        g = eng.tacaw_dispersion(ix[:n] * len(kys) + iy[:n], src=src).astype(np.float64)      # (B, F, n)
        out[:, :n] = g[:len(self.probe_positions)].mean(axis=0) if probe_index is None else g[probe_index]
        return out
