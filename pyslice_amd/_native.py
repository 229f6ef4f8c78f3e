"""ctypes binding of libmslice.so (include/mslice.h) -- the only door to the device.

There is deliberately no CPU fallback: if the shared library is missing or no HIP device is
present every call raises.  The NumPy oracle lives under oracle/ and is never imported here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSL_LIB") or os.path.join(_HERE, "libmslice.so")

ABI_VERSION = 3          # include/mslice.h: MSL_ABI_VERSION
MSL_OK, MSL_ERR_INVALID, MSL_ERR_HIP, MSL_ERR_UNSUPPORTED, MSL_ERR_STATE, MSL_ERR_NOMEM = 0, -1, -2, -3, -4, -5
(BUF_PROBES, BUF_EXIT, BUF_POTENTIAL, BUF_TRANSMISSION, BUF_WAVEFUNCTION, BUF_INTENSITY, BUF_FORMFACTOR,
 BUF_STREAM_ACC, BUF_STREAM_S1, BUF_STREAM_S2, BUF_STREAM_REF) = range(11)

EXPORTS = [
    "msl_abi_version", "msl_line_kernel_class", "msl_last_error", "msl_create", "msl_destroy", "msl_set_kirkland", "msl_set_slices",
    "msl_set_beam", "msl_resize_probes", "msl_set_probes", "msl_upload_probes", "msl_shift_probes",
    "msl_build_potential", "msl_upload_potential", "msl_propagate", "msl_propagate_frame", "msl_tacaw",
    "msl_download", "msl_download_wavefunction_c128", "msl_download_frame", "msl_upload_frame", "msl_buffer_bytes", "msl_result_pitch", "msl_device_ptr", "msl_synchronize",
    "msl_get_counters",
    "msl_reset_counters", "msl_fft2_host",
    "msl_tacaw_spectrum", "msl_tacaw_spectrum_weighted", "msl_tacaw_diffraction", "msl_tacaw_dispersion", "msl_adf",
    "msl_select_batch_slot", "msl_propagate_frames", "msl_frame_batch", "msl_build_potentials",
    "msl_tacaw_stream_begin", "msl_tacaw_stream_push", "msl_tacaw_stream_finish",
    "msl_tacaw_stream_set_reference", "msl_tacaw_stream_finish_range",
]


class MslConfig(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
                ("wavelength", C.c_double), ("sigma", C.c_double),
                ("n_probes", C.c_int32), ("n_frames", C.c_int32), ("device", C.c_int32),
                ("keep_potential", C.c_int32), ("fft_path", C.c_int32),
                ("window_nx", C.c_int32), ("window_ny", C.c_int32), ("launch_timing", C.c_int32),
                ("frame_batch", C.c_int32), ("bin_nx", C.c_int32), ("bin_ny", C.c_int32), ("reserved", C.c_int32 * 1)]


class MslCounters(C.Structure):
    _fields_ = [("slice_steps", C.c_uint64), ("frames", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("ms_potential", C.c_double), ("ms_propagate", C.c_double), ("ms_tacaw", C.c_double),
                ("slice_kernel_launches", C.c_uint64), ("ms_slice_kernels", C.c_double),
                ("row_launches", C.c_uint64), ("ms_row", C.c_double),
                ("col_launches", C.c_uint64), ("ms_col", C.c_double)]


_lib = None


def load():
    """Load libmslice.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `python -m pyslice_amd.build_native`). "
            "pyslice_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    sig = {
        "msl_abi_version": (C.c_int, []),
        "msl_line_kernel_class": (C.c_int, [i32]),
        "msl_last_error": (C.c_char_p, [vp]),
        "msl_create": (C.c_int, [C.POINTER(MslConfig), C.POINTER(vp)]),
        "msl_destroy": (C.c_int, [vp]),
        "msl_set_kirkland": (C.c_int, [vp, vp]),
        "msl_set_slices": (C.c_int, [vp, vp, vp]),
        "msl_set_beam": (C.c_int, [vp, dbl, dbl, dbl]),
        "msl_resize_probes": (C.c_int, [vp, i32]),
        "msl_set_probes": (C.c_int, [vp, dbl, vp, i32]),
        "msl_upload_probes": (C.c_int, [vp, vp, i32]),
        "msl_shift_probes": (C.c_int, [vp, vp, vp, i32]),
        "msl_build_potential": (C.c_int, [vp, vp, vp, i64, i32, i32, i32]),
        "msl_build_potentials": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, i32]),
        "msl_upload_potential": (C.c_int, [vp, vp]),
        "msl_propagate": (C.c_int, [vp]),
        "msl_propagate_frame": (C.c_int, [vp, i32]),
        "msl_tacaw": (C.c_int, [vp, vp, vp, i64, i32, i64]),
        "msl_download": (C.c_int, [vp, C.c_int, vp, C.c_size_t, i64, i64]),
        "msl_download_frame": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "msl_download_wavefunction_c128": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "msl_upload_frame": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "msl_buffer_bytes": (C.c_size_t, [vp, C.c_int]),
        "msl_result_pitch": (i64, [vp, C.c_int]),
        "msl_device_ptr": (vp, [vp, C.c_int]),
        "msl_synchronize": (C.c_int, [vp]),
        "msl_get_counters": (C.c_int, [vp, C.POINTER(MslCounters)]),
        "msl_reset_counters": (C.c_int, [vp]),
        "msl_fft2_host": (C.c_int, [vp, vp, vp, i32, i32]),
        "msl_tacaw_spectrum": (C.c_int, [vp, vp, i64, i64, i64, i64, vp, vp]),
        "msl_tacaw_spectrum_weighted": (C.c_int, [vp, vp, i64, i64, i64, i64, vp, vp]),
        "msl_tacaw_diffraction": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, dbl, vp]),
        "msl_tacaw_dispersion": (C.c_int, [vp, vp, i64, i64, i64, i64, vp, i64, vp]),
        "msl_adf": (C.c_int, [vp, vp, i64, i64, i64, i64, vp, vp]),
        "msl_select_batch_slot": (C.c_int, [vp, i32]),
        "msl_propagate_frames": (C.c_int, [vp, i32, i32]),
        "msl_frame_batch": (C.c_int, [vp]),
        "msl_tacaw_stream_begin": (C.c_int, [vp, i32, i32, vp]),
        "msl_tacaw_stream_push": (C.c_int, [vp, i32, i32, i32]),
        "msl_tacaw_stream_finish": (C.c_int, [vp, vp]),
        "msl_tacaw_stream_set_reference": (C.c_int, [vp, vp, i32]),
        "msl_tacaw_stream_finish_range": (C.c_int, [vp, i32, i32, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.msl_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.msl_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                           "(python -m pyslice_amd.build_native --force)")
    _lib = lib
    return lib


def line_kernel_class(n: int) -> int:
    """2: power-of-two register kernel, 1: direct mixed-radix pass, 0: zero-padded convolution / generic kernel (msl_line_kernel_class)"""
    return int(load().msl_line_kernel_class(int(n)))


def fast_lengths(lo: int = 129, hi: int = 2048):
    """line lengths in [lo, hi] that run on a direct slice-loop kernel"""
    return [n for n in range(int(lo), int(hi) + 1) if line_kernel_class(n) > 0]


def _raise(rc, msg):
    if rc == MSL_ERR_INVALID:
        raise ValueError(msg)
    if rc == MSL_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == MSL_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One libmslice handle: one HIP device, one stream, all device buffers of one grid."""

    def __init__(self, nx, ny, nz, dx, dy, dz, wavelength, sigma, n_probes=1, n_frames=0, device=0,
                 keep_potential=False, fft_path=0, window=None, launch_timing=False, frame_batch=1, k_bin=None):
        self._lib = load()
        self._h = C.c_void_p()
        cfg = MslConfig(nx=int(nx), ny=int(ny), nz=int(nz), dx=float(dx), dy=float(dy), dz=float(dz),
                        wavelength=float(wavelength), sigma=float(sigma), n_probes=int(n_probes),
                        n_frames=int(n_frames), device=int(device), keep_potential=int(bool(keep_potential)),
                        fft_path=int(fft_path), window_nx=int(window[0]) if window else 0,
                        window_ny=int(window[1]) if window else 0, launch_timing=int(bool(launch_timing)),
                        frame_batch=int(frame_batch), bin_nx=int(k_bin[0]) if k_bin else 0, bin_ny=int(k_bin[1]) if k_bin else 0)
        rc = self._lib.msl_create(C.byref(cfg), C.byref(self._h))
        if rc != MSL_OK:
            msg = (self._lib.msl_last_error(None) or b"msl_create failed").decode()
            self._h = C.c_void_p()
            _raise(rc, msg)
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.n_probes, self.n_frames, self.device = int(n_probes), int(n_frames), int(device)
        self.keep_potential = bool(keep_potential)
        # stored shape of one exit-wave spectrum: the k-window, or the whole grid, divided by the detector bin
        self.wx = int(window[0]) if window and window[0] else self.nx
        self.wy = int(window[1]) if window and window[1] else self.ny
        if k_bin:
            self.wx //= max(1, int(k_bin[0]))
            self.wy //= max(1, int(k_bin[1]))
        self.intensity_F = 0               # frequency bins of the resident intensity buffer
        self.frame_batch = int(self._lib.msl_frame_batch(self._h))      # frames that share one sequence of launches

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.msl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != MSL_OK:
            _raise(rc, (self._lib.msl_last_error(self._h) or b"libmslice error").decode())

    # -- setup
    def set_kirkland(self, table):
        t = np.ascontiguousarray(table, dtype=np.float64)
        if t.size != 103 * 12:
            raise ValueError(f"Kirkland table must hold 103x3x4 values, got shape {t.shape}")
        self._chk(self._lib.msl_set_kirkland(self._h, _ptr(t)))

    def set_slices(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        if lo.shape != (self.nz,) or hi.shape != (self.nz,):
            raise ValueError(f"slice edges must have shape ({self.nz},)")
        self._chk(self._lib.msl_set_slices(self._h, _ptr(lo), _ptr(hi)))

    def set_beam(self, wavelength, sigma, dz):
        self._chk(self._lib.msl_set_beam(self._h, float(wavelength), float(sigma), float(dz)))

    def resize_probes(self, n_probes):
        self._chk(self._lib.msl_resize_probes(self._h, int(n_probes)))
        self.n_probes = int(n_probes)

    def set_probes(self, mrad, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        self._chk(self._lib.msl_set_probes(self._h, float(mrad), _ptr(xy), xy.shape[0]))

    def upload_probes(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.complex64)
        if a.ndim == 2:
            a = a[None]
        if a.shape[1:] != (self.nx, self.ny):
            raise ValueError(f"probe array must be (P,{self.nx},{self.ny}), got {a.shape}")
        self._chk(self._lib.msl_upload_probes(self._h, _ptr(a), a.shape[0]))

    def shift_probes(self, base, xy):
        b = np.ascontiguousarray(base, dtype=np.complex64)
        if b.shape != (self.nx, self.ny):
            raise ValueError(f"base probe must be ({self.nx},{self.ny}), got {b.shape}")
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        self._chk(self._lib.msl_shift_probes(self._h, _ptr(b), _ptr(xy), xy.shape[0]))

    # -- per frame
    def build_potential(self, positions, Z, slice_axis=2):
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        if pos.ndim != 2 or pos.shape[1] != 3:
            raise ValueError(f"positions must be (n_atoms,3), got {pos.shape}")
        z = np.ascontiguousarray(Z, dtype=np.int32)
        if z.shape != (pos.shape[0],):
            raise ValueError("one atomic number per atom required")
        axes = [0, 1, 2]
        if slice_axis not in axes:
            raise ValueError(f"slice_axis must be 0, 1 or 2, got {slice_axis}")
        axes.remove(slice_axis)
        self._chk(self._lib.msl_build_potential(self._h, _ptr(pos), _ptr(z), pos.shape[0], axes[0], axes[1], slice_axis))

    def build_potentials(self, positions, Z, slice_axis=2):
        """potentials of B = positions.shape[0] frames (B <= frame_batch) into the batch slots 0..B-1, one sequence of launches"""
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        if pos.ndim != 3 or pos.shape[2] != 3:
            raise ValueError(f"positions must be (n_frames,n_atoms,3), got {pos.shape}")
        z = np.ascontiguousarray(Z, dtype=np.int32)
        if z.shape != (pos.shape[1],):
            raise ValueError("one atomic number per atom required")
        axes = [0, 1, 2]
        if slice_axis not in axes:
            raise ValueError(f"slice_axis must be 0, 1 or 2, got {slice_axis}")
        axes.remove(slice_axis)
        self._chk(self._lib.msl_build_potentials(self._h, _ptr(pos), _ptr(z), pos.shape[1], pos.shape[0], axes[0], axes[1], slice_axis))

    def upload_potential(self, V_nz_nx_ny):
        v = np.ascontiguousarray(V_nz_nx_ny, dtype=np.float32)
        if v.shape != (self.nz, self.nx, self.ny):
            raise ValueError(f"potential must be ({self.nz},{self.nx},{self.ny}), got {v.shape}")
        self._chk(self._lib.msl_upload_potential(self._h, _ptr(v)))

    def propagate(self):
        self._chk(self._lib.msl_propagate(self._h))

    def propagate_frame(self, slot):
        self._chk(self._lib.msl_propagate_frame(self._h, int(slot)))

    def select_batch_slot(self, b):
        """transmission stack (0 <= b < frame_batch) the next build_potential / upload_potential fills"""
        self._chk(self._lib.msl_select_batch_slot(self._h, int(b)))

    def propagate_frames(self, first_slot, count):
        """slice loop + exit FFT of the frames in batch slots 0..count-1 -> frame slots first_slot..first_slot+count-1"""
        self._chk(self._lib.msl_propagate_frames(self._h, int(first_slot), int(count)))

    def tacaw(self, src_ptr=None, dst_ptr=None, batch=0, T=0, npix=0):
        self._chk(self._lib.msl_tacaw(self._h, C.c_void_p(src_ptr) if src_ptr else None,
                                      C.c_void_p(dst_ptr) if dst_ptr else None, int(batch), int(T), int(npix)))
        if not src_ptr:
            self.intensity_F = self.n_frames

    # -- streaming TACAW: accumulate the time->frequency transform for chosen bins, tile of frames by tile of frames
    def tacaw_stream_begin(self, T_total, bins=None):
        b = None if bins is None else np.ascontiguousarray(bins, dtype=np.int32).reshape(-1)
        self._chk(self._lib.msl_tacaw_stream_begin(self._h, int(T_total), 0 if b is None else b.size, _ptr(b) if b is not None else None))
        self._stream_F = int(T_total) if b is None else int(b.size)

    def tacaw_stream_push(self, first_slot, count, t0):
        self._chk(self._lib.msl_tacaw_stream_push(self._h, int(first_slot), int(count), int(t0)))

    def tacaw_stream_set_reference(self, slot=0, ref_ptr=None):
        """frames pushed afterwards are folded as Psi - ref: ref = frame slot `slot` of the ring, or a device (P,K) c64 array"""
        self._chk(self._lib.msl_tacaw_stream_set_reference(self._h, C.c_void_p(int(ref_ptr)) if ref_ptr else None, int(slot)))

    def tacaw_stream_finish_range(self, p0, count, dst_ptr, want_total=True):
        """finish the probes [p0, p0+count) only (frame-sharded runs after the reduce): intensity (count, n_bins, K) f32 into the
        device array at dst_ptr; -> (count, wx, wy) float64 total over all bins (or None)"""
        tot = np.empty((int(count), self.wx, self.wy), dtype=np.float64) if want_total else None
        self._chk(self._lib.msl_tacaw_stream_finish_range(self._h, int(p0), int(count), C.c_void_p(int(dst_ptr)) if dst_ptr else None,
                                                          _ptr(tot) if tot is not None and tot.size else None))
        self.intensity_F = self._stream_F
        return tot

    def tacaw_stream_finish(self, want_total=True):
        """-> (P, wx, wy) float64: sum over ALL frequency bins of the intensity (or None); the selected bins become the
        resident intensity buffer (P, n_bins, wx, wy)"""
        tot = np.empty((self.n_probes, self.wx, self.wy), dtype=np.float64) if want_total else None
        self._chk(self._lib.msl_tacaw_stream_finish(self._h, _ptr(tot) if tot is not None else None))
        self.intensity_F = self._stream_F
        return tot

    # -- reductions over resident results; src = (device pointer, B, F, K[, ld]) or None for the handle's own buffer; ld = pitch
    #    in elements between rows of K pixels (default K; a pointer into a library buffer goes with result_pitch())
    @staticmethod
    def _src(src):
        if src is None:
            return None, 0, 0, 0, 0
        ptr, B, F, K = src[:4]
        ld = src[4] if len(src) > 4 else K
        return C.c_void_p(int(ptr)), int(B), int(F), int(K), int(ld)

    def _bfk(self, src):
        return (self.n_probes, self.intensity_F, self.wx * self.wy) if src is None else tuple(int(v) for v in src[1:4])

    def result_pitch(self, what=BUF_WAVEFUNCTION):
        """pixel pitch of the images of the wavefunction / intensity buffer (>= wx*wy, include/mslice.h: msl_result_pitch)"""
        return int(self._lib.msl_result_pitch(self._h, int(what)))

    def result_view(self, what, typestr, rows=None):
        """DeviceArray of a result buffer as (P, rows, wx, wy) with the library's image pitch in its strides
        (rows: n_frames for the wavefunction, intensity_F for the intensity)"""
        pitch = self.result_pitch(what)
        if rows is None:
            rows = self.n_frames if what == BUF_WAVEFUNCTION else self.intensity_F
        es = int(typestr[2:])
        strides = (rows * pitch * es, pitch * es, self.wy * es, es)
        return DeviceArray(self.device_ptr(what), (self.n_probes, rows, self.wx, self.wy), typestr, owner=self, strides=strides)

    def tacaw_spectrum(self, mask=None, src=None):
        """(B,F) float64: sum over k of the (masked) intensity."""
        B, F, K = self._bfk(src)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask).reshape(-1) != 0, dtype=np.uint8)
            if m.size != K:
                raise ValueError(f"mask has {m.size} entries, k-space has {K}")
        out = np.empty((B, F), dtype=np.float64)
        p, b, f, k, ld = self._src(src)
        self._chk(self._lib.msl_tacaw_spectrum(self._h, p, b, f, k, ld, _ptr(m) if m is not None else None, _ptr(out)))
        return out

    def tacaw_spectrum_weighted(self, weight, src=None):
        """(B,F) float64: sum over k of weight[k] * intensity (any float mask)."""
        B, F, K = self._bfk(src)
        w = np.ascontiguousarray(np.asarray(weight, dtype=np.float64).reshape(-1))
        if w.size != K:
            raise ValueError(f"mask has {w.size} entries, k-space has {K}")
        out = np.empty((B, F), dtype=np.float64)
        p, b, f, k, ld = self._src(src)
        self._chk(self._lib.msl_tacaw_spectrum_weighted(self._h, p, b, f, k, ld, _ptr(w), _ptr(out)))
        return out

    def tacaw_diffraction(self, probes=None, freqs=None, scale=1.0, src=None):
        """(K,) float64: scale * sum of I[b,f,:] over the half-open probe and frequency ranges (None = all)."""
        B, F, K = self._bfk(src)
        b0, b1 = (0, B) if probes is None else probes
        f0, f1 = (0, F) if freqs is None else freqs
        out = np.empty(K, dtype=np.float64)
        p, b, f, k, ld = self._src(src)
        self._chk(self._lib.msl_tacaw_diffraction(self._h, p, b, f, k, ld, int(b0), int(b1), int(f0), int(f1), float(scale), _ptr(out)))
        return out

    def tacaw_dispersion(self, flat_indices, src=None):
        """(B,F,n) float32: I[b,f,idx[i]] for flat k indices kx*ny+ky."""
        B, F, K = self._bfk(src)
        idx = np.ascontiguousarray(flat_indices, dtype=np.int64).reshape(-1)
        out = np.empty((B, F, idx.size), dtype=np.float32)
        p, b, f, k, ld = self._src(src)
        self._chk(self._lib.msl_tacaw_dispersion(self._h, p, b, f, k, ld, _ptr(idx), idx.size, _ptr(out)))
        return out

    def adf(self, mask, src=None):
        """(B,) float64: mean over frames of sum_k mask |Psi|."""
        B, T, K = self._bfk(src)
        m = np.ascontiguousarray(np.asarray(mask).reshape(-1) != 0, dtype=np.uint8)
        if m.size != K:
            raise ValueError(f"mask has {m.size} entries, k-space has {K}")
        out = np.empty(B, dtype=np.float64)
        p, b, t, k, ld = self._src(src)
        self._chk(self._lib.msl_adf(self._h, p, b, t, k, ld, _ptr(m), _ptr(out)))
        return out

    # -- results
    def buffer_bytes(self, what):
        return int(self._lib.msl_buffer_bytes(self._h, int(what)))

    def device_ptr(self, what):
        return self._lib.msl_device_ptr(self._h, int(what)) or 0

    def download(self, what, dtype, shape, first=0, count=0):
        out = np.empty(shape, dtype=dtype)
        self._chk(self._lib.msl_download(self._h, int(what), _ptr(out), out.nbytes, int(first), int(count)))
        return out

    def probes(self):
        return self.download(BUF_PROBES, np.complex64, (self.n_probes, self.nx, self.ny))

    def exit_waves(self):
        return self.download(BUF_EXIT, np.complex64, (self.n_probes, self.nx, self.ny))

    def potential(self):
        return self.download(BUF_POTENTIAL, np.float32, (self.nz, self.nx, self.ny))

    def transmission(self):
        return self.download(BUF_TRANSMISSION, np.complex64, (self.nz, self.nx, self.ny))

    def wavefunction(self, first=0, count=0):
        n = count if count else self.n_probes
        return self.download(BUF_WAVEFUNCTION, np.complex64, (n, self.n_frames, self.wx, self.wy), first, count)

    def wavefunction_c128(self, n_frames_used=0):
        """(P, n_frames_used, wx, wy) complex128: the reference's result dtype, widened on the device (no host astype)"""
        T = int(n_frames_used) if n_frames_used else self.n_frames
        out = np.empty((self.n_probes, T, self.wx, self.wy), dtype=np.complex128)
        self._chk(self._lib.msl_download_wavefunction_c128(self._h, T, _ptr(out), out.nbytes))
        return out

    def frame(self, slot):
        out = np.empty((self.n_probes, self.wx, self.wy), dtype=np.complex64)
        self._chk(self._lib.msl_download_frame(self._h, int(slot), _ptr(out), out.nbytes))
        return out

    def upload_frame(self, slot, data):
        a = np.ascontiguousarray(data, dtype=np.complex64)
        if a.shape != (self.n_probes, self.wx, self.wy):
            raise ValueError(f"frame must be ({self.n_probes},{self.wx},{self.wy}), got {a.shape}")
        self._chk(self._lib.msl_upload_frame(self._h, int(slot), _ptr(a), a.nbytes))

    def intensity(self, first=0, count=0):
        n = count if count else self.n_probes
        return self.download(BUF_INTENSITY, np.float32, (n, self.intensity_F, self.wx, self.wy), first, count)

    def form_factors(self, n_species):
        return self.download(BUF_FORMFACTOR, np.float32, (n_species, self.nx, self.ny))

    def synchronize(self):
        self._chk(self._lib.msl_synchronize(self._h))

    def counters(self) -> dict:
        c = MslCounters()
        self._chk(self._lib.msl_get_counters(self._h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in MslCounters._fields_}

    def reset_counters(self):
        self._chk(self._lib.msl_reset_counters(self._h))

    def fft2(self, arr, direction=+1):
        a = np.ascontiguousarray(arr, dtype=np.complex64)
        if a.ndim == 2:
            a = a[None]
        if a.shape[1:] != (self.nx, self.ny):
            raise ValueError(f"array must be (B,{self.nx},{self.ny}), got {a.shape}")
        out = np.empty_like(a)
        self._chk(self._lib.msl_fft2_host(self._h, _ptr(a), _ptr(out), a.shape[0], int(direction)))
        return out


class DeviceArray:
    """Zero-copy view of a library device buffer for torch (``torch.as_tensor(DeviceArray(...), device='cuda')``)."""

    def __init__(self, ptr, shape, typestr, owner=None, strides=None):
        self._owner = owner
        dense, acc = [], int(typestr[2:])
        for n in reversed(shape):
            dense.append(acc)
            acc *= int(n)
        if strides is not None and tuple(int(v) for v in strides) == tuple(reversed(dense)):
            strides = None                              # C-contiguous: the interface wants None
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2,
                                         "strides": None if strides is None else tuple(int(v) for v in strides)}
