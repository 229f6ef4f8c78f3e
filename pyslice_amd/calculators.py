"""MultisliceCalculator -- host mirror of src/multislice/calculators.py (the drop-in boundary).

`setup()` and `run()` keep the reference's signatures, defaults and the attributes callers read
(calculators.py:96-161, 163-250).  The per-frame work -- projected potential, probes, slice loop,
exit-wave FFT -- runs in the HIP library; the (P,T,nx,ny) result stays resident on the device
until the end of `run()`, when it is packed into a WFData with the reference's field names,
axis order, fftshift convention and (by default) dtype.

Multi-GPU: when torch.distributed is initialised, MD frames are sharded in contiguous blocks over
the ranks (one process per GPU); there is no collective on the data path, only one gather of the
shards at the end (pyslice_amd/distributed.py).
"""
from __future__ import annotations

import hashlib
import logging
import os
import time
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

from . import _native, distributed
from .multislice import Probe, interaction_sigma, wavelength
from .potentials import TORCH_AVAILABLE, _as_tensor, _device_index, gridFromTrajectory, loadKirkland, slice_edges, suggest_sampling
from .trajectory import Trajectory
from .wf_data import WFData

if TORCH_AVAILABLE:
    import torch

logger = logging.getLogger(__name__)


def default_frame_batch(n_probes: int, n_slices: int, nx: int, ny: int) -> int:
    """Frames per sequence of launches when the caller does not say (MultisliceCalculator(frame_batch=None), bench.py).

    About 256 images (probes x frames) per launch: a launch of the slice loop is one round of persistent workgroups over the
    256 CUs, and its fixed part (tables into LDS, the first un-prefetched line, the tail of the last round) is amortised over
    the items of a workgroup -- 64 probes x 1024^2 x 200 slices: 277 / 270 / 267 us per 64 images at 1 / 2 / 4 frames per
    launch.  Bounded by 48 GB for the two orientations of the batch's transmission stacks (a sixth of the card; single-probe runs
    still gain from 32 -> 128 frames per launch: 501^2 x 100 slices 160 k -> 167 k slice-steps/s) and 8 GB for the three work
    buffers; setup() halves the batch when the device cannot hold it next to the result."""
    by_images = -(-256 // max(1, n_probes))
    by_stacks = int(48e9 // (16.0 * n_slices * nx * ny))
    by_work = int(8e9 // (24.0 * nx * ny * max(1, n_probes)))
    batch = max(1, min(by_images, by_stacks, by_work))
    if batch >= 16:
        batch -= batch % 16               # whole rounds of work items over the CUs (16-line tiles, 256 CUs x 2 workgroups)
    return batch


def _widen_to_host(view, chunk_bytes=1 << 30):
    """complex64 device tensor (P, T, nx, ny) -> complex128 host tensor, widened on the device in chunks of about 1 GB of
    complex128 (falls back to widening on the host if the device cannot hold one chunk)."""
    # zeros, not empty: the copy engine faults fresh host pages in one by one (84 ms for the 393 MB of a 501 x 491 x 100-frame
    # result), a threaded memset touches them in 6 ms and the copy into touched pages takes 7
    out = torch.zeros(view.shape, dtype=torch.complex128)
    if view.numel() == 0:
        return out
    flat_in = view.reshape(view.shape[0] * view.shape[1], *view.shape[2:]) if view.is_contiguous() else None
    if flat_in is None:                                   # (a frame-sliced view of the engine's buffer: probe by probe)
        rows_in = [view[p] for p in range(view.shape[0])]
        rows_out = [out[p] for p in range(view.shape[0])]
    else:
        rows_in, rows_out = [flat_in], [out.reshape(flat_in.shape)]
    for src, dst in zip(rows_in, rows_out):
        per = max(1, src[0].numel() * 16)
        step = max(1, int(chunk_bytes // per))
        for i in range(0, src.shape[0], step):
            part = src[i:i + step]
            try:
                dst[i:i + step].copy_(part.to(torch.complex128))
            except (RuntimeError, MemoryError):           # no room for the widened chunk on the device
                dst[i:i + step].copy_(part.cpu().to(torch.complex128))
    return out


class MultisliceCalculator:

    def __init__(self, device=None, force_cpu=False, *, output="host", dtype="complex128", progress=True,
                 gather="rank0", cache=False, k_window=None, frame_batch=None, k_bin=None, stream_tile=None):
        """
        device / force_cpu: as the reference (calculators.py:41).  There is no CPU path here, so
        force_cpu=True raises.  Keyword-only extras (not in the reference):
          output   "host" (default; WFData.wavefunction_data is a CPU tensor like the reference) or
                   "device" (zero-copy torch view of the library's (P,T,nx,ny) buffer, complex64)
          dtype    "complex128" (default, reference dtype; upcast after download) or "complex64"
          gather   multi-process runs: "rank0" (default), "all" or "none" (keep the local frame shard)
          cache    True: per-frame cache files psi_data/torch_<key>/frame_<i>.npy, (P,nx,ny,1,1) complex128, in the
                   reference's own naming and format (calculators.py:140, 173, 259-260, 311): frames found there are
                   loaded instead of computed (resume), computed frames are written.  Off by default: the reference
                   writes 16*P*nx*ny bytes per frame to the CWD unconditionally (1 GB/frame at C3) and its key ignores
                   the atom positions (stale-hit hazard).
          k_window (wx, wy): keep only the central wx x wy pixels of every exit-wave spectrum (the detector window
                   around k = 0).  wavefunction_data becomes (P,T,wx,wy,1), kxs/kys are cropped to match,
                   and TACAWData / HAADFData work on the window.  Cuts the resident result by nx*ny/(wx*wy) -- the way
                   to hold 2048^2 x 1024-frame runs at all -- and the exit FFT only transforms the columns kept.
        """
        if force_cpu:
            raise NotImplementedError("pyslice_amd has no CPU path (force_cpu=True): use the reference for CPU runs")
        if output not in ("host", "device"):
            raise ValueError("output must be 'host' or 'device'")
        if dtype not in ("complex128", "complex64"):
            raise ValueError("dtype must be 'complex128' or 'complex64'")
        if gather not in ("rank0", "all", "none"):
            raise ValueError("gather must be 'rank0', 'all' or 'none'")
        self.device = device
        self._output, self._dtype, self._progress, self._gather = output, dtype, progress, gather
        self._cache = bool(cache)
        if k_window is not None:
            if len(k_window) != 2 or int(k_window[0]) < 1 or int(k_window[1]) < 1:
                raise ValueError("k_window must be two positive pixel counts (wx, wy)")
            if cache:
                raise ValueError("the frame cache stores full (P,nx,ny,1,1) frames: cache=True cannot be combined with k_window")
            k_window = (int(k_window[0]), int(k_window[1]))
        self._k_window = k_window
        if k_bin is not None:
            if len(k_bin) != 2 or int(k_bin[0]) < 1 or int(k_bin[1]) < 1:
                raise ValueError("k_bin must be two positive pixel counts (bx, by)")
            if cache:
                raise ValueError("the frame cache stores full (P,nx,ny,1,1) frames: cache=True cannot be combined with k_bin")
            k_bin = (int(k_bin[0]), int(k_bin[1]))
        self._k_bin = k_bin
        if stream_tile is not None and int(stream_tile) < 1:
            raise ValueError("stream_tile must be a positive frame count")
        if stream_tile is not None and cache:
            raise ValueError("stream_tile cannot be combined with cache=True")
        self._stream_tile = None if stream_tile is None else int(stream_tile)
        if frame_batch is not None and int(frame_batch) < 1:
            raise ValueError("frame_batch must be a positive frame count")
        self._frame_batch = None if frame_batch is None else int(frame_batch)
        self._engine = None
        # reference calculators.py:70-76 (display names for Z <= 36)
        self.element_map = {
            1: 'H', 2: 'He', 3: 'Li', 4: 'Be', 5: 'B', 6: 'C', 7: 'N', 8: 'O', 9: 'F', 10: 'Ne', 11: 'Na', 12: 'Mg',
            13: 'Al', 14: 'Si', 15: 'P', 16: 'S', 17: 'Cl', 18: 'Ar', 19: 'K', 20: 'Ca', 21: 'Sc', 22: 'Ti', 23: 'V',
            24: 'Cr', 25: 'Mn', 26: 'Fe', 27: 'Co', 28: 'Ni', 29: 'Cu', 30: 'Zn', 31: 'Ga', 32: 'Ge', 33: 'As',
            34: 'Se', 35: 'Br', 36: 'Kr'}

    def _generate_cache_key(self, trajectory, aperture, voltage_eV, slice_thickness, sampling, probe_positions):
        """reference calculators.py:78-94 (same recipe, so reference and build name the same directory)."""
        params = {
            'n_frames': trajectory.n_frames, 'n_atoms': trajectory.n_atoms,
            'box_matrix': trajectory.box_matrix.tolist(), 'atom_types': trajectory.atom_types.tolist(),
            'aperture': aperture, 'voltage_eV': voltage_eV, 'slice_thickness': slice_thickness,
            'sampling': sampling, 'probe_positions': probe_positions, 'backend': 'pytorch'}
        return hashlib.md5(str(sorted(params.items())).encode()).hexdigest()[:12]

    def setup(
        self,
        trajectory: Trajectory,
        aperture: float = 0.0,
        voltage_eV: float = 60e3,
        defocus: float = 0.0,
        slice_thickness: float = 0.5,
        sampling: float = 0.1,
        probe_positions: Optional[List[Tuple[float, float]]] = None,
        batch_size: int = 10,
        save_path: Optional[Path] = None,
        cleanup_temp_files: bool = False,
        slice_axis: int = 2,
    ):
        """reference calculators.py:96-161 -- same arguments, same defaults, same attributes."""
        self.trajectory = trajectory
        self.aperture = aperture
        self.voltage_eV = voltage_eV
        self.defocus = defocus                  # stored, never applied -- as in the reference (:129)
        self.slice_thickness = slice_thickness
        self.sampling = sampling
        self.probe_positions = probe_positions
        self.save_path = save_path
        self.cleanup_temp_files = cleanup_temp_files
        self.slice_axis = slice_axis

        cache_key = self._generate_cache_key(trajectory, aperture, voltage_eV, slice_thickness, sampling, probe_positions)
        self.output_dir = Path("psi_data") / f"torch_{cache_key}"    # created only when the frame cache is switched on
        if self._cache:
            self.output_dir.mkdir(parents=True, exist_ok=True)

        xs, ys, zs, lx, ly, lz = gridFromTrajectory(trajectory, sampling=sampling, slice_thickness=slice_thickness)
        nx, ny, nz = len(xs), len(ys), len(zs)
        self.xs, self.ys, self.zs = xs, ys, zs
        self.lx, self.ly, self.lz = lx, ly, lz
        self.nx, self.ny, self.nz = nx, ny, nz
        self.dx = xs[1] - xs[0]
        self.dy = ys[1] - ys[0]
        # (not in the reference) a line length without a slice-loop kernel of its own costs 2-4 x: name a nearby sampling that has one
        hint = suggest_sampling(trajectory, sampling)
        self.grid_hint = None if hint is None else (
            f"grid {nx} x {ny}: at least one axis runs as a zero-padded convolution; sampling={hint[0]:.6g} gives {hint[1]} x {hint[2]} "
            f"on direct kernels (modelled faster although finer)")
        if self.grid_hint and self._progress and distributed.rank_world()[0] == 0:
            print(self.grid_hint)

        if self.probe_positions is None:
            self.probe_positions = [(lx / 2, ly / 2)]
        self.base_probe = Probe(xs, ys, self.aperture, self.voltage_eV, device=self.device)

        self.n_frames = trajectory.n_frames
        self.n_probes = len(self.probe_positions)
        self.wavefunction_data = None           # filled by run(); the device holds (P,T_local,nx,ny) meanwhile

        # frame shard of this rank (contiguous block) and the device that serves it
        self._rank, self._world = distributed.rank_world()
        self._frames = distributed.shard_frames(self.n_frames, self._world, self._rank)
        dev = self.device
        if dev is None and self._world > 1:
            dev = int(os.environ.get("LOCAL_RANK", self._rank))
        # slice coordinates follow the slice axis (potentials.py:241-245); the Fresnel step uses zs (multislice.py:266)
        slice_coords = np.asarray([xs, ys, zs][slice_axis], dtype=np.float64)
        n_slices = len(slice_coords)
        dz = zs[1] - zs[0] if nz > 1 else 0.5
        # A previous run's WFData (and zero-copy device views of its buffers) may still hold the old engine: drop our
        # reference and let the last owner free it, instead of closing it under them.  (A caller that keeps an earlier result
        # and only needs its host arrays frees the device side with result.release().)
        self._engine = None
        batch = self._frame_batch
        if batch is None:
            batch = default_frame_batch(self.n_probes, n_slices, nx, ny)
        batch = 1 if self._cache else max(1, min(batch, len(self._frames)))
        slots = max(1, len(self._frames))
        if self._stream_tile is not None:
            slots = max(1, min(self._stream_tile, slots))
            batch = min(batch, slots)
        if self._k_bin is not None:
            wx, wy = self._k_window if self._k_window is not None else (nx, ny)
            if wx % self._k_bin[0] or wy % self._k_bin[1]:
                raise ValueError(f"the stored spectrum {wx} x {wy} is not a multiple of k_bin {self._k_bin}")
        # Headroom for what is allocated AFTER the engine exists: the phase tables of a frame group (up to 6 GB), the TACAW
        # intensity array (4 B per stored complex value), the streaming accumulators.  A default batch that passes msl_create
        # could otherwise run out of memory in the middle of a run (an explicit frame_batch is honoured as is).
        if self._frame_batch is None and batch > 1 and TORCH_AVAILABLE and torch.cuda.is_available():
            try:
                free_b = float(torch.cuda.mem_get_info(_device_index(dev))[0])
            except Exception:                              # pragma: no cover  (no device visible to torch: msl_create decides)
                free_b = None
            if free_b is not None:
                wx, wy = self._k_window if self._k_window is not None else (nx, ny)
                bx, by = self._k_bin if self._k_bin is not None else (1, 1)
                stored = float(self.n_probes) * slots * (wx // bx) * (wy // by)
                n_atoms = len(trajectory.atom_types)
                tables = min(6e9, batch * n_atoms * (nx // 2 + ny // 2 + 2) * 8.0)
                later = tables + 4.0 * stored + (24.0 * stored / slots if self._stream_tile is not None else 0.0)
                fixed = 8.0 * stored + later + 2e9
                per_frame = 16.0 * n_slices * nx * ny + 24.0 * nx * ny * self.n_probes
                while batch > 1 and fixed + batch * per_frame > 0.95 * free_b:
                    batch = max(1, batch // 2)
        # The frame batch costs batch x (two orientations of the transmission stack + three work buffers): when the device cannot
        # hold it next to the (P, T_local, wx, wy) result -- a result near capacity, a shared or smaller GPU -- halve it down to one
        # frame per launch sequence instead of failing a run that fits without batching (an explicit frame_batch is honoured as is)
        while True:
            try:
                self._engine = _native.Engine(nx, ny, n_slices, self.dx, self.dy, dz, wavelength(voltage_eV),
                                              interaction_sigma(voltage_eV), n_probes=self.n_probes,
                                              n_frames=slots, device=_device_index(dev),
                                              window=self._k_window, frame_batch=batch, k_bin=self._k_bin)
                break
            except MemoryError:
                if self._frame_batch is not None or batch <= 1:
                    raise
                batch = max(1, batch // 2)
                if self._stream_tile is not None:
                    batch = min(batch, slots)
                logger.info(f"device memory: frame batch reduced to {batch}")
        self._engine.set_kirkland(loadKirkland())
        lo, hi = slice_edges(slice_coords)
        self._engine.set_slices(lo, hi)
        self._engine.set_probes(self.aperture, np.asarray(self.probe_positions, dtype=np.float64))
        self._Z = np.asarray(trajectory.atom_types, dtype=np.int32)

    def run(self) -> WFData:
        """reference calculators.py:163-250: all frames, then pack WFData."""
        if self._engine is None:
            raise RuntimeError("call setup() before run()")
        if self._stream_tile is not None:
            raise RuntimeError("stream_tile is set: the device holds a ring of frames only -- call run_streaming_tacaw()")
        eng = self._engine
        t0 = time.time()
        frames = self._frames
        bar = None
        if self._progress and self._rank == 0:
            try:
                from tqdm import tqdm
                bar = tqdm(total=len(frames), desc="Processing frames", unit="frame")
            except ImportError:
                bar = None
        self.frames_computed = self.frames_cached = 0
        B = eng.frame_batch
        if B > 1:
            # batches of B frames: B potentials into the batch slots, then one slice loop over B x P images
            for s0 in range(0, len(frames), B):
                chunk = frames[s0:s0 + B]
                eng.build_potentials(self.trajectory.positions[chunk[0]:chunk[-1] + 1], self._Z, self.slice_axis)
                eng.propagate_frames(s0, len(chunk))
                self.frames_computed += len(chunk)
                if bar is not None:
                    bar.update(len(chunk))
            frames_iter = []
        else:
            frames_iter = list(enumerate(frames))
        for slot, frame_idx in frames_iter:
            cache_file = self.output_dir / f"frame_{frame_idx}.npy"
            if self._cache and cache_file.exists():
                eng.upload_frame(slot, np.load(cache_file)[:, :, :, 0, 0])
                self.frames_cached += 1
            else:
                eng.build_potential(self.trajectory.positions[frame_idx], self._Z, self.slice_axis)
                eng.propagate_frame(slot)
                self.frames_computed += 1
                if self._cache:
                    np.save(cache_file, eng.frame(slot).astype(np.complex128)[:, :, :, None, None])
            if bar is not None:
                bar.update(1)
        eng.synchronize()
        if bar is not None:
            bar.close()
        self.elapsed = time.time() - t0
        logger.info(f"Simulation completed in {self.elapsed:.2f}s ({self.frames_computed} computed, {self.frames_cached} cached)")

        # reference calculators.py:218-221 (quirk Q2: `sampling`, not dx; torch default float32)
        kxs, kys = self._k_axes()
        time_array = np.arange(self.n_frames) * self.trajectory.timestep
        layer_array = np.array([0])

        data, resident = self._collect()
        self.wavefunction_data = data
        wf = WFData(probe_positions=self.probe_positions, time=time_array, kxs=_as_tensor(kxs), kys=_as_tensor(kys),
                    layer=layer_array, wavefunction_data=data, probe=self.base_probe)
        # private riders: let TACAWData transform the device-resident copy without a host round trip
        wf._engine = eng
        wf._resident = resident
        wf._output = self._output
        if self._world > 1 and self._gather == "none":
            wf._frame_shard = (self.n_frames, len(frames))      # lets TACAWData do the all-to-all itself
        return wf

    def _k_axes(self):
        """kxs, kys of the stored spectra: reference calculators.py:218-219 (quirk Q2), cropped to the k-window (centred on
        the DC pixel, index n//2 after the shift) and averaged over every detector bin"""
        kxs = np.fft.fftshift(np.fft.fftfreq(self.nx, self.sampling)).astype(np.float32)
        kys = np.fft.fftshift(np.fft.fftfreq(self.ny, self.sampling)).astype(np.float32)
        if self._k_window is not None:
            wx, wy = self._k_window
            x0, y0 = self.nx // 2 - wx // 2, self.ny // 2 - wy // 2
            kxs, kys = kxs[x0:x0 + wx], kys[y0:y0 + wy]
        if self._k_bin is not None:
            kxs = kxs.reshape(-1, self._k_bin[0]).mean(axis=1).astype(np.float32)
            kys = kys.reshape(-1, self._k_bin[1]).mean(axis=1).astype(np.float32)
        return kxs, kys

    def run_streaming_tacaw(self, freq_window=None, bins=None):
        """Streaming TACAW (needs stream_tile): all frames are propagated through a ring of `stream_tile` frame slots and
        folded, tile by tile, into the time->frequency transform of the selected bins.

        freq_window = (lo, hi): keep the bins of TACAWData.frequencies (fftshifted, reference tacaw_data.py:84-85) with
        lo <= f <= hi;  bins = explicit indices into that fftshifted axis;  neither: all T bins.
        Returns a TACAWData whose `frequencies` / `intensity` hold the selected bins only ((P,F,wx,wy), device-resident
        for the reductions), plus `total_diffraction` (P,wx,wy): the sum over ALL T bins (Parseval), i.e. what
        TACAWData.diffraction() of the full transform returns, without the transform being stored."""
        from .tacaw_data import TACAWData
        if self._engine is None:
            raise RuntimeError("call setup() before run_streaming_tacaw()")
        if self._stream_tile is None:
            raise RuntimeError("run_streaming_tacaw() needs MultisliceCalculator(stream_tile=Tt)")
        eng, T = self._engine, self.n_frames
        if T < 2:
            raise ValueError("TACAW needs at least 2 frames")
        time_array = np.arange(T) * self.trajectory.timestep
        freqs = np.fft.fftshift(np.fft.fftfreq(T, d=time_array[1] - time_array[0]))
        if bins is not None:
            sel = np.asarray(bins, dtype=np.int64).reshape(-1)
            if sel.size == 0 or sel.min() < 0 or sel.max() >= T:
                raise ValueError(f"bins must be indices into the {T} fftshifted frequencies")
        elif freq_window is not None:
            sel = np.nonzero((freqs >= freq_window[0]) & (freqs <= freq_window[1]))[0]
            if sel.size == 0:
                raise ValueError(f"no frequency bin inside {freq_window}")
        else:
            sel = np.arange(T)
        unshifted = (sel + (T + 1) // 2) % T              # fftshifted index s holds FFT bin (s + ceil(T/2)) mod T
        t0 = time.time()
        eng.tacaw_stream_begin(T, unshifted)
        ring, B = eng.n_frames, eng.frame_batch
        # This rank's MD frames (all of them in a single-process run): propagated through the ring tile by tile and folded with
        # their GLOBAL time indices.  The first frame of the run is the reference pattern every rank subtracts before folding
        # (msl_tacaw_stream_set_reference): rank 0 takes it from its first tile and broadcasts it (P x stored pixels, once).
        frames = self._frames
        have_ref = False
        for tile0 in range(0, max(len(frames), 1), ring):
            tile = frames[tile0:tile0 + ring]
            for s0 in range(0, len(tile), B):
                chunk = tile[s0:s0 + B]
                if B > 1:
                    eng.build_potentials(self.trajectory.positions[chunk[0]:chunk[-1] + 1], self._Z, self.slice_axis)
                    eng.propagate_frames(s0, len(chunk))
                else:
                    eng.build_potential(self.trajectory.positions[chunk[0]], self._Z, self.slice_axis)
                    eng.propagate_frame(s0)
            if not have_ref:
                self._stream_reference(eng)
                have_ref = True
            if tile:
                eng.tacaw_stream_push(0, len(tile), tile[0])
        kxs, kys = self._k_axes()
        tac = TACAWData.__new__(TACAWData)
        tac.__dict__.update(dict(probe_positions=self.probe_positions, time=time_array, kxs=_as_tensor(kxs), kys=_as_tensor(kys),
                                 layer=np.array([0]), wavefunction_data=None, probe=self.base_probe,
                                 frequencies=freqs[sel], frequency_bins=sel, _engine=eng, _output=self._output))
        if self._world > 1:
            self._finish_stream_sharded(eng, tac)
            self.elapsed = time.time() - t0
            return tac
        total = eng.tacaw_stream_finish(True)
        self.elapsed = time.time() - t0
        tac.total_diffraction = total
        tac._intensity_src = (eng, None)
        if self._output == "device":
            tac.intensity = torch.as_tensor(eng.result_view(_native.BUF_INTENSITY, "<f4"), device=f"cuda:{eng.device}")
        else:
            tac.intensity = _as_tensor(eng.intensity().astype(np.float64))
        return tac

    def _stream_reference(self, eng):
        """the run's first frame (frame slot 0 of rank 0's first tile) becomes the reference pattern of every rank's fold"""
        if self._world == 1:
            eng.tacaw_stream_set_reference(slot=0)
            return
        dev = torch.device("cuda", eng.device)
        P, K = eng.n_probes, eng.wx * eng.wy
        if self._rank == 0:
            eng.tacaw_stream_set_reference(slot=0)
            eng.synchronize()
            ref = torch.as_tensor(_native.DeviceArray(eng.device_ptr(_native.BUF_STREAM_REF), (P, K), "<c8", owner=eng), device=dev)
            distributed.broadcast_from(ref, src=0)
        else:
            ref = torch.empty((P, K), dtype=torch.complex64, device=dev)
            distributed.broadcast_from(ref, src=0)
            torch.cuda.synchronize(dev)
            eng.tacaw_stream_set_reference(ref_ptr=ref.data_ptr())
            eng.synchronize()                     # the copy out of `ref` is done before the tensor goes away

    def _finish_stream_sharded(self, eng, tac):
        """Frame-sharded streaming TACAW: sum the ranks' partial sums (reduce-scatter over probes, distributed.reduce_probes),
        finish this rank's probes, gather the intensities (distributed.gather_probes).  tacaw_data.py:89-104 on an array no
        rank ever holds."""
        dev = torch.device("cuda", eng.device)
        P, F, K = eng.n_probes, len(tac.frequency_bins), eng.wx * eng.wy
        eng.synchronize()

        def view(what, shape, typestr):
            return torch.as_tensor(_native.DeviceArray(eng.device_ptr(what), shape, typestr, owner=eng), device=dev)
        p0, p1 = distributed.reduce_probes(view(_native.BUF_STREAM_ACC, (P, F, K), "<c8"), P)
        distributed.reduce_probes(view(_native.BUF_STREAM_S1, (P, K, 2), "<f8"), P)
        distributed.reduce_probes(view(_native.BUF_STREAM_S2, (P, K), "<f8"), P)
        torch.cuda.synchronize(dev)
        mine = torch.empty((p1 - p0, F, eng.wx, eng.wy), dtype=torch.float32, device=dev)
        total = eng.tacaw_stream_finish_range(p0, p1 - p0, mine.data_ptr(), True)       # (an empty shard still closes the stream)
        tot = torch.from_numpy(total).to(dev)
        tac.probe_range = (p0, p1)
        if self._gather == "none":
            # this rank keeps ITS probes only: the TACAWData's probe list is the shard's (its methods index probes by
            # len(probe_positions)); probe_range holds the shard's place in the run's probe list
            full, tot_full = mine, tot
            tac.probe_positions = list(tac.probe_positions)[p0:p1]
        else:
            dst = None if self._gather == "all" else 0
            full = distributed.gather_probes(mine, P, dst=dst)
            tot_full = distributed.gather_probes(tot, P, dst=dst)
        if full is None:
            tac.intensity, tac.total_diffraction = None, None
            return
        tac._intensity_src = (eng, full)
        tac.total_diffraction = tot_full.cpu().numpy()
        tac.intensity = full if self._output == "device" else full.to(torch.float64).cpu()

    # ------------------------------------------------------------------------------------------
    def _collect(self):
        """Pack the device-resident (P,T_local,nx,ny) into the reference's (P,T,nx,ny,1) array."""
        eng = self._engine
        P, nx, ny = self.n_probes, eng.wx, eng.wy          # stored spectrum shape (the k-window, or the grid)
        T_local = len(self._frames)
        if self._world == 1 or self._gather == "none":
            if self._output == "device":
                # (the images sit at the library's line-aligned pixel pitch: a strided view when nx*ny is not a multiple of 32)
                view = torch.as_tensor(eng.result_view(_native.BUF_WAVEFUNCTION, "<c8"), device=f"cuda:{eng.device}")
                return view[:, :T_local].unsqueeze(-1), T_local == eng.n_frames
            if self._dtype == "complex128":
                # the reference's dtype: widened on the device and copied out as complex128 (msl_download_wavefunction_c128) --
                # the single-threaded numpy astype on the host took twice the whole multislice run of the default single-probe
                # case (501 x 491 x 100 frames: run() 0.171 s, of which 0.051 s on the GPU)
                return _as_tensor(eng.wavefunction_c128(T_local)[..., None]), T_local == eng.n_frames
            local = eng.wavefunction()[:, :T_local]
            return _as_tensor(np.ascontiguousarray(local[..., None])), T_local == eng.n_frames
        # multi-process: one gather of the frame shards (no collective during the frames)
        local = torch.as_tensor(eng.result_view(_native.BUF_WAVEFUNCTION, "<c8"), device=f"cuda:{eng.device}")[:, :T_local]
        full = distributed.gather_frames(local, self.n_frames, dst=None if self._gather == "all" else 0)
        if full is None:
            return None, False
        if self._output == "device":
            return full.unsqueeze(-1), False
        full = _widen_to_host(full) if self._dtype == "complex128" else full.cpu()
        return full.unsqueeze(-1), False
