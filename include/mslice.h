/*
 * libmslice -- C ABI of the MI355X (gfx950) multislice engine.
 *
 * This is the drop-in boundary of the hot path.  The reference (h-walk/PySlice) has no
 * native layer: its path is Python calling torch/numpy ops.  Every entry point below
 * therefore replaces the *body* of one reference Python function; the reference-side
 * binding a maintainer would add is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative msl_status;
 *     no C++ exception crosses the ABI; msl_last_error() gives the message.
 *   - caller owns every host buffer; the library owns all device buffers for the life of
 *     the handle and keeps no host pointer after a call returns.
 *   - one handle == one HIP device + one HIP stream; a handle is not thread-safe, distinct
 *     handles (one per GPU / per process) are independent.
 *   - arithmetic is complex64 / float32 on the device ("c64" below = interleaved float
 *     re,im).  Setup scalars are taken as double and reduced on the host.
 *   - wave-function layout everywhere: [probe][x][y] with y fastest (reference axis order,
 *     src/multislice/multislice.py:285-294).
 */
#ifndef MSLICE_H
#define MSLICE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSL_ABI_VERSION 3   /* 2: reduction / streaming entry points and enum values added in rounds 2-3;
                              * 3: line-aligned pixel pitch of the result buffers (msl_result_pitch), `ld` argument of the reductions */

typedef struct msl_handle msl_handle;

typedef enum {
    MSL_OK = 0,
    MSL_ERR_INVALID = -1,     /* bad argument / shape (Python wrapper raises ValueError)   */
    MSL_ERR_HIP = -2,         /* HIP runtime failure (RuntimeError)                        */
    MSL_ERR_UNSUPPORTED = -3, /* e.g. grid length with a prime factor the FFT cannot do    */
    MSL_ERR_STATE = -4,       /* call order violated (e.g. propagate before potential)     */
    MSL_ERR_NOMEM = -5
} msl_status;

/* Grid + beam description.  Replaces the scalars MultisliceCalculator.setup() derives
 * (src/multislice/calculators.py:144-161) and Propagate() derives
 * (src/multislice/multislice.py:258-275). */
typedef struct {
    int32_t nx, ny, nz;        /* grid: len(xs), len(ys), number of slices                 */
    double  dx, dy;            /* xs[1]-xs[0], ys[1]-ys[0]   (potentials.py:228-229)       */
    double  dz;                /* slice spacing used by the Fresnel propagator (multislice.py:266) */
    double  wavelength;        /* Angstrom           (multislice.py:41-42)                 */
    double  sigma;             /* interaction parameter (multislice.py:258-260)            */
    int32_t n_probes;          /* P                                                         */
    int32_t n_frames;          /* T_local: frame slots of the (P,T_local,nx,ny) result; 0 = no result buffer */
    int32_t device;            /* HIP device ordinal                                        */
    int32_t keep_potential;    /* 1: also keep V (nz,nx,ny) float32 for msl_download(MSL_BUF_POTENTIAL) */
    int32_t fft_path;          /* 0 = auto (fast kernels when the size allows), 1 = generic LDS Stockham only */
    int32_t window_nx, window_ny; /* k-window (not in the reference, which keeps every pixel: calculators.py:161):
                                * keep only the central window_nx x window_ny pixels of each fftshifted exit-wave spectrum,
                                * rows [nx/2 - window_nx/2, +window_nx), columns likewise; the result, intensity and frame
                                * buffers then have shape (.., window_nx, window_ny).  0 = the full axis. */
    int32_t launch_timing;     /* 1: record a HIP event after every slice-loop launch so that msl_get_counters reports per-kernel
                                * launch counts and durations (bench.py's roofline); 0: no events (production) */
    int32_t frame_batch;       /* B > 1: up to B MD frames share every slice-loop launch (image index = frame * P + probe, each frame with
                                * its own transmission stack): the single-probe default of the reference (calculators.py:152-153) is
                                * otherwise launch-bound.  Potentials are built into batch slots (msl_select_batch_slot) and run by
                                * msl_propagate_frames.  Costs B x the work buffers and transmission stacks.  0 / 1 = off; ignored
                                * (treated as 1) with keep_potential. */
    int32_t bin_nx, bin_ny;    /* detector binning: every stored pixel is the sum of bin_nx x bin_ny neighbouring pixels of the
                                * fftshifted spectrum (of the k-window, when there is one; the window must be a multiple of the bin).  The
                                * result, intensity and frame buffers then have shape (.., wx/bin_nx, wy/bin_ny).  0 / 1 = off. */
    int32_t reserved[1];
} msl_config;

typedef enum {
    MSL_BUF_PROBES = 0,        /* (P,nx,ny) c64     initial probes psi_0                    */
    MSL_BUF_EXIT = 1,          /* (P,nx,ny) c64     real-space exit waves of the last msl_propagate */
    MSL_BUF_POTENTIAL = 2,     /* (nz,nx,ny) f32    V of the last msl_build_potential (slice-major!) */
    MSL_BUF_TRANSMISSION = 3,  /* (nz,nx,ny) c64    exp(i sigma V)                           */
    MSL_BUF_WAVEFUNCTION = 4,  /* (P,T_local,pitch) c64  fftshift(fft2(exit)) per frame slot: wx*wy pixels (wx,wy = nx,ny or the
                                *                    k-window) at a pitch of msl_result_pitch() pixels, the rest of a row is zero */
    MSL_BUF_INTENSITY = 5,     /* (P,T,pitch) f32   TACAW |FFT_t|^2 of the last msl_tacaw (same pitch; after a stream: pitch = wx*wy) */
    MSL_BUF_FORMFACTOR = 6,    /* (n_species,nx,ny) f32 Kirkland f_Z(q^2) of the last potential build */
    /* streaming TACAW, while a stream is open (msl_tacaw_stream_begin .. _finish): the partial sums of this handle, for the
     * caller's collective when the frames of a run are sharded over several handles / processes (msl_device_ptr only) */
    MSL_BUF_STREAM_ACC = 7,    /* (P,n_bins,K) c64  sum_t (Psi[p,t,k] - ref[p,k]) exp(-2 pi i u t / T) over the frames pushed so far */
    MSL_BUF_STREAM_S1 = 8,     /* (P,K) 2 x f64     sum_t Psi */
    MSL_BUF_STREAM_S2 = 9,     /* (P,K) f64         sum_t |Psi|^2 */
    MSL_BUF_STREAM_REF = 10    /* (P,K) c64         the reference pattern (msl_tacaw_stream_set_reference), NULL when none is set */
} msl_buffer;

typedef struct {
    uint64_t slice_steps;      /* probes x slices propagated since create/reset             */
    uint64_t frames;           /* frames propagated                                         */
    uint64_t algorithmic_bytes;/* 16 B x nx x ny per slice-step on the one-pass loop, 32 B on the two-pass loop (+ t and epilogue terms) */
    double   ms_potential;     /* device time (HIP events) spent in potential builds        */
    double   ms_propagate;     /* ... in slice loops (+ epilogue)                           */
    double   ms_tacaw;
    uint64_t slice_kernel_launches; /* launches of the dominant slice-loop kernels         */
    double   ms_slice_kernels; /* device time of those launches only (HIP events on the handle's stream) */
    uint64_t row_launches;     /* row-pass launches (ifft_y, x t, fft_y, x Py) and their device time */
    double   ms_row;
    uint64_t col_launches;     /* column-pass launches (fft_x, x Px, ifft_x) and their device time   */
    double   ms_col;
} msl_counters;

int  msl_abi_version(void);
/* Which slice-loop kernel a line of n points gets (the reference's grids are int(L / sampling) + 1 points, potentials.py:123-125,
 * so a user picks the line lengths with `sampling`): 2 = power-of-two register kernel (256, 512, 1024, 2048), 1 = direct
 * mixed-radix pass (99 lengths 135 ... 1728 with factors 2, 3, 5, 7), 0 = any other length: zero-padded convolution on the next
 * power-of-two transform (2-4 x the work per point) or the generic LDS kernel.  No handle, no device needed. */
int  msl_line_kernel_class(int32_t n);
/* Message of the last failure on this handle (or of the last failed msl_create when h==NULL). */
const char* msl_last_error(const msl_handle* h);

/* Create / destroy.  Replaces the allocation side of MultisliceCalculator.setup()
 * (calculators.py:154-161: base probe, wavefunction_data zeros). */
int  msl_create(const msl_config* cfg, msl_handle** out);
int  msl_destroy(msl_handle* h);

/* Kirkland parameter table, 103 elements x 3 terms x (a,b,c,d), row-major doubles.
 * Replaces loadKirkland() (potentials.py:134-185). */
int  msl_set_kirkland(msl_handle* h, const double* abcd_103x3x4);

/* Slice bin edges [lo[s], hi[s]) along the beam axis, nz doubles each.
 * Replaces the slice_min/slice_max rule of Potential.__init__ (potentials.py:302-307). */
int  msl_set_slices(msl_handle* h, const double* lo, const double* hi);

/* Change the beam after create: recomputes the Fresnel tables and, when a potential V is held
 * (keep_potential), re-derives exp(i sigma V).  Lets Potential() (which knows no beam energy,
 * potentials.py:188) be built first and Propagate() (multislice.py:258-275) supply the beam. */
int  msl_set_beam(msl_handle* h, double wavelength, double sigma, double dz);

/* Re-size the probe batch (re-allocates the (P,nx,ny) working buffers; probes must be set again). */
int  msl_resize_probes(msl_handle* h, int32_t n_probes);

/* Build the P shifted probes on the device: psi0[p] = ifft2(mask * exp(2 pi i (kx px + ky py)) * centre-shift).
 * mrad == 0 gives plane waves (ones).  xy = P x 2 doubles (Angstrom).
 * Replaces Probe.__init__ + create_batched_probes (multislice.py:112-124, 198-235). */
int  msl_set_probes(msl_handle* h, double mrad, const double* xy, int32_t n_probes);

/* Upload arbitrary initial waves (P,nx,ny) c64 (used when a caller hands Propagate() a
 * Probe built from its own array, multislice.py:104-109). */
int  msl_upload_probes(msl_handle* h, const float* c64, int32_t n_probes);

/* Shift an arbitrary base probe (nx,ny) c64 to P positions: psi0[p] = ifft2(fft2(base) * ramp_p).
 * Replaces create_batched_probes for probes built from a caller array (multislice.py:216-227). */
int  msl_shift_probes(msl_handle* h, const float* base_c64, const double* xy, int32_t n_probes);

/* Execution model: one handle = one HIP device + one stream.  With msl_config.launch_timing == 0 the per-frame calls
 * (msl_build_potential, msl_propagate, msl_propagate_frame) copy their host arguments into library-owned pinned memory,
 * queue the work on the stream and return; results are complete after msl_synchronize or any msl_download* call (which
 * wait for the stream).  With launch_timing == 1 they also wait, so that msl_get_counters can attribute time. */

/* Projected Kirkland potential + transmission functions of one MD frame.
 * pos = n x 3 doubles, Z = n atomic numbers (1..103); ax1/ax2 = in-plane axes, axs = slice axis.
 * Replaces Potential.__init__ (potentials.py:188-348) and the per-slice exp(i sigma V) of
 * Propagate (multislice.py:281-282). */
int  msl_build_potential(msl_handle* h, const double* pos, const int32_t* Z, int64_t n_atoms,
                         int32_t ax1, int32_t ax2, int32_t axs);

/* Upload a caller-made potential V (nz,nx,ny) float32 slice-major and derive exp(i sigma V)
 * (Propagate() accepts any Potential object, multislice.py:237). */
int  msl_upload_potential(msl_handle* h, const float* V_nz_nx_ny);

/* Slice loop for all probes against the current potential: nz transmissions, nz-1 Fresnel steps.
 * Leaves real-space exit waves in MSL_BUF_EXIT.  Replaces Propagate() (multislice.py:237-299). */
int  msl_propagate(msl_handle* h);

/* Slice loop + fused epilogue fftshift(fft2(exit)) written into frame slot `slot` of the
 * (P,T_local,nx,ny) result.  Replaces _process_frame_worker_torch after the potential
 * (calculators.py:281-290) and the scatter loop (calculators.py:185-186). */
int  msl_propagate_frame(msl_handle* h, int32_t slot);

/* Frame batching (msl_config.frame_batch = B > 1).  msl_select_batch_slot picks the transmission stack (0 <= b < B) that the
 * next msl_build_potential / msl_upload_potential fills and that msl_propagate / msl_propagate_frame use;
 * msl_propagate_frames runs the slice loop + exit FFT for the frames in batch slots 0..count-1 in ONE sequence of
 * launches and writes them to frame slots first_slot .. first_slot+count-1 of the (P,T_local,wx,wy) result.
 * msl_frame_batch returns the batch size the handle really uses (1 when batching is off).
 * Replaces count iterations of the reference's serial frame loop (calculators.py:172-186). */
int  msl_select_batch_slot(msl_handle* h, int32_t b);
int  msl_propagate_frames(msl_handle* h, int32_t first_slot, int32_t count);
/* The potentials of `count` MD frames (1 <= count <= frame_batch) into the batch slots 0 .. count-1 in ONE sequence of launches:
 * pos = count x n_atoms x 3 doubles (frame-major), Z = the n_atoms atomic numbers every frame shares (Trajectory.atom_types,
 * trajectory.py:8-14); axes as msl_build_potential.  Replaces `count` constructions of Potential (calculators.py:172-186 builds
 * one per frame, potentials.py:188-348) -- with the reference's default single probe the per-frame build is the frame. */
int  msl_build_potentials(msl_handle* h, const double* pos, const int32_t* Z, int64_t n_atoms, int32_t count,
                          int32_t ax1, int32_t ax2, int32_t axs);
int  msl_frame_batch(const msl_handle* h);

/* TACAW: intensity[p,w,kx,ky] = | fftshift_t fft_t( Psi - <Psi>_t ) |^2 over a (B,T,npix) c64 device
 * array.  src == NULL uses the handle's own wavefunction buffer (B=P, T=T_local, npix=nx*ny) and
 * its own intensity buffer.  With src/dst given (device pointers, e.g. the output of an RCCL
 * all-to-all held by the caller) the transform is applied to that memory.
 * Replaces TACAWData.fft_from_wf_data (tacaw_data.py:89-104). */
int  msl_tacaw(msl_handle* h, const void* d_src_c64, void* d_dst_f32, int64_t batch, int32_t T, int64_t npix);

/* Streaming TACAW: the time -> frequency transform accumulated tile of frames by tile of frames, for n_bins
 * chosen frequency bins, so that the handle holds a ring of frame slots (msl_config.n_frames = tile length) and the
 * accumulators instead of every frame (the reference needs the whole (P,T,nx,ny) array: tacaw_data.py:94-96).
 *   begin:  T_total = frames of the run, bins = n_bins unshifted FFT bin numbers u in [0,T_total) (NULL: all T_total);
 *   push:   folds the frames in slots [first_slot, first_slot+count), whose time indices are t0, t0+1, ..;
 *   finish: intensity[p,i,k] = | sum_t Psi[p,t,k] exp(-2 pi i bins[i] t / T) |^2 (0 for bin 0: mean subtraction) becomes the
 *           handle's intensity buffer, shape (P, n_bins, K), ready for the reductions below; total_PK (host, P*K float64,
 *           may be NULL) receives sum over ALL T_total bins of the intensity (Parseval: T sum|Psi|^2 - |sum Psi|^2), i.e.
 *           TACAWData.diffraction() of the full transform without any of it being stored. */
int  msl_tacaw_stream_begin(msl_handle* h, int32_t T_total, int32_t n_bins, const int32_t* bins);
int  msl_tacaw_stream_push(msl_handle* h, int32_t first_slot, int32_t count, int32_t t0);
int  msl_tacaw_stream_finish(msl_handle* h, double* total_PK);
/* Reference pattern of an open stream: every frame pushed afterwards is folded as Psi[p,t,k] - ref[p,k].  A time-independent
 * offset only changes the u = 0 bin, which the reference's mean subtraction (tacaw_data.py:94) zeroes anyway, so the result is
 * unchanged -- but the float32 accumulators then hold the thermal part instead of T Bragg amplitudes that have to cancel.
 * d_ref_c64: device (P,K) c64, or NULL to take frame slot `slot` of this handle's ring.  Call it before the first push; all
 * handles that share one run (frame shards) must use the SAME reference (MSL_BUF_STREAM_REF gives the pointer to broadcast). */
int  msl_tacaw_stream_set_reference(msl_handle* h, const void* d_ref_c64, int32_t slot);
/* Frame-sharded runs (one handle per GPU, each pushing its own frames with their global time indices t0): the partial sums
 * MSL_BUF_STREAM_ACC / _S1 / _S2 are linear in the frames, so the caller sum-reduces them over the handles (RCCL
 * reduce-scatter over probes: pyslice_amd/distributed.py) into the probe range [p0, p0+count) of THIS handle's buffers and
 * then finishes that range only: intensity (count, n_bins, K) f32 into d_dst_f32 (device; NULL only for the full range, which
 * goes to the handle's intensity buffer like msl_tacaw_stream_finish), total_host (count*K f64, may be NULL).  Closes the stream.
 * Replaces, together with the pushes, TACAWData.fft_from_wf_data on the (P,T,nx,ny) array no rank holds (tacaw_data.py:89-104). */
int  msl_tacaw_stream_finish_range(msl_handle* h, int32_t p0, int32_t count, void* d_dst_f32, double* total_host);

/* ---- consumers of the resident results: reductions that stream the array once on the device ----
 * The TACAW reductions take a (B,F,K) float32 intensity array: d_src == NULL selects the handle's own intensity buffer
 * (B=P, F=T after msl_tacaw or n_bins after msl_tacaw_stream_finish, K=stored pixels); otherwise a caller-held device pointer with the given shape.  Results are
 * written to HOST memory; sums are accumulated in float64 like the reference's.
 * ld = distance in elements between consecutive (b,f) rows of K pixels (0: K; ignored with d_src == NULL): the library's own
 * result buffers keep every image at a pitch of msl_result_pitch() >= K pixels, so a pointer INTO them goes with that ld.
 *
 * msl_tacaw_spectrum: out[b*F+f] = sum_k w(k) I[b,f,k], w = 1 or mask[k] != 0 (mask: K host bytes or NULL).
 *   Replaces the k-space sums of TACAWData.spectrum (tacaw_data.py:109-143), spectrum_image (:145-179) and
 *   masked_spectrum (:256-300); the mean over probes / the frequency pick is a lookup in the (B,F) result. */
int  msl_tacaw_spectrum(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, const uint8_t* mask, double* out);
/* msl_tacaw_spectrum_weighted: out[b*F+f] = sum_k weight[k] I[b,f,k] with K float64 host weights: a non-boolean mask of
 *   TACAWData.masked_spectrum, which multiplies the intensity (tacaw_data.py:286-296). */
int  msl_tacaw_spectrum_weighted(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, const double* weight, double* out);
/* msl_tacaw_diffraction: out[k] = scale * sum_{b0<=b<b1} sum_{f0<=f<f1} I[b,f,k]   (K float64).
 *   Replaces TACAWData.diffraction (tacaw_data.py:183-217: all f, one probe or scale=1/P over all probes) and
 *   spectral_diffraction (:219-254: one f). */
int  msl_tacaw_diffraction(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, int64_t b0, int64_t b1,
                           int64_t f0, int64_t f1, double scale, double* out);
/* msl_tacaw_dispersion: out[(b*F+f)*n + i] = I[b,f,idx[i]] for n flat k indices (kx*ny+ky) along a path (float32).
 *   Replaces the gather loop of TACAWData.dispersion (tacaw_data.py:302-353). */
int  msl_tacaw_dispersion(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, const int64_t* idx, int64_t n,
                          float* out);
/* msl_adf: out[b] = mean_t sum_k w(k) |Psi[b,t,k]| over a (B,T,K) complex64 array (NULL: the handle's wavefunction
 *   buffer); mask = the annulus q > collection_angle*1e-3/lambda as K host bytes.
 *   Replaces the masked |.| sum and frame mean of HAADFData.calculateADF (haadf_data.py:72-94). */
int  msl_adf(msl_handle* h, const void* d_src_c64, int64_t B, int64_t T, int64_t K, int64_t ld, const uint8_t* mask, double* out);

/* Copy a device buffer to the host (dst must hold `bytes` = full buffer size, see msl_buffer_bytes).
 * For MSL_BUF_WAVEFUNCTION / MSL_BUF_INTENSITY the host copy is dense -- (P,T,wx,wy), bytes = P*T*wx*wy*8 or *4, the pixel
 * pitch of the device buffer (msl_result_pitch) is dropped on the way -- and `first`/`count` select a probe range (count==0: all). */
int  msl_download(msl_handle* h, msl_buffer what, void* dst, size_t bytes, int64_t first, int64_t count);
/* The (P, T_local, nx, ny) result as complex128 -- the dtype the reference returns (calculators.py:161, 284-290: its arrays are
 * torch.complex128) -- for the frames [0, n_frames_used) of every probe: widened on the device chunk by chunk and copied out as
 * 16 B per element, instead of a complex64 download followed by a single-threaded astype on the host.
 * dst_c128: host, n_probes * n_frames_used * wx * wy complex128. */
int  msl_download_wavefunction_c128(msl_handle* h, int32_t n_frames_used, void* dst_c128, size_t bytes);
size_t msl_buffer_bytes(const msl_handle* h, msl_buffer what);
/* Pixel pitch of the images of MSL_BUF_WAVEFUNCTION / MSL_BUF_INTENSITY: wx*wy rounded up to a multiple of 32 pixels, so that
 * every (probe, frame) image starts on a 256-byte (c64) / 128-byte (f32) boundary and the time kernels of msl_tacaw read and
 * write whole lines on grids with odd pixel counts too (the reference's own test grid is 501 x 491, 00_probe.py:7-8; its
 * time FFT runs over exactly such arrays, tacaw_data.py:94-96).  The pad pixels hold zeros.  msl_download*, msl_*_frame and the
 * reductions with d_src == NULL hide the pitch; a caller that takes msl_device_ptr() builds its view with it.
 * The intensity buffer written by msl_tacaw_stream_finish is dense (pitch = wx*wy).  Other buffers: 0. */
int64_t msl_result_pitch(const msl_handle* h, msl_buffer what);
/* Raw device pointer of a library buffer, for zero-copy use by the caller's collective (RCCL). */
void* msl_device_ptr(msl_handle* h, msl_buffer what);

/* One frame slot of the (P,T_local,nx,ny) result, host (P,nx,ny) c64 <-> device.  Used by the opt-in frame cache
 * (reference: psi_data/torch_<key>/frame_<i>.npy written and re-read per frame, calculators.py:173, 259-260, 311). */
int  msl_download_frame(msl_handle* h, int32_t slot, void* dst_c64, size_t bytes);
int  msl_upload_frame(msl_handle* h, int32_t slot, const void* src_c64, size_t bytes);

/* Wait for everything queued on the handle's stream. */
int  msl_synchronize(msl_handle* h);
int  msl_get_counters(const msl_handle* h, msl_counters* out);
int  msl_reset_counters(msl_handle* h);

/* Batched 2-D FFT self-test entry (parity tests of the FFT kernels alone): in/out host (B,nx,ny) c64,
 * dir=+1 forward / -1 inverse (1/(nx*ny) normalised), path as msl_config.fft_path. */
int  msl_fft2_host(msl_handle* h, const float* in_c64, float* out_c64, int32_t batch, int32_t dir);

#ifdef __cplusplus
}
#endif
#endif /* MSLICE_H */
