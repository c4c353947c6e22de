/*
 * p3d.h -- C ABI of libp3d_hip.so: the MI355X (gfx950) implementation of the POCS hot path of
 * fwrnke/pseudo-3D-interpolation.
 *
 * The reference has no FFI layer; its boundary for this path is the Python callable
 *     POCS_algorithm(x, mask, auxiliary_data, transform, itransform, transform_kind, niter, thresh_op,
 *                    thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, verbose, version,
 *                    results_dict, path_results)         pseudo_3D_interpolation/functions/POCS.py:371-391
 * called once per (iline, xline) slice by xr.apply_ufunc(..., vectorize=True)
 *                                                        pseudo_3D_interpolation/cube_POCS_interpolation_3D.py:314-340
 * The entry points below are what a ctypes binding of that callable needs (INTEGRATION.md shows
 * the stub): plain pointers and sizes, no Python / torch types.
 *
 * Conventions
 *   - every function returns P3D_OK (0) or a negative error code; p3d_last_error() gives the text
 *     (thread-local);
 *   - a "slice" is one (nil x nxl) array, C-contiguous, xline fastest; a cube is [nslices][nil][nxl]
 *     (the slice-major layout written by cube_binning_3D.py:1313-1351);
 *   - *_dev functions take DEVICE pointers (hipMalloc / p3d_malloc / torch data_ptr), the
 *     un-suffixed ones take HOST pointers and stage through buffers owned by the plan;
 *   - a plan is bound to one device and one internal stream and is not re-entrant; different plans
 *     may be used from different host threads / processes;
 *   - all calls are synchronous: when they return, outputs are complete.
 */
#ifndef P3D_H
#define P3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_ABI_VERSION 1

#define P3D_OK 0
#define P3D_ERR_INVALID (-1)     /* bad argument */
#define P3D_ERR_UNSUPPORTED (-2) /* shape / option not covered by the HIP kernels */
#define P3D_ERR_HIP (-3)         /* a HIP runtime call failed (no GPU, out of memory, ...) */

/* element type of the observed cube `x` and of the result (np.iscomplexobj(x), POCS.py:511, 653-656) */
#define P3D_C64 0 /* complex64: frequency-domain cube (dim 'freq_twt') */
#define P3D_F32 1 /* float32 : time-domain cube (dim 'twt'); result is np.real() of the iterate */
#define P3D_C128 2 /* complex128 and */
#define P3D_F64 3  /* float64 cubes: the double-precision entry points only (p3d_pocs64_*) */

/* thresh_op (POCS.py:91-102 -> threshold_operator.py:9-112) */
#define P3D_OP_HARD 0
#define P3D_OP_SOFT 1
#define P3D_OP_GARROTE 2
/* '-percentile' variants (POCS.py:43-58, 95-102): Re(tau) of every iteration is a PERCENTAGE; the threshold applied to a
 * slice is np.percentile(abs(X), tau) of that slice's spectrum at that iteration (linear interpolation).  These run on the
 * unfused pipeline (the spectrum has to exist in memory to be ranked). */
#define P3D_OP_PERCENTILE 16 /* OR-ed with P3D_OP_HARD / _SOFT / _GARROTE */

/* version (POCS.py:564-575).  FAST is accepted and runs REGULAR: in the reference the momentum
 * term of 'fast' is identically zero (POCS.py:549-550, 566-571, 629). */
#define P3D_VER_REGULAR 0
#define P3D_VER_FAST 1
#define P3D_VER_ADAPTIVE 2

/* p3d_pocs_params.flags */
#define P3D_FLAG_PROFILE 1 /* bracket every kernel launch with HIP events; read with p3d_last_profile() */
#define P3D_FLAG_PRIMED 2  /* p3d_pocs_prime_dev has just run on exactly this cube and mask: p3d_pocs_run_dev may skip its first pass */

typedef struct p3d_plan p3d_plan;

typedef struct p3d_pocs_params {
    int32_t niter;     /* number of iterations (POCS.py:560) */
    int32_t thresh_op; /* P3D_OP_* */
    int32_t version;   /* P3D_VER_* */
    int32_t flags;     /* P3D_FLAG_* */
    double eps;        /* early exit: iiter > 2 and cost < eps (POCS.py:631); 0 disables */
    double alpha;      /* re-insertion weight (POCS.py:616-619) */
} p3d_pocs_params;

/* number of doubles per slice written by p3d_pocs_stats*: everything get_threshold_decay
 * (POCS.py:169-368) needs from X0 = fft2(x):
 *   [0] Re, [1] Im of the lexicographic max of X0 (numpy's complex .max(), POCS.py:288)
 *   [2] max |X0|, [3] min |X0|   (POCS.py:261-262)
 *   [4] sum |X0|^2               (POCS.py:299)
 *   [5] reserved (0)                                                                       */
#define P3D_STATS_PER_SLICE 6

int p3d_abi_version(void);
const char* p3d_last_error(void);
/* HIP_VERSION this library was compiled against and the version of the HIP runtime the process bound (they differ when another copy of
 * libamdhip64 was mapped first; the Python binding warns when major.minor disagree) */
int p3d_runtime_info(int* compiled_hip_version, int* runtime_hip_version);
int p3d_device_count(int* n);

/* 1 when the HIP kernels cover an (nil, nxl) slice shape, else 0 */
int p3d_shape_supported(int nil, int nxl);

int p3d_plan_create(p3d_plan** out, int device, int nil, int nxl, int max_slices);
int p3d_plan_destroy(p3d_plan* plan);

/* device-memory helpers so that a pure-ctypes caller needs no other GPU runtime */
int p3d_malloc(p3d_plan* plan, void** dptr, size_t bytes);
int p3d_free(p3d_plan* plan, void* dptr);   /* `plan` may be NULL (the buffer outlived its plan) */
int p3d_memcpy_h2d(p3d_plan* plan, void* dst_dev, const void* src_host, size_t bytes);
int p3d_memcpy_d2h(p3d_plan* plan, void* dst_host, const void* src_dev, size_t bytes);
/* The same without a plan, for callers that keep whole cubes resident in HBM across several plans (bench.py, a pipeline that chains
 * steps 12 -> 13 -> 14 on device buffers) and use no other GPU runtime: blocking calls on the device's null stream.
 * kind: 0 host -> device, 1 device -> host, 2 device -> device. */
int p3d_dev_malloc(int device, void** dptr, size_t bytes);
int p3d_dev_free(void* dptr);
int p3d_dev_memcpy(int device, void* dst, const void* src, size_t bytes, int kind);
int p3d_dev_memset(int device, void* dptr, int value, size_t bytes);
int p3d_dev_synchronize(int device);
int p3d_dev_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* page-locked host memory: copies to / from it run at the PCIe rate (pageable NumPy memory is staged by the runtime at a
 * quarter of it); the chunk pipeline of pocs_cube keeps its staging buffers here */
int p3d_host_alloc(void** hptr, size_t bytes);
int p3d_host_free(void* hptr);
/* page-lock a caller's array IN PLACE for the duration of a job (the host-buffer entry point of the Python mirror,
 * functions/POCS.py pocs_cube, does this with the cube it is handed and the result it returns -- the stand-in for the dask workers'
 * netCDF chunks of cube_POCS_interpolation_3D.py:314-340): transfers to / from it are DMA both ways at once.
 * P3D_ERR_UNSUPPORTED when the runtime refuses the range (the caller then simply goes on with pageable memory). */
int p3d_host_register(void* hptr, size_t bytes);
int p3d_host_unregister(void* hptr);

/* Batched 2-D FFT of complex64 slices, numpy.fft.fft2 / ifft2 conventions (unnormalised forward,
 * 1/(nil*nxl) inverse).  Replaces the callables injected at cube_POCS_interpolation_3D.py:255-257;
 * exported as a test hook.  in == out is allowed. */
int p3d_fft2_c64_dev(p3d_plan* plan, const void* in_dev, void* out_dev, int nslices, int inverse);
int p3d_fft2_c64(p3d_plan* plan, const void* in_host, void* out_host, int nslices, int inverse);

/* Thresholded spectrum of every slice: threshold(fft2(x), tau_s, kind) as the first half of one POCS
 * iteration computes it (POCS.py:592-599 -> threshold_operator.py:9-112).  tau: HOST [nslices][2]
 * doubles (Re, Im).  Test hook: lets a checker see the keep/zero decision of every coefficient. */
int p3d_fft2_shrink_c64(p3d_plan* plan, const void* in_host, const double* tau, int thresh_op, void* out_host,
                        int nslices);

/* Per-slice statistics of X0 = fft2(x) for the threshold schedule (replaces the device-independent
 * part of get_threshold_decay, POCS.py:535-546).  stats_host: [nslices][P3D_STATS_PER_SLICE]. */
int p3d_pocs_stats_dev(p3d_plan* plan, const void* x_dev, int dtype, int nslices, double* stats_host);
/* p3d_pocs_stats_dev for a caller that is about to run the job on the same cube: with the mask at hand the statistics pass IS the
 * first pass of the job (forward row transform into the work buffer, compact copy of the observed samples, sum |x_obs|), so a
 * following p3d_pocs_run_dev(..., flags | P3D_FLAG_PRIMED) on the same plan, pointers, dtype and batch skips it (one full pass over
 * the cube per job).  The flag is a promise about the CONTENTS of x_dev and mask_dev (unchanged in between); the plan checks the
 * rest and runs the ordinary first pass when anything else used it meanwhile, for APOCS and for float32 cubes that take the
 * half-spectrum path.  Same statistics, same results, bit for bit. */
int p3d_pocs_prime_dev(p3d_plan* plan, const void* x_dev, int dtype, const float* mask_dev, int nslices, double* stats_host);
int p3d_pocs_stats(p3d_plan* plan, const void* x_host, int dtype, int nslices, double* stats_host);

/* thresh_model = 'data-driven' (get_threshold_decay, functions/POCS.py:356-362): the schedule is read off the sorted forward
 * transform of the slice, complex numbers in NumPy's lexicographic order.  Two steps, so that the bounds are formed by the caller
 * in the reference's own arithmetic (tau = p * x_fwd.max(), complex64):
 *   p3d_pocs_sorted_spectrum   x (HOST or device, complex64 [nslices][nil][nxl]) -> fft2 -> sorted per slice, descending, kept
 *                              in the plan's staging buffers; peaks_host [nslices][2] = x_fwd.max() (POCS.py:288)
 *   p3d_pocs_data_driven_pick  bounds_host [nslices][4] = tau_min (re, im), tau_max (re, im) ->
 *                              count_host [nslices] = Nv = #{tau_min < X < tau_max} (0: the reference raises IndexError),
 *                              tau_host [nslices][niter][2] = v[0], v[ceil(i (Nv - 1) / (niter - 1))] (float32 pairs)
 * The pick must follow the sort directly (any other call on the plan in between invalidates the sorted keys: P3D_ERR_INVALID). */
int p3d_pocs_sorted_spectrum(p3d_plan* plan, const void* x, int nslices, float* peaks_host);
int p3d_pocs_data_driven_pick(p3d_plan* plan, int nslices, int niter, const float* bounds_host, float* tau_host, int64_t* count_host);

/* The POCS loop (POCS.py:549-632) for a batch of slices sharing one trace mask.
 *   x        [nslices][nil][nxl] observed data, zeros at missing traces, dtype as given
 *   mask     [nil][nxl] float32, 1 = observed trace, 0 = missing (cube_POCS_interpolation_3D.py:242-244)
 *   tau      HOST, [nslices][niter][2] doubles: Re, Im of the threshold used at each iteration
 *            (decay[k], or sqrt(decay[k]) for sqrt_decay; POCS.py:595)
 *   active   HOST, [nslices] uint8 or NULL: 0 marks an all-zero slice, which the reference returns
 *            untouched with niterations = 0 (POCS.py:515-521)
 *   out      [nslices][nil][nxl], same dtype as x
 *   niter_done  HOST [nslices] int32: iterations executed per slice (POCS.py:634)
 *   sums     HOST [(niter+1)][nslices] doubles or NULL: sums[0][s] = sum|x_s|, sums[k+1][s] =
 *            sum|x_k| after iteration k (0 where not executed); cost_k = ((S_k+1 - S_k)/S_k+1)^2
 *            (POCS.py:622)
 *   elapsed_ms  device time of the whole call measured with HIP events on the plan's stream, or NULL */
int p3d_pocs_run_dev(p3d_plan* plan, const void* x_dev, int dtype, const float* mask_dev, const double* tau,
                     const uint8_t* active, const p3d_pocs_params* params, void* out_dev, int nslices,
                     int32_t* niter_done, double* sums, double* elapsed_ms);
int p3d_pocs_run(p3d_plan* plan, const void* x_host, int dtype, const float* mask_host, const double* tau,
                 const uint8_t* active, const p3d_pocs_params* params, void* out_host, int nslices,
                 int32_t* niter_done, double* sums, double* elapsed_ms);

/* The same two calls for a cube in HOST memory spread over several devices from ONE process: contiguous blocks of the slice axis
 * (the split of sharding.slice_block) go to devices[0 .. ndev-1], one host thread and one plan per entry (a device may be listed
 * more than once), chunks of ~256 MiB; every device copies its own block back, there is no collective.  The reference's
 * counterpart is the dask worker farm over slices (cube_POCS_interpolation_3D.py:291-340).  Arguments as p3d_pocs_stats /
 * p3d_pocs_run; tau is required. */
int p3d_multi_stats(int ndev, const int* devices, int nil, int nxl, const void* x_host, int dtype, int nslices, double* stats_host);
int p3d_multi_run(int ndev, const int* devices, int nil, int nxl, const void* x_host, int dtype, const float* mask_host,
                  const double* tau, const uint8_t* active, const p3d_pocs_params* params, void* out_host, int nslices,
                  int32_t* niter_done, double* sums);

/* The same loop in the REFERENCE's precision.  POCS_algorithm computes in complex128 / float64 whenever its input is, under NumPy < 2 for every
 * input, and under NumPy >= 2 for the soft / garrote operators, FPOCS and APOCS on complex64 / float32 input as well (threshold_operator.py:37-39,
 * 76-78; POCS.py:566-575, 616: tau, the momentum scalar and the weights 1 - alpha * mask are float64 / complex128).  A plan64 runs
 * fft2 -> threshold -> ifft2 -> re-insertion -> cost in double precision for any slice extents up to 5120: two fused kernels per iteration on tiles
 * of lines in LDS (extents beyond 5088, whose tiles do not fit: six plain passes over the cube).  A precision path, bound by its double-precision
 * butterflies: about a tenth of the float32 rate.  dtype of x and out: P3D_C128, P3D_F64, or P3D_C64 / P3D_F32 (converted on load / store: what the reference's final
 * cast to the input dtype does, cube_POCS_interpolation_3D.py:324); x, out, mask may be host or device pointers; mask is DOUBLE [nil][nxl];
 * tau [nslices][niter][2], stats, sums and the error behaviour as p3d_pocs_stats / p3d_pocs_run; thresh_op: hard, soft, garrote. */
typedef struct p3d_plan64 p3d_plan64;
int p3d_plan64_create(p3d_plan64** out, int device, int nil, int nxl, int max_slices);
int p3d_plan64_destroy(p3d_plan64* plan);
int p3d_pocs64_stats(p3d_plan64* plan, const void* x, int dtype, int nslices, double* stats_host);
/* test hook: batched fft2 / ifft2 (numpy.fft conventions) of HOST complex128 slices [nslices][nil][nxl] through the loop's own passes (p3d_f64.hip) */
int p3d_fft2_c128(p3d_plan64* plan, const void* in_host, void* out_host, int nslices, int inverse);
int p3d_pocs64_run(p3d_plan64* plan, const void* x, int dtype, const double* mask, const double* tau, const uint8_t* active,
                   const p3d_pocs_params* params, void* out, int nslices, int32_t* niter_done, double* sums, double* elapsed_ms);

/* Steps 12 / 14 of the workflow: transform every trace of a (nt, ntraces) time-domain cube (ntraces = nil*nxl, the
 * slice-major layout: sample index slowest) to the frequency domain and back, with the conventions of
 *     xrft.fft(da, dim='twt', shift=False, true_phase=True, true_amplitude=True, shape={dim: nfft})
 *                                                                                  cube_apply_FFT.py:240-254
 *     xrft.ifft(da, dim='freq_twt', true_phase=True, true_amplitude=True)          cube_apply_IFFT.py:83-94
 * i.e. F[k] = dt * exp(-2*pi*i*f_k*t0) * sum_n x[n] exp(-2*pi*i*k*n/nfft) = dt * sum_n x[n] exp(-2*pi*i*f_k*(t0 + n*dt)),
 * f_k = fftfreq(nfft, dt)[k], and its exact inverse.  (The xrft fork the reference pins is not available; this is upstream
 * xrft's documented convention.)  For nfft == nt this is what xrft computes whichever way it gets there: its true_phase path
 * rotates the trace by nt/2 samples (ifftshift) and refers the phase to the centre sample t[nt/2], and the two shifts cancel
 * exactly.  For nfft > nt (--upsampling-factor; `shape=` exists only in the fork) the trace is zero-padded at its END here and
 * every sample keeps its physical time t0 + n*dt; an implementation that rotates BEFORE padding would instead place the first
 * half of the trace one record length later (phase exp(-2*pi*i*f_k*nt*dt) on those samples) and report direct_lag = t[nt/2].
 * Which of the two the fork does cannot be established here (parity unpinned); step 14 inverts THIS
 * convention exactly, so 12 -> 13 -> 14 round trips are unaffected, but spectra of an upsampled step 12 should not be mixed
 * with spectra written by the reference's step 12.
 *   p3d_time2freq: x HOST float32 [nt][ntraces] -> out HOST complex64 [nfreq][ntraces]; the trace is zero-padded to
 *       nfft >= nt (--upsampling-factor); real_only != 0 keeps k = 0..nfft/2 (--compute_real, nfreq = nfft/2+1) else
 *       nfreq = nfft; window (HOST float32 [nfreq] or NULL) multiplies every frequency sample (cube_apply_FFT.py:273-278).
 *   p3d_freq2time: X HOST complex64 [nfreq][ntraces] -> out HOST float32 [nfft][ntraces] (real part).  kidx (HOST int32
 *       [nfreq]) gives the FFT bin k of every stored frequency sample (0..nfft-1), so spectra whose filtered samples were
 *       dropped (--drop-filtered-freq) are zero-filled; real_only != 0 completes the Hermitian half.
 * Any nfft up to 10240 (any-length line FFT).  Both allocate their own device buffers on `device`. */
int p3d_time2freq(int device, const float* x, int nt, size_t ntraces, double dt, double t0, int nfft, int real_only,
                  const float* window, void* out);
int p3d_freq2time(int device, const void* X, int nfreq, const int32_t* kidx, size_t ntraces, double dt, double t0, int nfft,
                  int real_only, float* out);
/* the same transforms on DEVICE buffers (x / out of p3d_time2freq, X / out of p3d_freq2time), for callers that keep the cube in HBM
 * across steps 12 -> 13 -> 14; window and kidx stay HOST arrays */
int p3d_time2freq_dev(int device, const float* x_dev, int nt, size_t ntraces, double dt, double t0, int nfft, int real_only,
                      const float* window, void* out_dev);
int p3d_freq2time_dev(int device, const void* X_dev, int nfreq, const int32_t* kidx, size_t ntraces, double dt, double t0, int nfft,
                      int real_only, float* out_dev);

/* Sparse-spectrum statistics of the last p3d_pocs_run[_dev] on this plan: fraction of 8-column blocks of the thresholded spectra
 * that kept at least one coefficient, averaged over slices and iterations (blocks the threshold empties are neither transformed
 * back, stored nor re-read -- exact, since they are zeros); -1 when the dense path ran (shape not eligible, P3D_NO_SPARSE=1,
 * generic lengths). */
int p3d_last_sparsity(p3d_plan* plan, double* nonzero_fraction);

/* After a run with P3D_FLAG_PROFILE: average duration (ms) and launch count of the spectrum
 * (column) pass and of the space (row) pass kernels of the iteration loop. */
int p3d_last_profile(p3d_plan* plan, double* colpass_ms, int* colpass_launches, double* rowpass_ms,
                     int* rowpass_launches);

/* ---- WAVELET variant (transform_kind = 'WAVELET') ---------------------------------------------------------------------------
 * Replaces the pywt.wavedec2 / pywt.waverec2(wavelet, mode='smooth') pair the reference passes to POCS_algorithm
 * (cube_POCS_interpolation_3D.py:260-264; POCS.py:524-525, 585-588, 608-609) and the per-level, per-detail thresholding of
 * threshold_wavelet (POCS.py:105-166).  The caller passes the four filters of the orthogonal / biorthogonal bank (doubles,
 * PyWavelets' dec_lo, dec_hi, rec_lo, rec_hi; 2..64 taps); `level` < 0 selects pywt.dwt_max_level(min(nil, nxl), flen).
 * Coefficients of one slice are a flat complex64 vector: cA, then (cH, cV, cD) of every level, coarsest first (PyWavelets'
 * list order); p3d_wavelet_info reports nlev, the vector length and the (rows, cols) of cA and of each level's details.
 * In p3d_wavelet_stats / p3d_wavelet_run (and p3d_shearlet_stats / p3d_shearlet_run below) the cube pointers `x`, `mask` and `out`
 * may be host OR device pointers (the copy into the plan's staging area is a hipMemcpyDefault): a caller that keeps the cube in
 * HBM pays a device-to-device copy instead of the PCIe transfer.  `elapsed_ms` is the device time of the loop alone. */
typedef struct p3d_wplan p3d_wplan;
int p3d_wavelet_plan_create(p3d_wplan** out, int device, int nil, int nxl, int max_slices, const double* dec_lo,
                            const double* dec_hi, const double* rec_lo, const double* rec_hi, int flen, int level);
int p3d_wavelet_plan_destroy(p3d_wplan* plan);
int p3d_wavelet_info(p3d_wplan* plan, int* nlev, int64_t* ncoef, int32_t* shapes);
/* test hooks: x HOST complex64 [nslices][nil][nxl] <-> coef HOST complex64 [nslices][ncoef] (waverec2 crops to nil x nxl) */
int p3d_wavedec2_c64(p3d_wplan* plan, const void* x, void* coef, int nslices);
int p3d_waverec2_c64(p3d_wplan* plan, const void* coef, void* x, int nslices);
/* statistics for the threshold schedule (POCS.py:253-254, 281): stats HOST double [nslices][nlev][3][4] =
 * (Re, Im of the lexicographic max; max |d|; min |d|) of each detail array, levels coarsest first. dtype as p3d_pocs_run. */
int p3d_wavelet_stats(p3d_wplan* plan, const void* x, int dtype, int nslices, double* stats);
/* the loop (POCS.py:549-632, WAVELET branches).  x/out HOST [nslices][nil][nxl] (dtype), mask HOST float32 [nil][nxl],
 * tau HOST double [nslices][niter][nlev][3][2] (Re, Im), active HOST uint8 [nslices] or NULL, niter_done HOST int32
 * [nslices] or NULL, sums HOST double [niter + 1][nslices] or NULL (sum |x| per iteration; row 0 = the input). */
int p3d_wavelet_run(p3d_wplan* plan, const void* x, int dtype, const float* mask, const double* tau, const uint8_t* active,
                    const p3d_pocs_params* prm, void* out, int nslices, int32_t* niter_done, double* sums,
                    double* elapsed_ms);

/* The WAVELET loop in the REFERENCE's double precision (p3d_wavelet64.hip): pywt.wavedec2 / waverec2 keep float64 for float64 input and
 * POCS_algorithm never narrows (functions/POCS.py:585-588, 596-597, 608-609; threshold_wavelet POCS.py:105-166); the driver casts to the input
 * dtype only at the end (cube_POCS_interpolation_3D.py:324).  Same decomposition, thresholds, schedule layout (tau [nslices][niter][nlev][3][2],
 * stats [nslices][nlev][3][4]) and error behaviour as p3d_wavelet_*; every sample, tap, weight and statistic in double.  dtype of x / out:
 * P3D_C128, P3D_F64, or P3D_C64 / P3D_F32 (converted on load / store); x, out, mask may be host or device pointers; mask is DOUBLE [nil][nxl].
 * A precision path (one thread per output sample and axis, no LDS tiles). */
typedef struct p3d_wplan64 p3d_wplan64;
int p3d_wavelet64_plan_create(p3d_wplan64** out, int device, int nil, int nxl, int max_slices, const double* dec_lo,
                              const double* dec_hi, const double* rec_lo, const double* rec_hi, int filter_len, int level);
int p3d_wavelet64_plan_destroy(p3d_wplan64* plan);
int p3d_wavelet64_info(p3d_wplan64* plan, int* nlev, int64_t* ncoef);
int p3d_wavelet64_stats(p3d_wplan64* plan, const void* x, int dtype, int nslices, double* stats);
int p3d_wavelet64_run(p3d_wplan64* plan, const void* x, int dtype, const double* mask, const double* tau, const uint8_t* active,
                      const p3d_pocs_params* params, void* out, int nslices, int32_t* niter_done, double* sums, double* elapsed_ms);

/* ---- SHEARLET variant (transform_kind = 'SHEARLET') --------------------------------------------------------------------------
 * Replaces FFST.shearletTransformSpect / inverseShearletTransformSpect (cube_POCS_interpolation_3D.py:269-274; POCS.py:526-527,
 * 589-590, 610-611) for spectra Psi supplied by the caller (the reference's `auxiliary_data`): ST_s = ifft2(Psi_s * fft2(x)),
 * x = ifft2(sum_s fft2(ST_s) * Psi_s); per-shearlet thresholds (POCS.py:598 with a (nsh,) tau).  psi: HOST float32
 * [nsh][nil][nxl], FFT order (what fftshift_spectra=True yields), real.  float32 cubes keep real coefficients. */
typedef struct p3d_splan p3d_splan;
int p3d_shearlet_plan_create(p3d_splan** out, int device, int nil, int nxl, int nsh, const float* psi, int max_slices);
int p3d_shearlet_plan_destroy(p3d_splan* plan);
/* How much of the frame the loop has to touch: a shearlet's spectrum vanishes on most rows of the frequency plane (a Parseval frame
 * covers every frequency about twice), and the fused passes skip the 8-row groups on which it does -- exact, those rows carry only
 * zeros through the iteration.  row_group_fraction: share of the (shearlet, 8-row group) pairs that are NOT skipped (1.0: dense
 * path, e.g. P3D_SHEARLET_NO_SUPPORT=1 or the unfused passes).  paired (may be NULL): 1 when float32 cubes take the Hermitian form of
 * the loop -- symmetric spectra (Psi_s(-k) = Psi_s(k), FFST's realCoefficients=True) give real coefficients, the work slices are
 * Hermitian along the rows, so only rows 0 ... nil/2 are computed / stored / read and the column pass sends two columns through
 * one complex transform (results agree with the general form to float32 rounding; P3D_SHEARLET_NO_PAIR=1 switches it off). */
int p3d_shearlet_info(p3d_splan* plan, double* row_group_fraction, int* paired);
/* test hooks: x HOST complex64 [nslices][nil][nxl] <-> st HOST complex64 [nslices][nsh][nil][nxl] */
int p3d_shearlet_transform_c64(p3d_splan* plan, const void* x, void* st, int nslices);
int p3d_shearlet_inverse_c64(p3d_splan* plan, const void* st, void* x, int nslices);
/* statistics for the schedule (POCS.py:257-258, 285, 318): stats HOST double [nslices][nsh][5] = (Re, Im of the lexicographic
 * maximum -- signed maximum for float32 cubes; max |c|; min |c|; sum |c|^2) of each shearlet's coefficients */
int p3d_shearlet_stats(p3d_splan* plan, const void* x, int dtype, int nslices, double* stats);
/* the loop (POCS.py:549-632, SHEARLET branches); arguments as p3d_wavelet_run with tau HOST double [nslices][niter][nsh][2] */
int p3d_shearlet_run(p3d_splan* plan, const void* x, int dtype, const float* mask, const double* tau, const uint8_t* active,
                     const p3d_pocs_params* prm, void* out, int nslices, int32_t* niter_done, double* sums,
                     double* elapsed_ms);
/* The SHEARLET loop in the REFERENCE's double precision (p3d_shearlet64.hip): np.fft.fft2 / ifft2 inside FFST compute in double and hand back
 * complex128 / float64 coefficients, so POCS_algorithm's loop runs in double whatever the cube's dtype; the driver narrows at the end
 * (cube_POCS_interpolation_3D.py:324).  Same frame, thresholds, schedule layout (tau [nslices][niter][nsh][2], stats [nslices][nsh][5]) and error
 * behaviour as p3d_shearlet_*; every sample, spectrum, weight and statistic in double.  psi: HOST DOUBLE [nsh][nil][nxl].  dtype of x / out:
 * P3D_C128, P3D_F64, or P3D_C64 / P3D_F32 (converted on load / store); x, out, mask may be host or device pointers; mask is DOUBLE [nil][nxl].
 * Three fused passes per iteration over the coefficients where both extents have a plan on the double-precision register engine (p3d_mix64.hip), unfused
 * passes on the line transforms of p3d_plan64 otherwise. */
typedef struct p3d_splan64 p3d_splan64;
int p3d_shearlet64_plan_create(p3d_splan64** out, int device, int nil, int nxl, int nsh, const double* psi, int max_slices);
int p3d_shearlet64_plan_destroy(p3d_splan64* plan);
/* fused: 1 when the loop runs its three fused passes on the double-precision register engine (both extents have a plan there; P3D_SHEARLET64_UNFUSED=1 at
 * plan creation switches them off; bit 1 set as well -- 3 -- when REAL cubes take the Hermitian form of those passes: symmetric spectra, even extents, rows
 * 0 ... nil/2 only and two columns per transform, P3D_SHEARLET64_NO_PAIR=1 switches it off), 0 for the unfused passes; row_group_fraction (may be NULL): share of the (shearlet, row group) pairs the fused passes
 * touch -- rows on which a shearlet's spectrum vanishes are skipped, exactly (1.0: none skipped; P3D_SHEARLET64_NO_SUPPORT=1) */
int p3d_shearlet64_info(p3d_splan64* plan, int* fused, double* row_group_fraction);
/* 1 when a plan for (nil, nxl) slices would run the fused passes */
int p3d_shearlet64_fused_shape(int nil, int nxl);
int p3d_shearlet64_stats(p3d_splan64* plan, const void* x, int dtype, int nslices, double* stats);
int p3d_shearlet64_run(p3d_splan64* plan, const void* x, int dtype, const double* mask, const double* tau, const uint8_t* active,
                       const p3d_pocs_params* params, void* out, int nslices, int32_t* niter_done, double* sums, double* elapsed_ms);

/* ---- step-15 slice smoothing (cube_postprocessing_3D.py:88-124 wraps scipy.ndimage.gaussian_filter / median_filter) ----------
 * x/out HOST float32 [nslices][ny][nx]; boundary mode 'reflect'.  gaussian: separable, radius int(truncate * sigma + 0.5);
 * median: size x size window, size in {3, 5, 7}. */
int p3d_smooth_gaussian(int device, const float* x, size_t nslices, int ny, int nx, double sigma, double truncate, float* out);
int p3d_smooth_median(int device, const float* x, size_t nslices, int ny, int nx, int size, float* out);

#ifdef __cplusplus
}
#endif
#endif /* P3D_H */
