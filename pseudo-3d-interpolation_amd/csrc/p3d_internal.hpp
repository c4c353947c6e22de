// p3d_internal.hpp -- what the translation units of libp3d_hip.so share beyond the public C ABI (include/p3d.h).
#pragma once
#include <hip/hip_runtime.h>

#include "p3d.h"
#include "p3d_fft.hpp"

namespace p3d {

// A threshold of the host's float64 schedule as the float32 kernels get it.  The reference compares FLOAT32 moduli with a FLOAT64
// (complex128) threshold in double precision -- np.less(np.absolute(X), tau) with tau a NumPy float64 / complex128 scalar
// (threshold_operator.py:110-112; NEP 50 promotes the comparison, complex operands compare lexicographically).  For a float v and a
// double t:  v < t  <=>  v < the smallest float >= t, and the lexicographic tie "v == Re tau and 0 < Im tau" needs Re tau to BE a
// float.  So the hard operator gets Re tau rounded UP and the sign of Im tau only where the tie can happen; rounding to nearest would
// keep, half of the time, a coefficient whose modulus is the float just below tau.  (Soft / garrote are continuous in tau and the
// percentile operators carry percentages: nearest.)
inline c32 tau_for_device(double re, double im, bool hard)
{
    if (!hard) return c32{(float)re, (float)im};
    float t = (float)re;
    if ((double)t < re) t = __builtin_nextafterf(t, __builtin_inff());
    const bool tie = im > 0.0 && (double)t == re;
    return c32{t, tie ? 1.0f : 0.0f};
}

// thread-local message returned by p3d_last_error() (p3d_api.hip)
void set_last_error(const char* msg);

// the stream all launches of a plan go to
hipStream_t plan_stream(p3d_plan* plan);

// p3d_fft2_c64_dev without the trailing stream synchronisation: batched 2-D FFT of complex64 slices on device buffers,
// numpy.fft conventions, in == out allowed; enqueued on plan_stream(plan)
int fft2_async(p3d_plan* plan, const c32* in, c32* out, int nslices, int inverse);

// ---- fused passes of one SHEARLET iteration on the plan's work buffer (batch entry b*nsh + s = shearlet s of slice b) ------------
// available for the tuned (power-of-two) plans; psi: real spectra [nsh][nil][nxl] (row-major, FFT order); F / out: spectra
// [nb][nil][nxl] row-major.  Together: out[b] = fft2( sum_s Psi_s-weighted analysis -> threshold -> synthesis ) without the
// final inverse transform, i.e. out = sum_s Psi_s * fft2(T_s(ifft2(Psi_s * F))).
bool shearlet_fused_supported(p3d_plan* plan);
// work[b*nsh + s] = inverse row FFT of psi_s * F[b]
int shearlet_spread_inv(p3d_plan* plan, const c32* F, const float* psi, int nb, int nsh, const unsigned* sup = nullptr, int sup_words = 0, bool pair = false);
// per work slice: inverse column FFT, 1/(nil*nxl), real part if real_only, threshold with tau[b][iter][s], forward column FFT
// pair: float32 cubes with symmetric spectra may send two columns through one transform (p3d_col_shear.hpp)
int shearlet_col_shrink(p3d_plan* plan, const c32* tau, int nb, int nsh, int niter, int iter, int op, int real_only, const unsigned* sup = nullptr,
                        int sup_words = 0, bool pair = false);
// out[b] = sum_s psi_s * forward row FFT of work[b*nsh + s]
int shearlet_gather_fwd(p3d_plan* plan, const float* psi, c32* out, int nb, int nsh, const unsigned* sup = nullptr, int sup_words = 0, bool pair = false);
// pair (all three passes alike, float32 cubes with symmetric spectra where shearlet_pair_supported): the work slices are Hermitian
// along the rows -- only rows 0 ... nil/2 are computed, stored and read, the column pass sends two columns through one transform;
// the rows nil/2 + 1 ... of the gather pass's output are then NOT written (the caller mirrors them)
bool shearlet_pair_supported(p3d_plan* plan);
// statistics of the coefficients for the schedule through the paired pass (float32 cubes, after shearlet_spread_inv(..., pair = true)):
// host_stats [nb * nsh][5] = signed maximum, 0, max |c|, min |c|, sum c^2 per (slice, shearlet); synchronises the plan's stream
int shearlet_col_stats_pair(p3d_plan* plan, int nb, int nsh, const unsigned* sup, int sup_words, float* host_stats);
// sup: device bitmap [nsh][sup_words] of the 8-row groups on which a shearlet's spectrum does not vanish (ShearArgs::sup); the three
// passes of one iteration must be given the SAME table (a group one pass skips is never stored for the next to read)

// ---- the double-precision 2-D FFT of p3d_f64.hip, borrowed by the double-precision SHEARLET loop (p3d_shearlet64.hip) --------------------
// a plan without the loop's staging buffers: twiddles and the complex128 work buffer [max_slices][nil][nxl]
int plan64_create_bare(p3d_plan64** out, int device, int nil, int nxl, int max_slices);
hipStream_t plan64_stream(p3d_plan64* plan);
void* plan64_work(p3d_plan64* plan);
// in-place batched fft2 / ifft2 (numpy.fft conventions) of complex128 slices [nslices][nil][nxl] at `buf` (device), enqueued on the plan's
// stream; slice i is skipped where done[i / done_group] != 0 (done: device, may be NULL)
int plan64_fft2(p3d_plan64* plan, void* buf, int nslices, bool inverse, const int* done, int done_group);
// ---- ... and the fused passes of that loop on the register engine (p3d_mix64.hip), for a FULL plan64 of nb slices whose two extents have a plan:
// the plan's work buffer holds the slices' spectra F, U is the caller's coefficient buffer [nb * nsh][nil][nxl] complex128, psi [nsh][nil][nxl] doubles
bool plan64_shear_supported(p3d_plan64* plan);
bool plan64_engine_shape(int nil, int nxl);   // both extents have a plan on the register engine (what plan64_shear_supported will find for such a plan)
double* plan64_mask(p3d_plan64* plan);      // device [nil][nxl]: the caller fills it
void* plan64_stage_x(p3d_plan64* plan);     // staging buffers for host cubes (max_slices slices of complex128)
void* plan64_stage_out(p3d_plan64* plan);
void plan64_bind(p3d_plan64* plan, const void* x, void* out);   // device pointers of the observed cube and the result the row passes read / write
// work = fft2 of the first input (x, or its APOCS mix); sums_row[slice] = sum |x|
int plan64_shear_first(p3d_plan64* plan, int dtype, double* sums_row, int adaptive, double alpha, int nslices, const int* done);
// U[b * nsh + s] = inverse row transform of psi_s x work[b]
int plan64_shear_spread(p3d_plan64* plan, const double* psi, void* U, int nb, int nsh, const int* done, const unsigned char* sup, int rows);
// columns of U (NULL: of the work buffer, nsh = 1): inverse transform, x scale, real part if real_only; mode 0: threshold with tau[b][iter][s] and forward
// transform; mode 1: nothing more (samples)
int plan64_shear_cols(p3d_plan64* plan, void* U, const void* tau, int nb, int nsh, int niter, int iter, int op, int real_only, int mode, double scale, const int* done,
                      const unsigned char* sup, int pair);
// work[b] = sum_s psi_s x forward row transform of U[b * nsh + s]
int plan64_shear_gather(p3d_plan64* plan, const void* U, const double* psi, int nb, int nsh, const int* done, const unsigned char* sup, int rows);
// rows (spread / gather): 0 = all nil rows; nil / 2 + 1 with pair = 1 in the column pass -- REAL cubes on symmetric spectra have real coefficients, U is Hermitian along
// its columns, only rows 0 ... nil / 2 exist and two adjacent columns share one transform (mix64::ShearCol64::pair); the gather pass then leaves rows nil / 2 + 1 ... of
// the work buffer unwritten (the caller mirrors them: the spectrum of a real slice)
// sup (the three passes alike, device [nsh][groups] bytes or NULL): 0 = the spectrum of shearlet s vanishes on the rows of row group g (plan64_shear_row_group rows
// each: the rows of one workgroup of the row passes); such rows of U are never written, read as zeros by the column pass and skipped by the gather pass (exact)
int plan64_shear_row_group(p3d_plan64* plan);
// work (spectra) -> ifft2, re-insertion (POCS.py:616-619), sums_row[slice] = sum |x_new|, result to `out` if write_out; unless last: work = fft2 of the next input
int plan64_shear_back(p3d_plan64* plan, int dtype, double* sums_row, bool last, int adaptive, int write_out, double alpha, int nslices, const int* done, int zero_fill);

}  // namespace p3d
