// p3d_generic.hpp -- interface of the any-length fallback (p3d_generic.hip) towards the API layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

#include "p3d_fft.hpp"

namespace p3d {

constexpr int GEN_MAX_FACTORS = 32;
constexpr int GEN_MAX_N = 10240;  // two LDS copies of a line: 2 * N * 8 B <= 160 KiB

struct GenPlan {
    int n;
    int nf;                   // number of factors (-1: not representable)
    int f[GEN_MAX_FACTORS];   // n = f[0] * f[1] * ... ; 4s and 2s first, then odd primes
};

GenPlan gen_make_plan(int n);
void gen_build_twiddles(int n, c32* out);  // out[k] = exp(-2*pi*i*k/n), k < n

// one transform pass over all rows (rows = true) or all columns of a row-major [nslices][n1][n2] cube
hipError_t gen_launch_line_fft(const c32* in, c32* out, const c32* tw, const GenPlan& pl, int dir, float scale, int nslices,
                               int n1, int n2, bool rows, const int* done, hipStream_t st);
hipError_t gen_launch_shrink(c32* w, const c32* tau, int niter, int iter, int op, int nslices, size_t per_slice, const int* done,
                             hipStream_t st);
hipError_t gen_launch_update(c32* w, const void* x, int dtype, const float* mask, void* out, double* sums, int mode, int adaptive,
                             int write_out, float alpha, int nslices, size_t per_slice, const int* done, int zero_fill, hipStream_t st);
hipError_t gen_launch_stats(const c32* w, float* partial, int nslices, size_t per_slice, int blocks, hipStream_t st);

// percentile thresholds: exact order statistics of |X| per slice (3-level radix select)
hipError_t gen_launch_pct_hist(const c32* w, size_t per_slice, const unsigned* sel, unsigned* hist, int level, int nslices, hipStream_t st);
hipError_t gen_launch_pct_scan(unsigned* sel, unsigned* hist, int level, int nslices, hipStream_t st);
hipError_t gen_launch_pct_tau(const unsigned* sel_lo, const unsigned* sel_hi, const float* frac, c32* tau, int niter, int iter, int nslices,
                              hipStream_t st);

// time <-> frequency helper kernels
hipError_t gen_launch_t2f_pad(const float* x, c32* work, int nt, int nfft, size_t ntr, hipStream_t st);
hipError_t gen_launch_scale_rows(const c32* work, c32* out, const c32* factor, int nrows, size_t ntr, hipStream_t st);
hipError_t gen_launch_f2t_fill(const c32* X, c32* work, const c32* factor, const int* src, int nfft, size_t ntr, hipStream_t st);
hipError_t gen_launch_real_part(const c32* work, float* out, size_t total, float scale, hipStream_t st);

}  // namespace p3d
