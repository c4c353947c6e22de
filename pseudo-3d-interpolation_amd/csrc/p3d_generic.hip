// p3d_generic.hip -- any-length fallback of the POCS path (slice extents that are not powers of two).
//
// numpy.fft (pocketfft) accepts every length, and so does the reference (cube_POCS_interpolation_3D.py:255-257);
// the tuned kernels of p3d_kernels.hpp cover powers of two only.  This file provides the same pipeline
// for arbitrary (nil, nxl) from simple building blocks: an LDS-resident mixed-radix Stockham line FFT whose
// factor list is computed at run time (prime factors larger than 8 fall back to a direct O(p^2) butterfly), and
// element-wise kernels for thresholding, re-insertion and the reductions.  One iteration is NOT fused here
// (4 transform passes + 2 element-wise passes): correctness and coverage first, the hot sizes have their own path.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <vector>

#include "p3d_generic.hpp"
#include "p3d_shrink.hpp"

namespace p3d {

// radix-R butterfly with compile-time R (registers, no dynamic indexing): v[t] = A[j + t*m] * w^(t*jm), then the
// direct R-point DFT with constants taken from the same table
template <int R>
__device__ inline void small_butterfly(const c32* A, c32* B, const c32* tw, int j, int m, int jm, int j0, int ns, int tstep,
                                       int rstep, int dir)
{
    c32 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        c32 w = tw[t * jm * tstep];
        if (dir > 0) w.y = -w.y;
        v[t] = A[j + t * m] * w;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        c32 acc = v[0];
#pragma unroll
        for (int t = 1; t < R; ++t) {
            c32 w = tw[((t * k) % R) * rstep];
            if (dir > 0) w.y = -w.y;
            const c32 pr = v[t] * w;
            acc = acc + pr;
        }
        B[j0 + k * ns] = acc;
    }
}

// ---- line FFT: one workgroup per line, line resident in LDS (ping-pong) --------------------------------------
// in/out element (line l, index i) at  base_of(l) + i*es,  base_of(l) = (l / lpo)*outer + (l % lpo)*inner
//   rows of a [slice][n1][n2] cube:    es = 1,  lpo = n1,  outer = n1*n2, inner = n2   (n = n2)
//   columns:                            es = n2, lpo = n2,  outer = n1*n2, inner = 1    (n = n1)
// tw[k] = exp(-2*pi*i*k/n) (double precision, rounded once); dir = -1 forward, +1 inverse (conjugated table).
__global__ void gen_line_fft(const c32* in, c32* out, const c32* tw, GenPlan pl, int dir, float scale, int es, int lpo,
                             size_t outer, size_t inner, const int* done, int lines_per_slice)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = pl.n;
    c32* A = reinterpret_cast<c32*>(smem_raw);
    c32* B = A + n;
    const size_t l = blockIdx.x;
    if (done && done[l / lines_per_slice] != 0) return;
    const size_t base = (l / lpo) * outer + (l % lpo) * inner;
    for (int i = threadIdx.x; i < n; i += blockDim.x) A[i] = in[base + (size_t)i * es];
    __syncthreads();
    int ns = 1;
    for (int p = 0; p < pl.nf; ++p) {
        const int R = pl.f[p];
        const int m = n / R;              // butterflies of this pass
        const int tstep = n / (ns * R);   // tw index of exp(-2*pi*i/(ns*R))
        const int rstep = n / R;          // tw index of exp(-2*pi*i/R)
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            const int jm = j % ns;
            const int j0 = (j / ns) * ns * R + jm;
            if (R <= 8) {
                switch (R) {
                    case 2: small_butterfly<2>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                    case 3: small_butterfly<3>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                    case 4: small_butterfly<4>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                    case 5: small_butterfly<5>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                    case 7: small_butterfly<7>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                    default: break;  // the factoriser only emits 2, 4 and odd primes
                }
            } else {  // large prime factor: direct butterfly, inputs re-read from LDS
                for (int k = 0; k < R; ++k) {
                    c32 acc{0.f, 0.f};
                    for (int t = 0; t < R; ++t) {
                        // twiddle of the pass and of the butterfly combined: exp(-+2*pi*i*(t*jm*tstep + ((t*k)%R)*rstep)/n)
                        const long idx = ((long)t * jm * tstep + (long)((long)t * k % R) * rstep) % n;
                        c32 w = tw[idx];
                        if (dir > 0) w.y = -w.y;
                        const c32 pr = A[j + t * m] * w;
                        acc = acc + pr;
                    }
                    B[j0 + k * ns] = acc;
                }
            }
        }
        __syncthreads();
        c32* t = A; A = B; B = t;
        ns *= R;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const c32 v = A[i];
        out[base + (size_t)i * es] = c32{v.x * scale, v.y * scale};
    }
}

// ---- element-wise kernels -----------------------------------------------------------------------------------------
__global__ void gen_shrink_kernel(c32* w, const c32* tau, int niter, int iter, int op, size_t per_slice, const int* done)
{
    const int s = blockIdx.y;
    if (done && done[s] != 0) return;
    const Shrink shr(tau[(size_t)s * niter + iter], op);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x)
        w[(size_t)s * per_slice + i] = shr(w[(size_t)s * per_slice + i]);
}

__device__ inline double block_sum(double v, double* sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    return sh[0];
}

// mode 0: first input  (w = x or its APOCS mix, sums[0] += |x|)
// mode 1: re-insertion (xn = w*(1-alpha*m) + alpha*x; sums += |xn|; optional out = xn; w = xn or its APOCS mix)
__global__ void gen_update_kernel(c32* w, const void* x, int dtype, const float* mask, void* out, double* sums, int mode,
                                  int adaptive, int write_out, float alpha, size_t per_slice, const int* done, int zero_fill)
{
    __shared__ double sh[256];
    const int s = blockIdx.y;
    const int dn = done ? done[s] : 0;
    if (zero_fill) {  // final pass: an empty slice is handed back untouched (zeros)
        if (dn < 0)
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x) {
                if (dtype == 0) reinterpret_cast<c32*>(out)[(size_t)s * per_slice + i] = c32{0.f, 0.f};
                else reinterpret_cast<float*>(out)[(size_t)s * per_slice + i] = 0.f;
            }
    }
    if (dn != 0) return;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = (size_t)s * per_slice + i;
        c32 xo;
        if (dtype == 0) xo = reinterpret_cast<const c32*>(x)[g];
        else xo = c32{reinterpret_cast<const float*>(x)[g], 0.f};
        const float m = mask ? mask[i] : 0.f;
        const float wgt = 1.0f - alpha * m;
        c32 xn;
        if (mode == 0) {
            xn = xo;
        } else {
            xn = axpby(w[g], wgt, xo, alpha);
            if (write_out) {
                if (dtype == 0) reinterpret_cast<c32*>(out)[g] = xn;
                else reinterpret_cast<float*>(out)[g] = xn.x;
            }
        }
        acc += (double)sqrtf(xn.x * xn.x + xn.y * xn.y);
        if (adaptive) {
            const c32 blend = xo * alpha + xn * wgt;
            w[g] = blend + (xo - xn * m) * (1.0f - alpha);
        } else {
            w[g] = xn;
        }
    }
    const double tot = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(sums + s, tot);
}

// per block: lexicographic max, max |X|^2, min |X|^2, sum |X|^2 -> partial[(s*gridDim.x + b)*8 ..]
__global__ void gen_stats_kernel(const c32* w, float* partial, size_t per_slice)
{
    __shared__ float sh[256 * 5];
    const int s = blockIdx.y;
    float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY, sq = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x) {
        const c32 v = w[(size_t)s * per_slice + i];
        const float p = v.x * v.x + v.y * v.y;
        if (v.x > lr || (v.x == lr && v.y > li)) { lr = v.x; li = v.y; }
        mx = fmaxf(mx, p);
        mn = fminf(mn, p);
        sq += p;
    }
    float* me = sh + threadIdx.x * 5;
    me[0] = lr; me[1] = li; me[2] = mx; me[3] = mn; me[4] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < (int)blockDim.x; ++t) {
            const float* o = sh + t * 5;
            if (o[0] > lr || (o[0] == lr && o[1] > li)) { lr = o[0]; li = o[1]; }
            mx = fmaxf(mx, o[2]);
            mn = fminf(mn, o[3]);
            sq += o[4];
        }
        float* p = partial + ((size_t)s * gridDim.x + blockIdx.x) * 8;
        p[0] = lr; p[1] = li; p[2] = sqrtf(mx); p[3] = sqrtf(mn); p[4] = sq;
    }
}

// ---- percentile thresholds (POCS.py:43-58): tau = np.percentile(|X|, perc) per slice ----------------------------------------
// Exact k-th order statistic of the moduli of one slice by a 3-level radix select on the float bit pattern (moduli are
// non-negative, so their bits order like unsigned integers): 11 + 11 + 10 bits, one histogram pass per level.
// sel[s*8 + ..]: [0] rank still to find inside the current prefix, [1] prefix bits found so far, [2] result bits
__global__ void pct_hist_kernel(const c32* w, size_t per_slice, const unsigned* sel, unsigned* hist, int level)
{
    // (a workgroup counts into LDS first: the top bits of |X| fall into a handful of bins, and every thread of the slice used to
    // queue up on those few words of global memory)
    __shared__ unsigned h[2048];
    const int s = blockIdx.y;
    const unsigned prefix = sel[s * 8 + 1];
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    const int bits = level == 2 ? 10 : 11;
    const unsigned pmask = level == 0 ? 0u : (level == 1 ? 0xFFE00000u : 0xFFFFFC00u);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x) {
        const c32 v = w[(size_t)s * per_slice + i];
        const unsigned key = __float_as_uint(sqrtf(v.x * v.x + v.y * v.y));
        if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & ((1u << bits) - 1u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        if (h[i] != 0u) atomicAdd(&hist[(size_t)s * 2048 + i], h[i]);
}
__global__ void pct_scan_kernel(unsigned* sel, unsigned* hist, int level, int nslices)
{
    // one wavefront per slice: lane l sums bins [32 l, 32 l + 32), a shuffle scan finds the lane whose range holds the rank
    const int s = blockIdx.x, lane = threadIdx.x;
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    const int nb = level == 2 ? 1024 : 2048, per = nb / 64;
    unsigned* h = hist + (size_t)s * 2048;
    const unsigned rank = sel[s * 8 + 0];
    unsigned mine = 0;
    for (int i = 0; i < per; ++i) mine += h[lane * per + i];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const unsigned before = incl - mine;
    const bool holds = rank >= before && rank < incl;
    const unsigned long long vote = __ballot(holds);
    // (rank beyond the total -- cannot happen for ranks < per_slice -- falls into the last bin, as the serial scan did)
    const int owner = vote ? __ffsll((long long)vote) - 1 : 63;
    if (lane == owner) {
        unsigned run = before;
        int b = lane * per;
        const int end = b + per;
        for (; b < end; ++b) {
            const unsigned c = h[b];
            if (rank < run + c) break;
            run += c;
        }
        if (b >= nb) b = nb - 1;
        if (b == end) { b = end - 1; run -= h[b]; }
        sel[s * 8 + 0] = rank - run;
        sel[s * 8 + 1] |= (unsigned)b << shift;
    }
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) h[i] = 0;
}
// tau[s] = lo + (hi - lo) * frac   (np.percentile, linear interpolation between the two neighbouring order statistics)
__global__ void pct_tau_kernel(const unsigned* sel_lo, const unsigned* sel_hi, const float* frac, c32* tau, int niter, int iter, int nslices)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    const double lo = __uint_as_float(sel_lo[s * 8 + 1]), hi = __uint_as_float(sel_hi[s * 8 + 1]);
    tau[(size_t)s * niter + iter] = c32{(float)(lo + (hi - lo) * (double)frac[s]), 0.f};
}

hipError_t gen_launch_pct_hist(const c32* w, size_t per_slice, const unsigned* sel, unsigned* hist, int level, int nslices, hipStream_t st)
{
    const size_t want = (per_slice + 4095) / 4096;   // ~16 samples per thread
    const unsigned bx = (unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want));
    pct_hist_kernel<<<dim3(bx, nslices), 256, 0, st>>>(w, per_slice, sel, hist, level);
    return hipGetLastError();
}
hipError_t gen_launch_pct_scan(unsigned* sel, unsigned* hist, int level, int nslices, hipStream_t st)
{
    pct_scan_kernel<<<nslices, 64, 0, st>>>(sel, hist, level, nslices);
    return hipGetLastError();
}
hipError_t gen_launch_pct_tau(const unsigned* sel_lo, const unsigned* sel_hi, const float* frac, c32* tau, int niter, int iter, int nslices,
                              hipStream_t st)
{
    pct_tau_kernel<<<(nslices + 63) / 64, 64, 0, st>>>(sel_lo, sel_hi, frac, tau, niter, iter, nslices);
    return hipGetLastError();
}

// ---- time <-> frequency helpers (cube_apply_FFT.py:240-254, cube_apply_IFFT.py:83-94) ---------------------------------
// xrft.fft(..., shift=False, true_phase=True, true_amplitude=True):
//     F[k] = dt * exp(-2*pi*i*f_k*t0) * sum_n x[n] exp(-2*pi*i*k*n/nfft),   f_k = fftfreq(nfft, dt)[k]
// (the xrft fork the reference pins is not on disk; this is upstream xrft's documented convention).
// pad_kernel: work[k][j] = (x[k][j], 0) for k < nt, 0 for nt <= k < nfft
__global__ void t2f_pad_kernel(const float* x, c32* work, int nt, int nfft, size_t ntr)
{
    const size_t total = (size_t)nfft * ntr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        work[i] = (i / ntr) < (size_t)nt ? c32{x[i], 0.f} : c32{0.f, 0.f};
}
// out[k][j] = work[k][j] * factor[k]  (factor = dt * phase * optional window, complex, per frequency sample)
__global__ void scale_rows_kernel(const c32* work, c32* out, const c32* factor, int nrows, size_t ntr)
{
    const size_t total = (size_t)nrows * ntr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        out[i] = work[i] * factor[i / ntr];
}
// inverse side: work[k][j] = X[src[k]][j] * factor[k] (conjugated when src[k] < 0 -> Hermitian half), 0 when src[k] == INT_MIN
__global__ void f2t_fill_kernel(const c32* X, c32* work, const c32* factor, const int* src, int nfft, size_t ntr)
{
    const size_t total = (size_t)nfft * ntr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t k = i / ntr, j = i - k * ntr;
        const int sk = src[k];
        c32 v{0.f, 0.f};
        if (sk != INT_MIN) {
            const int row = sk < 0 ? -(sk + 1) : sk;
            v = X[(size_t)row * ntr + j];
            if (sk < 0) v.y = -v.y;
            v = v * factor[k];
        }
        work[i] = v;
    }
}
__global__ void real_part_kernel(const c32* work, float* out, size_t total, float scale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) out[i] = work[i].x * scale;
}

hipError_t gen_launch_t2f_pad(const float* x, c32* work, int nt, int nfft, size_t ntr, hipStream_t st)
{
    t2f_pad_kernel<<<4096, 256, 0, st>>>(x, work, nt, nfft, ntr);
    return hipGetLastError();
}
hipError_t gen_launch_scale_rows(const c32* work, c32* out, const c32* factor, int nrows, size_t ntr, hipStream_t st)
{
    scale_rows_kernel<<<4096, 256, 0, st>>>(work, out, factor, nrows, ntr);
    return hipGetLastError();
}
hipError_t gen_launch_f2t_fill(const c32* X, c32* work, const c32* factor, const int* src, int nfft, size_t ntr, hipStream_t st)
{
    f2t_fill_kernel<<<4096, 256, 0, st>>>(X, work, factor, src, nfft, ntr);
    return hipGetLastError();
}
hipError_t gen_launch_real_part(const c32* work, float* out, size_t total, float scale, hipStream_t st)
{
    real_part_kernel<<<4096, 256, 0, st>>>(work, out, total, scale);
    return hipGetLastError();
}

// ---- host side ------------------------------------------------------------------------------------------------------
GenPlan gen_make_plan(int n)
{
    GenPlan pl{};
    pl.n = n;
    pl.nf = 0;
    int r = n;
    // radix 4 first (fewer passes), then 2, then odd primes; anything left is one (possibly large) prime
    while (r % 4 == 0 && pl.nf < GEN_MAX_FACTORS) { pl.f[pl.nf++] = 4; r /= 4; }
    while (r % 2 == 0 && pl.nf < GEN_MAX_FACTORS) { pl.f[pl.nf++] = 2; r /= 2; }
    for (int p = 3; (long)p * p <= r; p += 2)
        while (r % p == 0 && pl.nf < GEN_MAX_FACTORS) { pl.f[pl.nf++] = p; r /= p; }
    if (r > 1 && pl.nf < GEN_MAX_FACTORS) { pl.f[pl.nf++] = r; r = 1; }
    if (r != 1) pl.nf = -1;  // too many factors for the table (cannot happen below 2^31 with 32 slots)
    return pl;
}

void gen_build_twiddles(int n, c32* out)
{
    for (int k = 0; k < n; ++k) {
        const double ang = -6.283185307179586476925286766559 * double(k) / double(n);
        out[k] = c32{float(std::cos(ang)), float(std::sin(ang))};
    }
}

hipError_t gen_launch_line_fft(const c32* in, c32* out, const c32* tw, const GenPlan& pl, int dir, float scale, int nslices,
                               int n1, int n2, bool rows, const int* done, hipStream_t st)
{
    const size_t lds = sizeof(c32) * 2 * (size_t)pl.n;
    if (lds > 160 * 1024) return hipErrorNotSupported;
    hipError_t e = hipSuccess;
    if (lds > 64 * 1024)
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(gen_line_fft), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
    const size_t outer = (size_t)n1 * n2;
    const int lines_per_slice = rows ? n1 : n2;
    const size_t nlines = (size_t)nslices * lines_per_slice;
    int threads = 64;
    while (threads < 256 && threads * 4 < pl.n) threads *= 2;
    if (rows) gen_line_fft<<<dim3((unsigned)nlines), threads, lds, st>>>(in, out, tw, pl, dir, scale, 1, n1, outer, (size_t)n2, done, lines_per_slice);
    else gen_line_fft<<<dim3((unsigned)nlines), threads, lds, st>>>(in, out, tw, pl, dir, scale, n2, n2, outer, (size_t)1, done, lines_per_slice);
    return hipGetLastError();
}

hipError_t gen_launch_shrink(c32* w, const c32* tau, int niter, int iter, int op, int nslices, size_t per_slice, const int* done,
                             hipStream_t st)
{
    const unsigned bx = (unsigned)((per_slice + 255) / 256 < 1024 ? (per_slice + 255) / 256 : 1024);
    gen_shrink_kernel<<<dim3(bx, nslices), 256, 0, st>>>(w, tau, niter, iter, op, per_slice, done);
    return hipGetLastError();
}

hipError_t gen_launch_update(c32* w, const void* x, int dtype, const float* mask, void* out, double* sums, int mode, int adaptive,
                             int write_out, float alpha, int nslices, size_t per_slice, const int* done, int zero_fill, hipStream_t st)
{
    const unsigned bx = (unsigned)((per_slice + 255) / 256 < 256 ? (per_slice + 255) / 256 : 256);
    gen_update_kernel<<<dim3(bx, nslices), 256, 0, st>>>(w, x, dtype, mask, out, sums, mode, adaptive, write_out, alpha, per_slice, done,
                                                       zero_fill);
    return hipGetLastError();
}

hipError_t gen_launch_stats(const c32* w, float* partial, int nslices, size_t per_slice, int blocks, hipStream_t st)
{
    gen_stats_kernel<<<dim3(blocks, nslices), 256, 0, st>>>(w, partial, per_slice);
    return hipGetLastError();
}

}  // namespace p3d
