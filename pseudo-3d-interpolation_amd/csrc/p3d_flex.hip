// p3d_flex.hip -- the two-pass POCS pipeline for line lengths that are NOT powers of two.
//
// numpy.fft accepts every length and so does the reference (cube_POCS_interpolation_3D.py:255-257); survey grids are rarely
// powers of two.  The register-resident engine of p3d_fft.hpp is specialised per power-of-two length at compile time; this file
// provides the same two kernels -- column pass (forward transform, threshold, inverse transform) and row pass (inverse
// transform, re-insertion, cost sums, forward transform) -- for any length whose lines fit LDS, behind the same LineOps /
// RowArgs / ColArgs interface and on the same column-blocked work buffer, so that every caller in p3d_api.hip (POCS loop,
// statistics, fft2 hooks, early exit, APOCS) runs unchanged and a slice may mix a tuned axis with a flexible one.
//
// Line FFT: mixed-radix Stockham autosort in LDS (ping-pong), factor list computed on the host (odd factors first: the
// strided writes of a pass with small stride then have an odd stride in banks), radix 2 / 4 butterflies hard-wired, radix 3 / 5
// / 7 as direct DFTs with constants from the twiddle table, larger primes as direct O(p^2) butterflies.  The twiddle table
// exp(-2 pi i k / n) sits in LDS.
//   column pass: a workgroup owns T columns (one 64-byte column block for T = 8); element i of column c lives at X[i*T + c],
//                which is exactly the tile's layout in the work buffer: loads and stores are linear copies.
//   row pass:    one wavefront per row (wave-level synchronisation only), LB rows per workgroup.
// Arithmetic is float32 like the tuned path; results agree with it to rounding (different pass structure).
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "p3d_flex.hpp"
#include "p3d_kernels.hpp"

namespace p3d {

namespace {

struct FlexFactors {
    int n, nf;
    int f[GEN_MAX_FACTORS];
};

FlexFactors flex_factors(int n)
{
    const GenPlan g = gen_make_plan(n);
    FlexFactors p{};
    p.n = n;
    p.nf = g.nf;
    int k = 0;
    for (int i = 0; i < g.nf; ++i) if (g.f[i] % 2) p.f[k++] = g.f[i];   // odd first
    for (int i = 0; i < g.nf; ++i) if (g.f[i] % 2 == 0) p.f[k++] = g.f[i];
    return p;
}

constexpr size_t FLEX_LDS_MAX = 150 * 1024;
constexpr size_t FLEX_LDS_TWO = 80 * 1024;   // two workgroups per CU

size_t col_lds(int n, int T) { return sizeof(c32) * ((size_t)2 * T * n + n); }
size_t row_lds(int n, int LB) { return sizeof(c32) * ((size_t)2 * LB * n + n); }

int pick_col_tile(int n)
{
    if (col_lds(n, 8) <= FLEX_LDS_MAX) return 8;   // a whole 64-byte column block, even at one workgroup per CU
    for (int T : {4, 2, 1}) if (col_lds(n, T) <= FLEX_LDS_MAX) return T;
    return 0;
}
int pick_row_lines(int n)
{
    if (row_lds(n, 1) > FLEX_LDS_MAX) return 0;
    const int LB = n >= 768 ? 1 : (n >= 384 ? 2 : (n >= 192 ? 4 : 8));   // keep 256 threads busy in the narrowest pass (n/7 .. n/2 butterflies per line)
    return LB;
}

// ---- butterflies --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ c32 conj_if(c32 w, int dir) { return dir > 0 ? c32{w.x, -w.y} : w; }
__device__ __forceinline__ c32 mul_i(c32 a, int dir) { return dir > 0 ? c32{-a.y, a.x} : c32{a.y, -a.x}; }   // a * (dir * i)

// butterfly j of a line in pass (R, ns): inputs in[t*mstride] * w^(t*jm), outputs out[k*ostride]; wk[q] = exp(-+2 pi i q / R)
template <int R>
__device__ __forceinline__ void flex_bfly(const c32* in, int mstride, c32* out, int ostride, const c32* tw, int twi, const c32 (&wk)[R], int dir)
{
    c32 v[R];
    v[0] = in[0];
#pragma unroll
    for (int t = 1; t < R; ++t) v[t] = in[t * mstride] * conj_if(tw[t * twi], dir);
    if constexpr (R == 2) {
        out[0] = v[0] + v[1];
        out[ostride] = v[0] - v[1];
    } else if constexpr (R == 4) {
        const c32 a = v[0] + v[2], b = v[0] - v[2], s = v[1] + v[3], d = mul_i(v[1] - v[3], dir);
        out[0] = a + s;
        out[ostride] = b + d;
        out[2 * ostride] = a - s;
        out[3 * ostride] = b - d;
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            c32 acc = v[0];
#pragma unroll
            for (int t = 1; t < R; ++t) acc = acc + v[t] * wk[(t * k) % R];
            out[k * ostride] = acc;
        }
    }
}

// one pass of radix R over `lines` lines of n points: line l, element i at X[i*istr + l*lstr]; threads first, first+step, ...
template <int R>
__device__ __forceinline__ void flex_pass(const c32* A, c32* B, const c32* tw, int n, int ns, int dir, int lines, int istr, int lstr, int first, int step)
{
    const int m = n / R, tstep = n / (ns * R), rstep = n / R;
    c32 wk[R];
#pragma unroll
    for (int q = 0; q < R; ++q) wk[q] = conj_if(tw[q * rstep], dir);   // DFT constants: once per pass, not per butterfly
    const int total = m * lines;
    for (int b = first; b < total; b += step) {
        int l, j;
        if (lstr == 1) { j = b / lines; l = b - j * lines; }   // interleaved lines (column tile): line index fastest
        else { l = b / m; j = b - l * m; }                     // separate lines (rows)
        const int jq = j / ns, jm = j - jq * ns;
        const int j0 = jq * ns * R + jm;
        flex_bfly<R>(A + (size_t)j * istr + (size_t)l * lstr, m * istr, B + (size_t)j0 * istr + (size_t)l * lstr, ns * istr, tw, jm * tstep, wk, dir);
    }
}

// All passes of `lines` (= T) interleaved lines in LDS.  first/step: butterfly indices handled by this thread.  Returns the
// buffer that holds the result.  SYNC: 0 = workgroup barrier, 1 = wavefront-level (one wave owns its lines).
template <int SYNC>
__device__ __forceinline__ void flex_sync()
{
    if constexpr (SYNC == 0) __syncthreads();
    else exchange_sync<true>();
}

template <int SYNC>
__device__ c32* flex_fft(c32* A, c32* B, const c32* tw, const FlexFactors& pl, int dir, int lines, int istr, int lstr, int first, int step)
{
    const int n = pl.n;
    int ns = 1;
    for (int p = 0; p < pl.nf; ++p) {
        const int R = pl.f[p];
        switch (R) {
            case 2: flex_pass<2>(A, B, tw, n, ns, dir, lines, istr, lstr, first, step); break;
            case 3: flex_pass<3>(A, B, tw, n, ns, dir, lines, istr, lstr, first, step); break;
            case 4: flex_pass<4>(A, B, tw, n, ns, dir, lines, istr, lstr, first, step); break;
            case 5: flex_pass<5>(A, B, tw, n, ns, dir, lines, istr, lstr, first, step); break;
            case 7: flex_pass<7>(A, B, tw, n, ns, dir, lines, istr, lstr, first, step); break;
            default: {   // large prime factor: direct butterfly, inputs re-read from LDS
                const int m = n / R, tstep = n / (ns * R), rstep = n / R;
                for (int b = first; b < m * lines; b += step) {
                    int l, j;
                    if (lstr == 1) { j = b / lines; l = b - j * lines; }
                    else { l = b / m; j = b - l * m; }
                    const int jq = j / ns, jm = j - jq * ns;
                    const int j0 = jq * ns * R + jm;
                    for (int k = 0; k < R; ++k) {
                        c32 acc{0.f, 0.f};
                        for (int t = 0; t < R; ++t) {
                            const long idx = ((long)t * jm * tstep + (long)((long)t * k % R) * rstep) % n;
                            acc = acc + A[(size_t)(j + t * m) * istr + (size_t)l * lstr] * conj_if(tw[idx], dir);
                        }
                        B[(size_t)(j0 + k * ns) * istr + (size_t)l * lstr] = acc;
                    }
                }
            }
        }
        flex_sync<SYNC>();
        c32* t = A; A = B; B = t;
        ns *= R;
    }
    return A;
}

// ---- column pass ------------------------------------------------------------------------------------------------------------------
constexpr int FLEX_COL_THREADS = 1024;   // 16 waves per CU even where the tile allows only one workgroup per CU

__global__ __launch_bounds__(FLEX_COL_THREADS) void flex_col_kernel(const ColArgs a, const FlexFactors pl, int mode, int tshift)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = pl.n, T = 1 << tshift, tid = threadIdx.x;
    c32* tw = reinterpret_cast<c32*>(smem_raw);
    c32* A = tw + n;
    c32* B = A + (size_t)n * T;
    const int slice = blockIdx.y, col0 = blockIdx.x * T;
    if (a.done && a.done[slice] != 0) return;
    const bool iter = mode == COL_ITER || mode == COL_ITER_SOFT || mode == COL_ITER_GARROTE;

    for (int i = tid; i < n; i += FLEX_COL_THREADS) tw[i] = a.tw[i];
    const c32* const inb = a.in + (size_t)slice * (a.in_std ? (size_t)n * a.n2 : wk_slice_stride(n, a.n2));
    c32* const outb = a.out + (size_t)slice * (a.out_std ? (size_t)n * a.n2 : wk_slice_stride(n, a.n2));
    auto goff = [&](int std_layout, int i, int col) -> size_t {
        return std_layout ? (size_t)i * a.n2 + col : ((size_t)(col >> 3) * n + i) * 8 + (col & 7);
    };
    for (int e = tid; e < (n << tshift); e += FLEX_COL_THREADS) {
        const int c = e & (T - 1), i = e >> tshift, col = col0 + c;
        A[e] = col < a.n2 ? inb[goff(a.in_std, i, col)] : c32{0.f, 0.f};
    }
    __syncthreads();

    c32* X = A;
    c32* Y = B;
    if (mode != COL_INV) {
        X = flex_fft<0>(A, B, tw, pl, FWD, T, T, 1, tid, FLEX_COL_THREADS);
        Y = X == A ? B : A;
    }
    if (iter || (mode == COL_FWD && a.tau != nullptr)) {
        const c32 tau = a.tau[(size_t)slice * a.niter + a.iter];
        const int op = mode == COL_ITER_SOFT ? 1 : (mode == COL_ITER_GARROTE ? 2 : a.op);   // callers pass COL_ITER + a.op
        for (int e = tid; e < (n << tshift); e += FLEX_COL_THREADS) X[e] = shrink(X[e], tau, op);
        __syncthreads();
    }
    if (mode == COL_STATS) {
        // lexicographic complex max, max|X|, min|X|, sum|X|^2 of this tile (POCS.py:261-262, 288, 299)
        float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY, sq = 0.f;
        for (int e = tid; e < (n << tshift); e += FLEX_COL_THREADS) {
            if (col0 + (e & (T - 1)) >= a.n2) continue;
            const c32 v = X[e];
            const float p = v.x * v.x + v.y * v.y;
            if (lex_greater(v.x, v.y, lr, li)) { lr = v.x; li = v.y; }
            mx = fmaxf(mx, p);
            mn = fminf(mn, p);
            sq += p;
        }
        __shared__ float r[(FLEX_COL_THREADS / 64) * 5];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
            const float omx = __shfl_down(mx, o, 64), omn = __shfl_down(mn, o, 64), osq = __shfl_down(sq, o, 64);
            if (lex_greater(orr, oi, lr, li)) { lr = orr; li = oi; }
            mx = fmaxf(mx, omx);
            mn = fminf(mn, omn);
            sq += osq;
        }
        if ((tid & 63) == 0) {
            float* o = r + (tid >> 6) * 5;
            o[0] = lr; o[1] = li; o[2] = mx; o[3] = mn; o[4] = sq;
        }
        __syncthreads();
        if (tid == 0) {
            for (int t = 1; t < FLEX_COL_THREADS / 64; ++t) {
                const float* o = r + t * 5;
                if (lex_greater(o[0], o[1], lr, li)) { lr = o[0]; li = o[1]; }
                mx = fmaxf(mx, o[2]);
                mn = fminf(mn, o[3]);
                sq += o[4];
            }
            float* p = a.partials + ((size_t)slice * gridDim.x + blockIdx.x) * STATS_PARTIAL;
            p[0] = lr; p[1] = li; p[2] = sqrtf(mx); p[3] = sqrtf(mn); p[4] = sq;
        }
        return;
    }
    if (iter || mode == COL_INV) X = flex_fft<0>(X, Y, tw, pl, INV, T, T, 1, tid, FLEX_COL_THREADS);
    for (int e = tid; e < (n << tshift); e += FLEX_COL_THREADS) {
        const int c = e & (T - 1), i = e >> tshift, col = col0 + c;
        if (col < a.n2) outb[goff(a.out_std, i, col)] = X[e];
    }
}

// ---- row pass ------------------------------------------------------------------------------------------------------------------------
// LB adjacent rows per workgroup (256 threads work on all of them together; line l, element i at X[l*n + i]); modes ROW_FIRST /
// ROW_MID / ROW_LAST as in row_kernel (p3d_kernels.hpp)
constexpr int FLEX_ROW_THREADS = 256;

__global__ __launch_bounds__(FLEX_ROW_THREADS) void flex_row_kernel(const RowArgs a, const FlexFactors pl, int mode, int LB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double rsum[FLEX_ROW_THREADS / 64];
    const int n = pl.n, tid = threadIdx.x;
    c32* tw = reinterpret_cast<c32*>(smem_raw);
    c32* A = tw + n;
    c32* B = A + (size_t)LB * n;
    const int slice = blockIdx.y, row0 = blockIdx.x * LB;
    const int nrows = min(LB, a.n1 - row0);   // >= 1

    const int dn = a.done ? a.done[slice] : 0;   // uniform over the workgroup
    const size_t sbase0 = ((size_t)slice * a.n1 + row0) * n;   // row-major cubes (x, out): row l of the group at + l*n
    if (mode == ROW_LAST && a.only_done) {
        if (dn != a.only_done) return;
    } else if (mode == ROW_LAST) {
        if (dn > 0) return;   // converged earlier: `out` already holds that iterate
        if (dn < 0) {         // all-zero slice is handed back untouched (POCS.py:515-521)
            for (int e = tid; e < nrows * n; e += FLEX_ROW_THREADS) {
                if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[sbase0 + e] = c32{0.f, 0.f};
                else reinterpret_cast<float*>(a.out)[sbase0 + e] = 0.f;
            }
            return;
        }
    } else if (dn != 0) {
        return;
    }
    for (int i = tid; i < n; i += FLEX_ROW_THREADS) tw[i] = a.tw[i];
    c32* const wgrp = a.work + (size_t)slice * wk_slice_stride(a.n1, n) + (size_t)row0 * 8;   // row l, col i: + (i>>3)*n1*8 + l*8 + (i&7)
    const size_t wblk = (size_t)a.n1 * 8;
    auto obs_at = [&](size_t e) -> c32 {
        if (a.dtype == 0) return reinterpret_cast<const c32*>(a.x)[sbase0 + e];
        return c32{reinterpret_cast<const float*>(a.x)[sbase0 + e], 0.f};
    };
    const float* const mgrp = a.mask ? a.mask + (size_t)row0 * n : nullptr;
    const int total = LB * n, live = nrows * n;

    c32* X = A;
    if (mode == ROW_FIRST) {
        for (int l = 0; l < LB; ++l) {
            float acc = 0.f;
            for (int i = tid; i < n; i += FLEX_ROW_THREADS) {
                const int e = l * n + i;
                const c32 x = l < nrows ? obs_at(e) : c32{0.f, 0.f};
                acc += sqrtf(x.x * x.x + x.y * x.y);
                if (a.adaptive) {   // x_old = x at the first iteration (POCS.py:549, 574-575)
                    const float m = (mgrp && l < nrows) ? mgrp[e] : 0.f;
                    const float w = 1.0f - a.alpha * m;
                    const c32 blend = x * a.alpha + x * w;
                    A[e] = blend + (x - x * m) * (1.0f - a.alpha);
                } else {
                    A[e] = x;
                }
            }
            if (a.sums != nullptr) {   // one sum per row (uniform control flow: every thread takes part)
                double ws = (double)acc;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ws += __shfl_down(ws, o, 64);
                __syncthreads();
                if ((tid & 63) == 0) rsum[tid >> 6] = ws;
                __syncthreads();
                if (tid == 0 && l < nrows) a.sums[(size_t)slice * a.n1 + row0 + l] = rsum[0] + rsum[1] + rsum[2] + rsum[3];
            }
        }
        __syncthreads();
    } else {
        for (int e = tid; e < total; e += FLEX_ROW_THREADS) {
            const int l = e / n, i = e - l * n;
            A[e] = l < nrows ? wgrp[(size_t)(i >> 3) * wblk + (size_t)l * 8 + (i & 7)] : c32{0.f, 0.f};
        }
        __syncthreads();
        X = flex_fft<0>(A, B, tw, pl, INV, LB, 1, n, tid, FLEX_ROW_THREADS);
        for (int l = 0; l < LB; ++l) {
            float acc = 0.f;
            if (l < nrows) {
                for (int i = tid; i < n; i += FLEX_ROW_THREADS) {
                    const int e = l * n + i;
                    c32 xn = X[e] * a.scale;
                    float m = 0.f;
                    c32 xo{0.f, 0.f};
                    if (!a.plain) xo = obs_at(e);
                    if (mode == ROW_LAST && a.only_done) {
                        // the converged iterate up to one row-transform round trip; an observed trace with alpha = 1 IS the observation
                        if (a.alpha == 1.0f && mgrp && mgrp[e] == 1.0f) xn = xo;
                    } else if (!a.plain) {
                        m = mgrp ? mgrp[e] : 0.f;
                        const float w = 1.0f - a.alpha * m;        // POCS.py:616
                        xn = axpby(xn, w, xo, a.alpha);            // POCS.py:619
                    }
                    acc += sqrtf(xn.x * xn.x + xn.y * xn.y);
                    if (mode == ROW_LAST || a.write_out) {
                        if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[sbase0 + e] = xn;
                        else reinterpret_cast<float*>(a.out)[sbase0 + e] = xn.x;   // np.real(), POCS.py:656
                    }
                    if (mode == ROW_MID) {
                        if (a.adaptive) {   // x_input of the next iteration (POCS.py:574-575)
                            const float w = 1.0f - a.alpha * m;
                            const c32 blend = xo * a.alpha + xn * w;
                            X[e] = blend + (xo - xn * m) * (1.0f - a.alpha);
                        } else {
                            X[e] = xn;
                        }
                    }
                }
            }
            if (a.sums != nullptr) {
                double ws = (double)acc;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ws += __shfl_down(ws, o, 64);
                __syncthreads();
                if ((tid & 63) == 0) rsum[tid >> 6] = ws;
                __syncthreads();
                if (tid == 0 && l < nrows) a.sums[(size_t)slice * a.n1 + row0 + l] = rsum[0] + rsum[1] + rsum[2] + rsum[3];
            }
        }
        __syncthreads();
    }
    (void)live;
    if (mode != ROW_LAST) {
        c32* Y = X == A ? B : A;
        X = flex_fft<0>(X, Y, tw, pl, FWD, LB, 1, n, tid, FLEX_ROW_THREADS);
        for (int e = tid; e < total; e += FLEX_ROW_THREADS) {
            const int l = e / n, i = e - l * n;
            if (l < nrows) wgrp[(size_t)(i >> 3) * wblk + (size_t)l * 8 + (i & 7)] = X[e];
        }
    }
}

hipError_t flex_row(int mode, const RowArgs& a, hipStream_t st)
{
    if (mode != ROW_FIRST && mode != ROW_MID && mode != ROW_LAST) return hipErrorNotSupported;   // no shearlet passes here
    if (a.bits != nullptr) return hipErrorInvalidValue;   // packed masks belong to the tuned row pass
    const int n = a.len, LB = pick_row_lines(n);
    if (LB == 0) return hipErrorNotSupported;
    const FlexFactors pl = flex_factors(n);
    const size_t lds = row_lds(n, LB);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flex_row_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLEX_LDS_MAX);
    if (e != hipSuccess) return e;
    flex_row_kernel<<<dim3((a.n1 + LB - 1) / LB, a.nslices), FLEX_ROW_THREADS, lds, st>>>(a, pl, mode, LB);
    return hipGetLastError();
}

hipError_t flex_col(int mode, const ColArgs& a, hipStream_t st)
{
    if (mode == COL_SHRINK) return hipErrorNotSupported;
    const int n = a.len, T = pick_col_tile(n);
    if (T == 0) return hipErrorNotSupported;
    int tshift = 0;
    while ((1 << tshift) < T) ++tshift;
    const FlexFactors pl = flex_factors(n);
    const size_t lds = col_lds(n, T);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flex_col_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLEX_LDS_MAX);
    if (e != hipSuccess) return e;
    ColArgs b = a;
    b.nzflag = nullptr;   // no sparse-tile skipping on this path
    flex_col_kernel<<<dim3((a.n2 + T - 1) / T, a.nslices), FLEX_COL_THREADS, lds, st>>>(b, pl, mode, tshift);
    return hipGetLastError();
}

hipError_t flex_no_pipe(const RowArgs&, int, hipStream_t) { return hipErrorNotSupported; }

}  // namespace

bool flex_supported(int n) { return n >= 2 && n <= GEN_MAX_N && gen_make_plan(n).nf > 0 && pick_col_tile(n) > 0 && pick_row_lines(n) > 0; }
int flex_col_tile(int n) { return pick_col_tile(n); }

const LineOps* get_flex_ops()
{
    // tpl = 0 marks the flexible implementation: no packed mask words, no persistent row pass, no sparse-tile flags
    static const LineOps ops = {0, 0, 0, 0, &flex_row, &flex_col, &flex_no_pipe, 0, 0, nullptr};
    return &ops;
}

}  // namespace p3d
