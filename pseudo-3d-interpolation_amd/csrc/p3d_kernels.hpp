// p3d_kernels.hpp -- the two fused passes of one POCS iteration on a batch of slices.
//
// One iteration of the reference loop (pseudo_3D_interpolation/functions/POCS.py:560-632)
//     X = fft2(x_old); X = threshold(X, tau_k); x = ifft2(X); x *= 1 - alpha*mask; x += alpha*x_obs
// is regrouped so that every slice is read and written exactly twice per iteration:
//
//   spectrum (column) pass  col_kernel<N1,T,COL_ITER>:
//        load a tile of T columns of the row-transformed slice  -> forward column FFT
//        -> threshold (threshold_operator.py:9-112)             -> inverse column FFT -> store
//   space (row) pass        row_kernel<N2,ROW_MID>:
//        load rows -> inverse row FFT, 1/(N1*N2) -> re-insertion of the observed traces
//        (POCS.py:616-619) -> sum|x| for the cost (POCS.py:622) -> [APOCS input mix, POCS.py:574-575]
//        -> forward row FFT of the NEXT iteration -> store
//
// ROW_FIRST starts the chain (x_obs -> forward row FFT), ROW_LAST ends it (stores x instead of
// transforming again).  The work buffer therefore always holds either "rows transformed" (after a
// row pass) or "rows transformed, columns back in space" (after a column pass).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "p3d_fft.hpp"

namespace p3d {

enum RowMode { ROW_FIRST = 0, ROW_MID = 1, ROW_LAST = 2 };
enum ColMode { COL_ITER = 0, COL_STATS = 1, COL_FWD = 2, COL_INV = 3 };

constexpr int ROW_THREADS = 256;
constexpr int STATS_PARTIAL = 8;  // floats per (slice, tile) written by COL_STATS

struct RowArgs {
    const void* x;      // observed cube (c64 or f32), [nslices][n1][N]           (FIRST, MID, LAST*)
    const float* mask;  // [n1][N] or nullptr (LAST with nullptr = plain inverse transform)
    c32* work;          // [nslices][n1][N]
    void* out;          // result cube (c64 or f32)                                (MID if write_out, LAST)
    const c32* tw_fwd;  // twiddle tables of length N (device)
    const c32* tw_inv;
    double* sums;       // row `sum_row` of [(niter+1)][nslices] receives sum|x| per slice, or nullptr
    const int* done;    // per slice: 0 running, >0 finished at that iteration, <0 all-zero slice; or nullptr
    int n1;
    int nslices;
    int sum_row;
    int dtype;          // 0 = c64, 1 = f32 (of x and out)
    int adaptive;       // APOCS input mix
    int write_out;      // MID: also store the iterate to `out` (needed only when eps > 0)
    float alpha;
    float scale;        // 1/(n1*N)
};

struct ColArgs {
    const c32* in;      // [nslices][N][n2]
    c32* out;           // may alias `in`
    const c32* tw_fwd;  // twiddle tables of length N (device)
    const c32* tw_inv;
    const c32* tau;     // [nslices][niter] (COL_ITER)
    const int* done;
    float* partials;    // [nslices][tiles][STATS_PARTIAL] (COL_STATS)
    int n2;
    int nslices;
    int niter;
    int iter;
    int op;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---- thresholding of one coefficient ---------------------------------------------------------
// tau is complex because the reference scales its schedule with numpy's lexicographic complex
// max (POCS.py:288); comparisons and clipping against it are lexicographic as well.
__device__ __forceinline__ c32 shrink(c32 X, c32 tau, int op)
{
    const float m = sqrtf(X.x * X.x + X.y * X.y);
    if (op == 0) {  // hard: where(|X| < tau, 0, X)          threshold_operator.py:110-112
        const bool below = (m < tau.x) || (m == tau.x && 0.0f < tau.y);
        return below ? c32{0.f, 0.f} : X;
    }
    if (m == 0.0f) return c32{0.f, 0.f};  // 1 - tau/0 = -inf -> clipped to 0
    float gr, gi;
    if (op == 1) {  // soft: X * clip(1 - tau/|X|, 0)           threshold_operator.py:36-39
        const float r = 1.0f / m;
        gr = 1.0f - tau.x * r;
        gi = -tau.y * r;
    } else {        // garrote: X * clip(1 - tau^2/|X|^2, 0)    threshold_operator.py:75-78
        const float r = 1.0f / (m * m);
        gr = 1.0f - (tau.x * tau.x - tau.y * tau.y) * r;
        gi = -(2.0f * tau.x * tau.y) * r;
    }
    const bool keep = (gr > 0.0f) || (gr == 0.0f && gi >= 0.0f);  // lexicographic max(g, 0)
    return keep ? X * c32{gr, gi} : c32{0.f, 0.f};
}

// =================================================================================================
// space (row) pass
// =================================================================================================
template <int N, int MODE>
__global__ __launch_bounds__(ROW_THREADS) void row_kernel(const RowArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    constexpr int LB = ROW_THREADS / TPL;  // lines per workgroup
    constexpr int LSTR = LdsRow::stride(N);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* tabF = reinterpret_cast<c32*>(smem_raw);
    c32* tabI = tabF + N;
    c32* data = tabI + N;

    const int tid = threadIdx.x;
    const int line = tid / TPL;
    const int tl = tid - line * TPL;
    const int slice = blockIdx.y;
    const int row = blockIdx.x * LB + line;
    const bool valid = row < a.n1;

    const int dn = a.done ? a.done[slice] : 0;
    if (MODE == ROW_LAST) {
        if (dn > 0) return;  // converged earlier: `out` already holds that iterate
        if (dn < 0) {        // all-zero slice is handed back untouched (POCS.py:515-521)
            if (valid) {
                const size_t off = ((size_t)slice * a.n1 + row) * N;
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    const int e = tl + TPL * q;
                    if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[off + e] = c32{0.f, 0.f};
                    else reinterpret_cast<float*>(a.out)[off + e] = 0.f;
                }
            }
            return;
        }
    } else if (dn != 0) {
        return;
    }

    for (int i = tid; i < N; i += ROW_THREADS) {
        if (MODE != ROW_LAST) tabF[i] = a.tw_fwd[i];
        if (MODE != ROW_FIRST) tabI[i] = a.tw_inv[i];
    }
    __syncthreads();

    const LdsRow lds{data + line * LSTR};
    const size_t off = ((size_t)slice * a.n1 + (valid ? row : 0)) * N;
    const size_t moff = (size_t)(valid ? row : 0) * N;
    c32 v[PPT];
    c32 xo[PPT];

    // observed data (needed by every mode except a plain inverse transform)
    const bool need_obs = (MODE == ROW_FIRST) || (a.mask != nullptr);
    if (need_obs) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int e = tl + TPL * q;
            if (!valid) xo[q] = c32{0.f, 0.f};
            else if (a.dtype == 0) xo[q] = reinterpret_cast<const c32*>(a.x)[off + e];
            else xo[q] = c32{reinterpret_cast<const float*>(a.x)[off + e], 0.f};
        }
    }

    float acc = 0.f;
    if (MODE == ROW_FIRST) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const c32 x = xo[q];
            acc += sqrtf(x.x * x.x + x.y * x.y);
            if (a.adaptive) {
                // x_old = x at the first iteration (POCS.py:549, 574-575)
                const float m = valid ? a.mask[moff + tl + TPL * q] : 0.f;
                const float w = 1.0f - a.alpha * m;
                const c32 blend = x * a.alpha + x * w;
                v[q] = blend + (x - x * m) * (1.0f - a.alpha);
            } else {
                v[q] = x;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = valid ? a.work[off + tl + TPL * q] : c32{0.f, 0.f};
        line_fft<N, INV>(v, lds, tabI, tl);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int e = tl + TPL * q;
            c32 xn = v[q] * a.scale;
            float m = 0.f;
            if (a.mask != nullptr) {
                m = valid ? a.mask[moff + e] : 0.f;
                const float w = 1.0f - a.alpha * m;       // POCS.py:616
                xn = xn * w + xo[q] * a.alpha;            // POCS.py:619
            }
            acc += sqrtf(xn.x * xn.x + xn.y * xn.y);
            if (MODE == ROW_LAST || a.write_out) {
                if (valid) {
                    if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[off + e] = xn;
                    else reinterpret_cast<float*>(a.out)[off + e] = xn.x;  // np.real(), POCS.py:656
                }
            }
            if (MODE == ROW_MID) {
                if (a.adaptive) {  // x_input of the next iteration (POCS.py:574-575)
                    const float w = 1.0f - a.alpha * m;
                    const c32 blend = xo[q] * a.alpha + xn * w;
                    v[q] = blend + (xo[q] - xn * m) * (1.0f - a.alpha);
                } else {
                    v[q] = xn;
                }
            }
        }
    }

    if (a.sums != nullptr) {
        const float ws = wave_sum(valid ? acc : 0.f);
        if ((tid & 63) == 0) atomicAdd(&a.sums[(size_t)a.sum_row * a.nslices + slice], (double)ws);
    }

    if (MODE != ROW_LAST) {
        line_fft<N, FWD>(v, lds, tabF, tl);
        if (valid) {
#pragma unroll
            for (int q = 0; q < PPT; ++q) a.work[off + tl + TPL * q] = v[q];
        }
    }
}

// =================================================================================================
// spectrum (column) pass
// =================================================================================================
__device__ __forceinline__ bool lex_greater(float ar, float ai, float br, float bi)
{
    return (ar > br) || (ar == br && ai > bi);
}

template <int N, int T, int MODE>
__global__ __launch_bounds__(T* Plan<N>::TPL) void col_kernel(const ColArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    constexpr int THREADS = T * TPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* tabF = reinterpret_cast<c32*>(smem_raw);
    c32* tabI = tabF + N;
    c32* data = tabI + N;

    const int tid = threadIdx.x;
    const int tl = tid / T;
    const int c = tid - tl * T;
    const int slice = blockIdx.y;
    const int col = blockIdx.x * T + c;
    const bool valid = col < a.n2;

    if (a.done && a.done[slice] != 0) return;

    for (int i = tid; i < N; i += THREADS) {
        if (MODE != COL_INV) tabF[i] = a.tw_fwd[i];
        if (MODE == COL_ITER || MODE == COL_INV) tabI[i] = a.tw_inv[i];
    }
    __syncthreads();

    const LdsCol<T> lds{data + c};
    const size_t base = (size_t)slice * N * a.n2 + (valid ? col : 0);
    c32 v[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) v[q] = valid ? a.in[base + (size_t)(tl + TPL * q) * a.n2] : c32{0.f, 0.f};

    if (MODE != COL_INV) line_fft<N, FWD>(v, lds, tabF, tl);

    if (MODE == COL_ITER || (MODE == COL_FWD && a.tau != nullptr)) {
        const c32 tau = a.tau[(size_t)slice * a.niter + a.iter];
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = shrink(v[q], tau, a.op);
    }

    if (MODE == COL_STATS) {
        // lexicographic complex max, max|X|, min|X|, sum|X|^2 of this tile (POCS.py:261-262, 288, 299)
        float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY, sq = 0.f;
        if (valid) {
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const float p = v[q].x * v[q].x + v[q].y * v[q].y;
                if (lex_greater(v[q].x, v[q].y, lr, li)) { lr = v[q].x; li = v[q].y; }
                mx = fmaxf(mx, p);
                mn = fminf(mn, p);
                sq += p;
            }
        }
        // workgroups of short lines have fewer than 64 threads: never combine with an inactive lane
        const int lane = tid & 63;
        const int nact = (THREADS - (tid & ~63)) < 64 ? (THREADS - (tid & ~63)) : 64;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
            const float omx = __shfl_down(mx, o, 64), omn = __shfl_down(mn, o, 64), osq = __shfl_down(sq, o, 64);
            if (lane + o < nact) {
                if (lex_greater(orr, oi, lr, li)) { lr = orr; li = oi; }
                mx = fmaxf(mx, omx);
                mn = fminf(mn, omn);
                sq += osq;
            }
        }
        __syncthreads();  // LDS data region is free again
        float* red = reinterpret_cast<float*>(data);
        const int wave = tid >> 6, nw = (THREADS + 63) >> 6;
        if ((tid & 63) == 0) {
            red[wave * 5 + 0] = lr; red[wave * 5 + 1] = li; red[wave * 5 + 2] = mx;
            red[wave * 5 + 3] = mn; red[wave * 5 + 4] = sq;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nw; ++w) {
                if (lex_greater(red[w * 5], red[w * 5 + 1], lr, li)) { lr = red[w * 5]; li = red[w * 5 + 1]; }
                mx = fmaxf(mx, red[w * 5 + 2]);
                mn = fminf(mn, red[w * 5 + 3]);
                sq += red[w * 5 + 4];
            }
            float* p = a.partials + ((size_t)slice * gridDim.x + blockIdx.x) * STATS_PARTIAL;
            p[0] = lr; p[1] = li; p[2] = sqrtf(mx); p[3] = sqrtf(mn); p[4] = sq;
        }
        return;
    }

    if (MODE == COL_ITER || MODE == COL_INV) line_fft<N, INV>(v, lds, tabI, tl);

    if (valid) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) a.out[base + (size_t)(tl + TPL * q) * a.n2] = v[q];
    }
}

// ---- launch helpers, one instantiation set per line length --------------------------------------
template <int N>
constexpr int col_tile()
{
    return N <= 1024 ? 16 : (N == 2048 ? 8 : 2);
}

template <int N>
constexpr size_t row_lds_bytes()
{
    return sizeof(c32) * (2 * N + (ROW_THREADS / Plan<N>::TPL) * LdsRow::stride(N));
}
template <int N>
constexpr size_t col_lds_bytes()
{
    return sizeof(c32) * (2 * N + (size_t)N * col_tile<N>());
}

template <class K>
inline hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int N>
hipError_t launch_row(int mode, const RowArgs& a, hipStream_t st)
{
    constexpr int LB = ROW_THREADS / Plan<N>::TPL;
    const dim3 grid((a.n1 + LB - 1) / LB, a.nslices);
    constexpr size_t lds = row_lds_bytes<N>();
    hipError_t e = hipSuccess;
    switch (mode) {
        case ROW_FIRST:
            if ((e = allow_lds(row_kernel<N, ROW_FIRST>, lds)) != hipSuccess) return e;
            row_kernel<N, ROW_FIRST><<<grid, ROW_THREADS, lds, st>>>(a);
            break;
        case ROW_MID:
            if ((e = allow_lds(row_kernel<N, ROW_MID>, lds)) != hipSuccess) return e;
            row_kernel<N, ROW_MID><<<grid, ROW_THREADS, lds, st>>>(a);
            break;
        case ROW_LAST:
            if ((e = allow_lds(row_kernel<N, ROW_LAST>, lds)) != hipSuccess) return e;
            row_kernel<N, ROW_LAST><<<grid, ROW_THREADS, lds, st>>>(a);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int N>
hipError_t launch_col(int mode, const ColArgs& a, hipStream_t st)
{
    constexpr int T = col_tile<N>();
    constexpr int THREADS = T * Plan<N>::TPL;
    const dim3 grid((a.n2 + T - 1) / T, a.nslices);
    constexpr size_t lds = col_lds_bytes<N>();
    hipError_t e = hipSuccess;
    switch (mode) {
        case COL_ITER:
            if ((e = allow_lds(col_kernel<N, T, COL_ITER>, lds)) != hipSuccess) return e;
            col_kernel<N, T, COL_ITER><<<grid, THREADS, lds, st>>>(a);
            break;
        case COL_STATS:
            if ((e = allow_lds(col_kernel<N, T, COL_STATS>, lds)) != hipSuccess) return e;
            col_kernel<N, T, COL_STATS><<<grid, THREADS, lds, st>>>(a);
            break;
        case COL_FWD:
            if ((e = allow_lds(col_kernel<N, T, COL_FWD>, lds)) != hipSuccess) return e;
            col_kernel<N, T, COL_FWD><<<grid, THREADS, lds, st>>>(a);
            break;
        case COL_INV:
            if ((e = allow_lds(col_kernel<N, T, COL_INV>, lds)) != hipSuccess) return e;
            col_kernel<N, T, COL_INV><<<grid, THREADS, lds, st>>>(a);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// what the API layer sees of one line length
struct LineOps {
    int n;
    int col_tile;
    hipError_t (*row)(int mode, const RowArgs&, hipStream_t);
    hipError_t (*col)(int mode, const ColArgs&, hipStream_t);
    void (*twiddles)(int dir, c32* out);
};

}  // namespace p3d
