// p3d_wavelet.hip -- WAVELET variant of the POCS path (transform_kind = 'WAVELET').
//
// The reference hands pywt.wavedec2 / pywt.waverec2(wavelet, mode='smooth') to POCS_algorithm
// (pseudo_3D_interpolation/cube_POCS_interpolation_3D.py:260-264; used at functions/POCS.py:524-525, 585-588, 608-609) and
// thresholds every detail array of every level with its own tau (threshold_wavelet, POCS.py:105-166; schedule POCS.py:279-281,
// 338-339).  PyWavelets is a third-party C extension that is not part of the reference; its published algorithm is restated
// here (and, in NumPy, in oracle/wavelet_oracle.py, which is pinned against PyWavelets 1.1.1 outputs):
//   * single level along an axis: out[o] = sum_j f[j] * xe[2o + 1 - j], o < floor((n + L - 1) / 2), xe = x extended on both
//     sides by straight lines through the edge pairs ('smooth');
//   * inverse: out[m] = sum_k a[k] * rec_lo[m + L - 2 - 2k] + d[k] * rec_hi[m + L - 2 - 2k], m < 2n - L + 2;
//   * multilevel 2-D: level count floor(log2(min(shape) / (L - 1))); when an approximation is one sample longer than the
//     details of the next finer level its last sample is ignored;
//   * complex input = real and imaginary parts transformed independently (the filters are real).
// The loop runs on the tile kernels (dwt2_tile_kernel / idwt2_tile_kernel: one launch per level and direction, a tile + halo in
// LDS, both axes filtered there, thresholds and the re-insertion fused into the stores; filter length as a template parameter
// for db4 / sym4 and db2); the one-thread-per-output-sample per-axis kernels below them are the reference form kept behind
// P3D_WAVELET_UNFUSED=1 and used for filter banks whose tiles do not fit LDS.  Measurements: profiles/r02_wavelet_levels.txt.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "p3d.h"
#include "p3d_fft.hpp"
#include "p3d_internal.hpp"
#include "p3d_shrink.hpp"

using p3d::c32;

namespace {

constexpr int MAXL = 64;  // longest filter (PyWavelets: db38 has 76 taps; those are refused)

struct Filters {
    float dec_lo[MAXL], dec_hi[MAXL], rec_lo[MAXL], rec_hi[MAXL];
    int len;
};

// ---- arithmetic shared by the real (float) and complex (c32) instantiations ------------------------------------------------------
__device__ __forceinline__ c32 cmulf(c32 a, float s) { return c32{a.x * s, a.y * s}; }
__device__ __forceinline__ float cmulf(float a, float s) { return a * s; }
__device__ __forceinline__ void acc_tap(c32& acc, float f, c32 v) { acc.x += f * v.x; acc.y += f * v.y; }
__device__ __forceinline__ void acc_tap(float& acc, float f, float v) { acc += f * v; }
__device__ __forceinline__ float mag(c32 v) { return sqrtf(v.x * v.x + v.y * v.y); }
__device__ __forceinline__ float mag(float v) { return fabsf(v); }
template <typename T> __device__ __forceinline__ T zero_of() { return T{}; }
// straight line through the edge pair (e = edge sample, f = its neighbour), t samples beyond the edge
__device__ __forceinline__ c32 extrapolate(c32 e, c32 f, float t) { return c32{e.x + (e.x - f.x) * t, e.y + (e.y - f.y) * t}; }
__device__ __forceinline__ float extrapolate(float e, float f, float t) { return e + (e - f) * t; }

// Two filter outputs that consume the same samples (low-/high-pass of the analysis, even/odd sample of the synthesis), kept as
// one packed pair so that a tap is a single v_pk_fma_f32: these kernels are VALU-bound, not memory-bound.
typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T> struct Acc;
template <> struct Acc<float> {
    f2 v{0.f, 0.f};
    __device__ __forceinline__ void tap(float g0, float g1, float x) { v = __builtin_elementwise_fma(f2{g0, g1}, f2{x, x}, v); }
    __device__ __forceinline__ float first() const { return v.x; }
    __device__ __forceinline__ float second() const { return v.y; }
};
template <> struct Acc<c32> {
    f2 a{0.f, 0.f}, b{0.f, 0.f};
    __device__ __forceinline__ void tap(float g0, float g1, c32 x)
    {
        const f2 xx{x.x, x.y};
        a = __builtin_elementwise_fma(f2{g0, g0}, xx, a);
        b = __builtin_elementwise_fma(f2{g1, g1}, xx, b);
    }
    __device__ __forceinline__ c32 first() const { return c32{a.x, a.y}; }
    __device__ __forceinline__ c32 second() const { return c32{b.x, b.y}; }
};

// threshold of a real detail coefficient with a real tau, reciprocal instead of division (1 ulp; the operators are continuous
// or, for 'hard', do not divide)
__device__ __forceinline__ float shrink_fast(float x, c32 tau, int op)
{
    const float m = fabsf(x);
    if (op == 0) return m < tau.x ? 0.f : x;
    const float r = __builtin_amdgcn_rcpf(m);
    const float g = op == 1 ? 1.0f - tau.x * r : 1.0f - (tau.x * tau.x) * (r * r);
    return (m > 0.0f && g > 0.0f) ? x * g : 0.f;
}
// one threshold, many coefficients of a tile (what depends on tau alone is computed once: p3d::Shrink)
template <typename T>
struct ShrinkTile;
template <>
struct ShrinkTile<c32> : p3d::Shrink {
    __device__ __forceinline__ ShrinkTile(c32 t, int o) : p3d::Shrink(t, o) {}
};
template <>
struct ShrinkTile<float> {
    c32 tau;
    int op;
    __device__ __forceinline__ ShrinkTile(c32 t, int o) : tau(t), op(o) {}
    __device__ __forceinline__ float operator()(float x) const { return shrink_fast(x, tau, op); }
};

// sample k of a line of n samples at stride `st`, extended by straight lines through the edge pairs ('smooth')
template <typename T>
__device__ __forceinline__ T smooth_at(const T* line, int n, size_t st, int k)
{
    if (k >= 0 && k < n) return line[(size_t)k * st];
    if (n == 1) return line[0];
    if (k < 0) return extrapolate(line[0], line[st], (float)(-k));
    return extrapolate(line[(size_t)(n - 1) * st], line[(size_t)(n - 2) * st], (float)(k - n + 1));
}

// thresholds fused into the last analysis step of a level: z < 0 = leave that output alone (approximation / statistics pass)
struct Thresh {
    const c32* tau;  // [slice][niter][nlev][3]
    int niter, iter, nlev, lvl, op, z_lo, z_hi;
    const int* done;   // early exit: per-slice state, != 0 = leave the slice's arrays alone (its iterate is rebuilt from them after the loop); nullptr: all slices
};

// forward step along one axis of a batch of 2-D arrays.
//   in : [slice][nlines x n] with element (line, k) at line*in_lin + k*in_el (+ slice*in_slice)
//   lo/hi: same addressing with n -> nout (+ slice*lo_slice / hi_slice)
template <typename T>
__global__ void dwt_axis_kernel(const T* in, T* lo, T* hi, Filters f, int nlines, int n, int nout, size_t in_lin, size_t in_el, size_t in_slice,
                                size_t out_lin, size_t out_el, size_t lo_slice, size_t hi_slice, Thresh th)
{
    const int s = blockIdx.y;
    const size_t total = (size_t)nlines * nout;
    const int L = f.len;
    c32 t_lo{0.f, 0.f}, t_hi{0.f, 0.f};
    if (th.z_lo >= 0) t_lo = th.tau[(((size_t)s * th.niter + th.iter) * th.nlev + th.lvl) * 3 + th.z_lo];
    if (th.z_hi >= 0) t_hi = th.tau[(((size_t)s * th.niter + th.iter) * th.nlev + th.lvl) * 3 + th.z_hi];
    const p3d::ShrinkOf<T> sh_lo(t_lo, th.op), sh_hi(t_hi, th.op);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // neighbouring threads walk along the contiguous direction of the data
        int line, o;
        if (in_el == 1) { line = (int)(i / nout); o = (int)(i - (size_t)line * nout); }
        else { o = (int)(i / nlines); line = (int)(i - (size_t)o * nlines); }
        const T* src = in + (size_t)s * in_slice + (size_t)line * in_lin;
        T a = zero_of<T>(), d = zero_of<T>();
        const int top = 2 * o + 1;
        if (top - (L - 1) >= 0 && top < n) {  // interior: no extension
            const T* q = src + (size_t)top * in_el;
            for (int j = 0; j < L; ++j) {
                const T v = q[-(ptrdiff_t)((size_t)j * in_el)];
                acc_tap(a, f.dec_lo[j], v);
                acc_tap(d, f.dec_hi[j], v);
            }
        } else {
            for (int j = 0; j < L; ++j) {
                const T v = smooth_at(src, n, in_el, top - j);
                acc_tap(a, f.dec_lo[j], v);
                acc_tap(d, f.dec_hi[j], v);
            }
        }
        if (th.z_lo >= 0) a = sh_lo(a);
        if (th.z_hi >= 0) d = sh_hi(d);
        const size_t dst = (size_t)line * out_lin + (size_t)o * out_el;
        lo[(size_t)s * lo_slice + dst] = a;
        hi[(size_t)s * hi_slice + dst] = d;
    }
}

// inverse step along one axis: a, d hold n valid samples per line (their buffers may be longer: trimmed approximation)
template <typename T>
__global__ void idwt_axis_kernel(const T* a, const T* d, T* out, Filters f, int nlines, int n, int nout, size_t a_lin, size_t a_el, size_t a_slice,
                                 size_t d_lin, size_t d_el, size_t d_slice, size_t out_lin, size_t out_el, size_t out_slice)
{
    const int s = blockIdx.y;
    const size_t total = (size_t)nlines * nout;
    const int L = f.len;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int line, m;
        if (out_el == 1) { line = (int)(i / nout); m = (int)(i - (size_t)line * nout); }
        else { m = (int)(i / nlines); line = (int)(i - (size_t)m * nlines); }
        const T* pa = a + (size_t)s * a_slice + (size_t)line * a_lin;
        const T* pd = d + (size_t)s * d_slice + (size_t)line * d_lin;
        // taps j = m + L - 2 - 2k in [0, L): k from ceil((m - 1) / 2) = floor(m / 2) to floor((m + L - 2) / 2)
        const int k0 = m / 2;
        int k1 = (m + L - 2) / 2;
        if (k1 > n - 1) k1 = n - 1;
        T acc = zero_of<T>();
        for (int k = k0; k <= k1; ++k) {
            const int j = m + L - 2 - 2 * k;
            acc_tap(acc, f.rec_lo[j], pa[(size_t)k * a_el]);
            acc_tap(acc, f.rec_hi[j], pd[(size_t)k * d_el]);
        }
        out[(size_t)s * out_slice + (size_t)line * out_lin + (size_t)m * out_el] = acc;
    }
}

// ---- fused 2-D steps: one level per launch, a tile per workgroup, both axes through LDS ----------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  All tiles of a slice are
// given to one XCD (slice = 8 * group + xcd) so that the halo re-reads of neighbouring tiles and the partial cache lines they
// write meet in the same L2.  Grid: ntiles * (ns rounded up to a multiple of 8) workgroups, 1-D.
__device__ __forceinline__ bool xcd_decode(int ntiles, int ns, int& slice, int& tile)
{
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    slice = (j / ntiles) * 8 + xcd;
    tile = j % ntiles;
    return slice < ns;
}

// value of the 'smooth'-extended level array at (r, c), any integers: extension along axis 1 of the extension along axis 0.
// Branch-free (clamped addresses, zero weights inside) so that a tile's loads can all be in flight together.
template <typename T>
__device__ __forceinline__ T smooth2_at(const T* a, size_t ld, int H, int W, int r, int c)
{
    const int rc = min(max(r, 0), H - 1), cc = min(max(c, 0), W - 1);
    const int rn = r < 0 ? min(1, H - 1) : max(H - 2, 0);   // inward neighbour of the edge sample (weight 0 when r is inside)
    const int cn = c < 0 ? min(1, W - 1) : max(W - 2, 0);
    const float tr = r < 0 ? (float)(-r) : (r >= H ? (float)(r - H + 1) : 0.f);
    const float tc = c < 0 ? (float)(-c) : (c >= W ? (float)(c - W + 1) : 0.f);
    const T x00 = a[(size_t)rc * ld + cc], x10 = a[(size_t)rn * ld + cc], x01 = a[(size_t)rc * ld + cn], x11 = a[(size_t)rn * ld + cn];
    return extrapolate(extrapolate(x00, x10, tr), extrapolate(x01, x11, tr), tc);
}

// analysis of one level: in (H x W) -> cA, cH, cV, cD (Ho x Wo each).  Workgroup (256 threads as TILE lanes x 256/TILE rows) =
// TILE x TILE coefficients of every subband.  LDS traffic is what bounds these kernels, so axis 1 reads sample pairs (8 B per
// lane, conflict-free) and axis 0 slides a window down a column, every loaded sample feeding all outputs it belongs to.
// LT: the filter length at compile time (0: read from `f`).  With it the tile geometry is constant, the tap loops unroll and
// their coefficients stay in registers -- these kernels are bound by instruction issue, not by memory.
// Threads of a tile kernel's workgroup: a tile's LDS image decides how many workgroups share a CU (four / two for float / complex samples), the
// workgroup size how many wavefronts then cover each other between its barriers (see wfuse1_kernel).  512 instead of 256 threads: configs[3]
// 0.535 -> 0.520 ms per iteration, a complex64 cube 0.688 -> 0.648 (1024: no further gain).  32-point tiles only; the 16-point tiles of the
// coarse levels stay at 256.
#ifndef P3D_WTILE_THREADS
#define P3D_WTILE_THREADS 512
#endif
#ifndef P3D_WTILE_THREADS_COMPLEX
#define P3D_WTILE_THREADS_COMPLEX 512
#endif
template <typename T, int TILE> constexpr int WTILE_NT = TILE != 32 ? 256 : (sizeof(T) == sizeof(float) ? P3D_WTILE_THREADS : P3D_WTILE_THREADS_COMPLEX);
template <typename T, int TILE, int LT>
__global__ __launch_bounds__((WTILE_NT<T, TILE>)) void dwt2_tile_kernel(const T* in, size_t in_slice, int H, int W, T* cA, size_t cA_slice, T* det, size_t det_slice, int Ho, int Wo,
                                                        Filters f, int tiles_x, int ntiles, int ns, Thresh th)
{
    extern __shared__ __align__(16) unsigned char w_smem[];
    constexpr int NT = WTILE_NT<T, TILE>;
    constexpr int LX = TILE, LY = NT / TILE, R = TILE / LY > 0 ? TILE / LY : 1;
    const int L = LT ? LT : f.len, IH = 2 * TILE + L - 2, IW = IH;  // L is even: IW is even, rows of s_in start 8-byte aligned for float
    T* s_in = reinterpret_cast<T*>(w_smem);
    T* s_lo = s_in + (size_t)IH * IW;
    T* s_hi = s_lo + (size_t)IH * TILE;
    const int tx = threadIdx.x % LX, ty = threadIdx.x / LX;
    int s, tile;
    if (!xcd_decode(ntiles, ns, s, tile)) return;
    if (th.done != nullptr && th.done[s] != 0) return;   // (uniform over the workgroup)
    const int by = tile / tiles_x, bx = tile - by * tiles_x;
    const int or0 = by * TILE, oc0 = bx * TILE, r0 = 2 * or0 - L + 2, c0 = 2 * oc0 - L + 2;
    // the last tile of an axis holds Ho mod TILE (Wo mod TILE) outputs -- 3 of 32 for a 512-point axis and db4 -- and needs
    // that much of the input only: vh x vw valid outputs from IHv x IWv samples (the LDS pitch stays IW)
    const int vh = min(TILE, Ho - or0), vw = min(TILE, Wo - oc0), IHv = 2 * vh + L - 2, IWv = 2 * vw + L - 2;
    const T* src = in + (size_t)s * in_slice;
    const bool inside = r0 >= 0 && c0 >= 0 && r0 + IHv <= H && c0 + IWv <= W;
    __shared__ float4 s_tap[MAXL / 2];
    if ((int)threadIdx.x < L / 2) {
        const int j = 2 * threadIdx.x;
        s_tap[threadIdx.x] = float4{f.dec_lo[j], f.dec_hi[j], f.dec_lo[j + 1], f.dec_hi[j + 1]};
    }
    // loads in batches of KR rows x MC column steps per thread: all of a batch are issued before the first LDS write waits.
    // 32-bit element offsets from the slice base (a level of one slice has < 2^31 samples) keep the address arithmetic short --
    // it, not the filtering, is most of the VALU work of this kernel.
    constexpr int KR = TILE == 32 ? 9 : 6, MC = (2 * TILE + MAXL - 2 + LX - 1) / LX;   // db4 .. coif2 at TILE 32: one batch
    if (LT != 0 && inside && vh == TILE && vw == TILE) {
        // a whole interior tile with the filter length known: its IH x IW samples as ONE index range over the 256 threads
        // (20 loads per thread for db4 where the row / column-step batches below issue 27), LDS index = that index
        constexpr int IHc = 2 * TILE + LT - 2, IWc = IHc, NE = (IHc * IWc + NT - 1) / NT;
        const T* g = src + (size_t)r0 * W + c0;
        T v[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = (int)threadIdx.x + k * NT, r = e / IWc, c = e - r * IWc;
            v[k] = zero_of<T>();
            if (e < IHc * IWc) v[k] = g[(unsigned)r * (unsigned)W + (unsigned)c];
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = (int)threadIdx.x + k * NT;
            if (e < IHc * IWc) s_in[e] = v[k];
        }
    } else if (inside) {
        const T* g = src + (size_t)r0 * W + c0;
        for (int rb = ty; rb < IHv; rb += KR * LY) {
            T v[KR][MC];
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const unsigned ro = (unsigned)(rb + k * LY) * (unsigned)W + tx;
#pragma unroll
                for (int m = 0; m < MC; ++m) {
                    v[k][m] = zero_of<T>();
                    if (rb + k * LY < IHv && tx + m * LX < IWv) v[k][m] = g[ro + m * LX];
                }
            }
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                T* d = s_in + (rb + k * LY) * IW + tx;
#pragma unroll
                for (int m = 0; m < MC; ++m)
                    if (rb + k * LY < IHv && tx + m * LX < IWv) d[m * LX] = v[k][m];
            }
        }
    } else {
        // boundary tile: 'smooth' extension = straight lines through the edge pairs, along axis 0 then along axis 1; clamped
        // addresses and zero weights inside keep it branch-free (smooth2_at spelled out per row / per column)
        unsigned cc[MC], cn[MC];
        float tc[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            const int c = c0 + tx + m * LX;
            cc[m] = (unsigned)min(max(c, 0), W - 1);
            cn[m] = (unsigned)(c < 0 ? min(1, W - 1) : max(W - 2, 0));
            tc[m] = c < 0 ? (float)(-c) : (c >= W ? (float)(c - W + 1) : 0.f);
        }
        // Only samples beyond an edge need their neighbours: x10 where the row is outside (uniform over a wavefront's two rows but
        // for one wavefront at most), x01 where the column is, x11 where both are.  A sample that is not loaded is 0 and meets the
        // weight 0: e + (e - 0) * 0 = e.
        for (int rb = ty; rb < IHv; rb += KR * LY) {
            T v[KR][MC];
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int r = r0 + rb + k * LY;
                const unsigned rc = (unsigned)min(max(r, 0), H - 1) * (unsigned)W, rn = (unsigned)(r < 0 ? min(1, H - 1) : max(H - 2, 0)) * (unsigned)W;
                const float tr = r < 0 ? (float)(-r) : (r >= H ? (float)(r - H + 1) : 0.f);
#pragma unroll
                for (int m = 0; m < MC; ++m) {
                    v[k][m] = zero_of<T>();
                    if (rb + k * LY < IHv && tx + m * LX < IWv) {
                        T x10 = zero_of<T>(), x01 = zero_of<T>(), x11 = zero_of<T>();
                        const T x00 = src[rc + cc[m]];
                        if (tr != 0.f) x10 = src[rn + cc[m]];
                        if (tc[m] != 0.f) {
                            x01 = src[rc + cn[m]];
                            if (tr != 0.f) x11 = src[rn + cn[m]];
                        }
                        v[k][m] = extrapolate(extrapolate(x00, x10, tr), extrapolate(x01, x11, tr), tc[m]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                T* d = s_in + (rb + k * LY) * IW + tx;
#pragma unroll
                for (int m = 0; m < MC; ++m)
                    if (rb + k * LY < IHv && tx + m * LX < IWv) d[m * LX] = v[k][m];
            }
        }
    }
    __syncthreads();
    // The taps come from LDS (one broadcast read serves both filters and two taps): scalar loads inside these loops would share
    // the wait counter with the LDS reads and serialise every step.
    // axis 1: row r, output tx: sum_j f[j] * row[2 tx + L - 1 - j], two taps = one aligned pair of samples at a time
    struct alignas(2 * sizeof(T)) Pair { T lo, hi; };
    for (int r = ty; r < IHv; r += LY) {
        if (tx >= vw) continue;
        const Pair* q = reinterpret_cast<const Pair*>(s_in + r * IW + 2 * tx + L - 2);
        Acc<T> ad;
        for (int jj = 0; jj < L / 2; ++jj) {
            const float4 g = s_tap[jj];       // dec_lo[2jj], dec_hi[2jj], dec_lo[2jj+1], dec_hi[2jj+1]
            const Pair v = q[-jj];            // samples 2tx + L-2 - 2jj (tap 2jj+1) and the next one (tap 2jj)
            ad.tap(g.x, g.y, v.hi);
            ad.tap(g.z, g.w, v.lo);
        }
        s_lo[r * TILE + tx] = ad.first();
        s_hi[r * TILE + tx] = ad.second();
    }
    __syncthreads();
    if (tx >= vw || ty * R >= vh) return;   // (no barrier below)
    // axis 0: column tx, outputs o = ty*R + q: sum_j f[j] * col[2o + L - 1 - j]
    Acc<T> fl[R], fh[R];   // (aa, da) from the low-pass rows, (ad, dd) from the high-pass rows
    const T* cl = s_lo + (2 * ty * R + L - 1) * TILE + tx;
    const T* ch = s_hi + (2 * ty * R + L - 1) * TILE + tx;
    for (int jj = 0; jj < L / 2; ++jj) {
        const float4 g = s_tap[jj];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int o = (2 * q - 2 * jj) * TILE;
            fl[q].tap(g.x, g.y, cl[o]);
            fh[q].tap(g.x, g.y, ch[o]);
            fl[q].tap(g.z, g.w, cl[o - TILE]);
            fh[q].tap(g.z, g.w, ch[o - TILE]);
        }
    }
    c32 t0{0.f, 0.f}, t1{0.f, 0.f}, t2{0.f, 0.f};
    if (th.tau) {
        const c32* t = th.tau + (((size_t)s * th.niter + th.iter) * th.nlev + th.lvl) * 3;
        t0 = t[0]; t1 = t[1]; t2 = t[2];
    }
    const ShrinkTile<T> sh0(t0, th.op), sh1(t1, th.op), sh2(t2, th.op);
    const size_t cnt = (size_t)Ho * Wo;
    const int gc = oc0 + tx;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int go = or0 + ty * R + q;
        if (go >= Ho || gc >= Wo) continue;
        T da = fl[q].second(), ad = fh[q].first(), dd = fh[q].second();
        if (th.tau) {   // threshold_wavelet: details only
            da = sh0(da);
            ad = sh1(ad);
            dd = sh2(dd);
        }
        const size_t o = (size_t)go * Wo + gc;
        cA[(size_t)s * cA_slice + o] = fl[q].first();
        T* dp = det + (size_t)s * det_slice + o;
        dp[0] = da;
        dp[cnt] = ad;
        dp[2 * cnt] = dd;
    }
}

// ---- the coarse levels of a slice in ONE workgroup -------------------------------------------------------------------------------
// From the first level LC whose input approximation fits LDS (512 x 512, db4, float32: level 3, a 133 x 133 approximation) the levels
// are a few ten thousand samples per slice: as tile kernels they were eight launches of latency (profiles/r02_wavelet_levels.txt:
// 108 of 660 us per iteration).  Here one workgroup carries a slice through analysis LC .. nlev (details thresholded and stored to
// the coefficient vector as the tile kernels store them) and straight back through synthesis nlev .. LC; approximations never leave
// LDS, and rec[LC - 1] comes out as the input of the level-(LC-1) tile synthesis.  The arithmetic of every output sample is the
// tile kernels' (same taps in the same order, the 'smooth' extension evaluated by the same formula), so the two paths agree bit
// for bit (tests/test_gpu_wavelet.py); P3D_WAVELET_NO_COARSE=1 keeps the tile kernels for all levels.
// LDS: X = the level's input (h[LC-1] x w[LC-1] samples, later the approximations / reconstructions, all smaller), Y = the
// row-filtered rows of the analysis incl. their extension rows (2 x (2 h[LC] + L - 2) x w[LC]; the synthesis needs less).
constexpr int MAXLEV = 16;
constexpr int COARSE_THREADS = 1024;
struct CoarseArgs {
    const void* in;        // approx[LC-1] (T) [slice][h[LC-1] * w[LC-1]]                        (do_fwd)
    size_t in_slice;
    void* coef;            // coefficient vectors (T) [slice][ncoef]
    size_t coef_slice;
    void* rec;             // rec[LC-1] (T) [slice][rh * rw]                                       (do_inv)
    size_t rec_slice;
    size_t doff[MAXLEV + 1];
    int h[MAXLEV + 1], w[MAXLEV + 1];
    int nlev, LC, do_fwd, do_inv, x_elems, ns;
};

template <typename T, int LT>
__global__ __launch_bounds__(COARSE_THREADS) void wcoarse_kernel(CoarseArgs a, Filters f, Thresh th)
{
    extern __shared__ __align__(16) unsigned char w_smem[];
    __shared__ float4 s_dec[MAXL / 2], s_rec[MAXL / 2];
    const int L = LT ? LT : f.len, HL = L / 2;
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= a.ns) return;
    if (th.done != nullptr && th.done[s] != 0) return;
    T* X = reinterpret_cast<T*>(w_smem);
    T* Y = X + a.x_elems;
    if (tid < HL) {
        const int j = 2 * tid, k = L - 2 - 2 * tid;
        s_dec[tid] = float4{f.dec_lo[j], f.dec_hi[j], f.dec_lo[j + 1], f.dec_hi[j + 1]};
        s_rec[tid] = float4{f.rec_lo[k], f.rec_hi[k], f.rec_lo[k + 1], f.rec_hi[k + 1]};
    }
    T* coef = reinterpret_cast<T*>(a.coef) + (size_t)s * a.coef_slice;
    if (a.do_fwd) {
        const T* src = reinterpret_cast<const T*>(a.in) + (size_t)s * a.in_slice;
        const int n = a.h[a.LC - 1] * a.w[a.LC - 1];
        for (int i = tid; i < n; i += COARSE_THREADS) X[i] = src[i];
    }
    __syncthreads();
    if (a.do_fwd) {
        for (int l = a.LC; l <= a.nlev; ++l) {
            const int H = a.h[l - 1], W = a.w[l - 1], Ho = a.h[l], Wo = a.w[l], rows = 2 * Ho + L - 2;
            T* ylo = Y;
            T* yhi = Y + rows * Wo;
            // axis 1 of the 'smooth'-extended input (smooth2_at: extension along axis 0, then along axis 1), rows -(L-2) .. 2 Ho - 1
            for (int i = tid; i < rows * Wo; i += COARSE_THREADS) {
                const int e = i / Wo, o = i - e * Wo, r = e - (L - 2), top = 2 * o + 1;
                Acc<T> ad;
                if (r >= 0 && r < H && top - (L - 1) >= 0 && top < W) {
                    const T* q = X + r * W + top;
                    for (int jj = 0; jj < HL; ++jj) {
                        const float4 g = s_dec[jj];
                        ad.tap(g.x, g.y, q[-2 * jj]);
                        ad.tap(g.z, g.w, q[-2 * jj - 1]);
                    }
                } else {
                    for (int jj = 0; jj < HL; ++jj) {
                        const float4 g = s_dec[jj];
                        ad.tap(g.x, g.y, smooth2_at(X, (size_t)W, H, W, r, top - 2 * jj));
                        ad.tap(g.z, g.w, smooth2_at(X, (size_t)W, H, W, r, top - 2 * jj - 1));
                    }
                }
                ylo[i] = ad.first();
                yhi[i] = ad.second();
            }
            __syncthreads();
            // axis 0: cA stays in LDS (X, pitch Wo) for the next level, the details are thresholded and stored
            c32 t0{0.f, 0.f}, t1{0.f, 0.f}, t2{0.f, 0.f};
            if (th.tau) {
                const c32* t = th.tau + (((size_t)s * th.niter + th.iter) * th.nlev + (a.nlev - l)) * 3;
                t0 = t[0]; t1 = t[1]; t2 = t[2];
            }
            const ShrinkTile<T> sh0(t0, th.op), sh1(t1, th.op), sh2(t2, th.op);
            const size_t cnt = (size_t)Ho * Wo;
            T* det = coef + a.doff[l];
            for (int i = tid; i < Ho * Wo; i += COARSE_THREADS) {
                const int p = i / Wo, o = i - p * Wo;
                const T* cl = ylo + (2 * p + L - 1) * Wo + o;
                const T* ch = yhi + (2 * p + L - 1) * Wo + o;
                Acc<T> fl, fh;
                for (int jj = 0; jj < HL; ++jj) {
                    const float4 g = s_dec[jj];
                    fl.tap(g.x, g.y, cl[-2 * jj * Wo]);
                    fh.tap(g.x, g.y, ch[-2 * jj * Wo]);
                    fl.tap(g.z, g.w, cl[-(2 * jj + 1) * Wo]);
                    fh.tap(g.z, g.w, ch[-(2 * jj + 1) * Wo]);
                }
                T da = fl.second(), ad = fh.first(), dd = fh.second();
                if (th.tau) { da = sh0(da); ad = sh1(ad); dd = sh2(dd); }
                X[i] = fl.first();
                if (l == a.nlev) coef[i] = fl.first();
                det[i] = da;
                det[cnt + i] = ad;
                det[2 * cnt + i] = dd;
            }
            __syncthreads();   // (also: the details just stored are read back by the synthesis below)
        }
    } else if (a.do_inv) {
        const int n = a.h[a.nlev] * a.w[a.nlev];
        for (int i = tid; i < n; i += COARSE_THREADS) X[i] = coef[i];
        __syncthreads();
    }
    if (!a.do_inv) return;
    for (int l = a.nlev; l >= a.LC; --l) {
        const int Ho = a.h[l], Wo = a.w[l], RH = 2 * Ho - L + 2, RW = 2 * Wo - L + 2;
        // the approximation: cA (pitch Wo) at the coarsest level, else the reconstruction of level l (pitch 2 w[l+1] - L + 2,
        // possibly one row / column larger than Ho x Wo: the extra samples are ignored, as in pywt.waverec2)
        const int a_ld = l == a.nlev ? Wo : 2 * a.w[l + 1] - L + 2;
        const size_t cnt = (size_t)Ho * Wo;
        const T* det = coef + a.doff[l];
        T* ylo = Y;
        T* yhi = Y + RH * Wo;
        // undo axis 0: rows 2i and 2i + 1 share the coefficients k = i .. i + L/2 - 1
        for (int i = tid; i < (RH / 2) * Wo; i += COARSE_THREADS) {
            const int ip = i / Wo, kc = i - ip * Wo;
            Acc<T> lo, hi;
            for (int t = 0; t < HL; ++t) {
                const float4 g = s_rec[t];
                const int k = ip + t;
                const size_t o = (size_t)k * Wo + kc;
                lo.tap(g.x, g.z, X[k * a_ld + kc]);
                lo.tap(g.y, g.w, det[o]);
                hi.tap(g.x, g.z, det[cnt + o]);
                hi.tap(g.y, g.w, det[2 * cnt + o]);
            }
            ylo[(2 * ip) * Wo + kc] = lo.first();
            ylo[(2 * ip + 1) * Wo + kc] = lo.second();
            yhi[(2 * ip) * Wo + kc] = hi.first();
            yhi[(2 * ip + 1) * Wo + kc] = hi.second();
        }
        __syncthreads();
        // undo axis 1: outputs 2i and 2i + 1 of row m share the coefficients i .. i + L/2 - 1
        T* out = l > a.LC ? X : reinterpret_cast<T*>(a.rec) + (size_t)s * a.rec_slice;
        for (int i = tid; i < RH * (RW / 2); i += COARSE_THREADS) {
            const int m = i / (RW / 2), ii = i - m * (RW / 2);
            const T* ql = ylo + m * Wo + ii;
            const T* qh = yhi + m * Wo + ii;
            Acc<T> eo;
            for (int t = 0; t < HL; ++t) {
                const float4 g = s_rec[t];
                eo.tap(g.x, g.z, ql[t]);
                eo.tap(g.y, g.w, qh[t]);
            }
            out[m * RW + 2 * ii] = eo.first();
            out[m * RW + 2 * ii + 1] = eo.second();
        }
        __syncthreads();
    }
}

// mask [n1][n2] (float weights) -> one bit per sample, rows of `words` 32-bit words; *binary is cleared when a weight is neither 0 nor 1 (the bits are not
// used then).  One thread per word.
__global__ void wmask_pack_kernel(const float* mask, unsigned* bits, int n1, int n2, int words, int* binary)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1 * words) return;
    const int r = i / words, w = i - r * words;
    unsigned v = 0;
    bool ok = true;
    for (int b = 0; b < 32; ++b) {
        const int c = w * 32 + b;
        if (c >= n2) break;
        const float m = mask[(size_t)r * n2 + c];
        ok = ok && (m == 0.f || m == 1.f);
        v |= (m != 0.f ? 1u : 0u) << b;
    }
    bits[i] = v;
    if (!ok) *binary = 0;
}

// what the last synthesis step (level 1) does with its output instead of storing it: crop + re-insertion (POCS.py:609, 616-619)
struct Update {
    int enabled;
    void* feed;         // T*
    const void* x;      // observed slices (dtype)
    int dtype;
    const float* mask;
    const unsigned* mask_bits;   // the same as one bit per sample, rows of mask_words words -- NULL unless every weight is 0 or 1 (wfuse1_kernel)
    int mask_words;
    void* out;
    double* sums;       // [nslices] of this iteration
    int adaptive, write_out, zero_fill;
    float alpha;
    int n1, n2;
    const int* done;
    // "finalize" launch of the early exit (after the loop): only the slices that converged before the last iteration, done in 1 ... only_below - 1, and of
    // those the ones whose level-1 details sit in the buffer at hand: only_parity - 1 = (done - 1) & 1 (3: either).  Their iterate is rebuilt from the
    // coefficients they stopped at and stored -- the loop itself then stores the iterate in its last iteration only.  0: an ordinary pass
    int only_parity, only_below;
};

__device__ __forceinline__ c32 load_x(const void* x, int dtype, size_t g, c32*)
{
    return dtype == 0 ? reinterpret_cast<const c32*>(x)[g] : c32{reinterpret_cast<const float*>(x)[g], 0.f};
}
__device__ __forceinline__ float load_x(const void* x, int, size_t g, float*) { return reinterpret_cast<const float*>(x)[g]; }
// the same from a slice base with a 32-bit element offset
__device__ __forceinline__ c32 load_xs(const void* xs, int dtype, unsigned g, c32*)
{
    return dtype == 0 ? reinterpret_cast<const c32*>(xs)[g] : c32{reinterpret_cast<const float*>(xs)[g], 0.f};
}
__device__ __forceinline__ float load_xs(const void* xs, int, unsigned g, float*) { return reinterpret_cast<const float*>(xs)[g]; }
__device__ __forceinline__ void store_out(void* out, int dtype, size_t g, c32 v)
{
    if (dtype == 0) reinterpret_cast<c32*>(out)[g] = v;
    else reinterpret_cast<float*>(out)[g] = v.x;
}
__device__ __forceinline__ void store_out(void* out, int, size_t g, float v) { reinterpret_cast<float*>(out)[g] = v; }

// synthesis of one level: (a, cH, cV, cD) (Ho x Wo valid samples each; a may sit in a larger array) -> rec (RH x RW).
// Workgroup = 2 TILE x 2 TILE output samples; same thread layout and LDS economy as the analysis kernel.
template <typename T, int TILE, int LT>
__global__ __launch_bounds__((WTILE_NT<T, TILE>)) void idwt2_tile_kernel(const T* a, size_t a_ld, size_t a_slice, const T* det, size_t det_slice, int Ho, int Wo, T* rec,
                                                         size_t rec_slice, int RH, int RW, Filters f, int tiles_x, int ntiles, int ns, Update u)
{
    extern __shared__ __align__(16) unsigned char w_smem[];
    constexpr int NT = WTILE_NT<T, TILE>;
    __shared__ double red[NT / 64];
    constexpr int LX = TILE, LY = NT / TILE, OH = 2 * TILE, OW = 2 * TILE, R = OH / LY;   // R output rows per thread along axis 0
    const int L = LT ? LT : f.len, HL = L / 2, KH = TILE + HL - 1, KW = KH;
    T* s_a = reinterpret_cast<T*>(w_smem);
    T* s_h = s_a + (size_t)KH * KW;
    T* s_v = s_h + (size_t)KH * KW;
    T* s_d = s_v + (size_t)KH * KW;
    T* s_lo = s_d + (size_t)KH * KW;   // [OH][KW]
    T* s_hi = s_lo + (size_t)OH * KW;
    const int tx = threadIdx.x % LX, ty = threadIdx.x / LX;
    int s, tile;
    if (!xcd_decode(ntiles, ns, s, tile)) return;
    const int by = tile / tiles_x, bx = tile - by * tiles_x;
    const int m0 = by * OH, n0 = bx * OW, kr0 = m0 / 2, kc0 = n0 / 2;
    const int dn = u.done ? u.done[s] : 0;
    if (!u.enabled && dn != 0) return;   // a level above the first of a finished slice (early exit): its arrays stay as they are
    const bool fin = u.enabled && u.only_parity != 0;
    if (fin && !(dn > 0 && dn < u.only_below && (u.only_parity == 3 || ((dn - 1) & 1) == u.only_parity - 1))) return;   // (uniform over the workgroup)
    const size_t cnt = (size_t)Ho * Wo;
    const T* pa = a + (size_t)s * a_slice;
    const T* pd = det + (size_t)s * det_slice;
    __shared__ float4 s_tap[MAXL / 2];
    if ((int)threadIdx.x < HL) {
        const int j = L - 2 - 2 * threadIdx.x;
        s_tap[threadIdx.x] = float4{f.rec_lo[j], f.rec_hi[j], f.rec_lo[j + 1], f.rec_hi[j + 1]};
    }
    // valid outputs of this tile (the last tile of an axis is mostly empty) and the coefficients they need
    const int nvh = min(OH, (u.enabled ? u.n1 : RH) - m0), nvw = min(OW, (u.enabled ? u.n2 : RW) - n0);
    const int KHv = (nvh + 1) / 2 + HL - 1, KWv = (nvw + 1) / 2 + HL - 1;
    if constexpr (LT != 0) {
        // the KH x KW coefficients of the tile as ONE index range over the 256 threads (5 loads per array and thread for db4
        // instead of 5 x 2, half of them for the three columns beyond the lanes' own); LDS index = that index (pitch KW)
        constexpr int KHc = TILE + LT / 2 - 1, KWc = KHc, NE = (KHc * KWc + NT - 1) / NT;
        T va[NE], vh[NE], vv[NE], vd[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = (int)threadIdx.x + k * NT, kr = e / KWc, kc = e - kr * KWc, gr = kr0 + kr, gc = kc0 + kc;
            va[k] = vh[k] = vv[k] = vd[k] = zero_of<T>();
            if (kr < KHv && kc < KWv && gr < Ho && gc < Wo) {
                const size_t o = (size_t)gr * Wo + gc;
                va[k] = pa[(size_t)gr * a_ld + gc];
                vh[k] = pd[o];
                vv[k] = pd[cnt + o];
                vd[k] = pd[2 * cnt + o];
            }
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = (int)threadIdx.x + k * NT, kr = e / KWc, kc = e - kr * KWc;
            if (kr < KHv && kc < KWv) { s_a[e] = va[k]; s_h[e] = vh[k]; s_v[e] = vv[k]; s_d[e] = vd[k]; }
        }
    } else {
        constexpr int KR = TILE == 32 ? 5 : 3, MC = (TILE + MAXL / 2 - 1 + LX - 1) / LX;
        for (int rb = ty; rb < KHv; rb += KR * LY) {
            T va[KR][MC], vh[KR][MC], vv[KR][MC], vd[KR][MC];
    #pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int kr = rb + k * LY, gr = kr0 + kr;
    #pragma unroll
                for (int m = 0; m < MC; ++m) {
                    const int kc = tx + m * LX, gc = kc0 + kc;
                    va[k][m] = vh[k][m] = vv[k][m] = vd[k][m] = zero_of<T>();
                    if (kr < KHv && kc < KWv && gr < Ho && gc < Wo) {
                        const size_t o = (size_t)gr * Wo + gc;
                        va[k][m] = pa[(size_t)gr * a_ld + gc];
                        vh[k][m] = pd[o];
                        vv[k][m] = pd[cnt + o];
                        vd[k][m] = pd[2 * cnt + o];
                    }
                }
            }
    #pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int kr = rb + k * LY;
    #pragma unroll
                for (int m = 0; m < MC; ++m) {
                    const int kc = tx + m * LX, i = kr * KW + kc;
                    if (kr < KHv && kc < KWv) { s_a[i] = va[k][m]; s_h[i] = vh[k][m]; s_v[i] = vv[k][m]; s_d[i] = vd[k][m]; }
                }
            }
        }
    }
    __syncthreads();
    // undo axis 0: out[m] = sum_t a[m/2 + t] rec_lo[(m&1) + L-2 - 2t] + d[...] rec_hi[...], t < L/2.  Column kc, output rows
    // ml = ty*R + q; rows 2i and 2i+1 read the same coefficients.  Taps from LDS, as in the analysis kernel.
    auto undo_axis0 = [&](const int kc, const int tyg) {
        Acc<T> lo[R / 2], hi[R / 2];   // (even row, odd row) pairs
        const int i0 = (tyg * R / 2) * KW + kc;
        for (int t = 0; t < HL; ++t) {
            const float4 g = s_tap[t];        // rec_lo[L-2-2t], rec_hi[L-2-2t] (even rows), rec_lo[L-1-2t], rec_hi[L-1-2t] (odd rows)
#pragma unroll
            for (int q = 0; q < R / 2; ++q) {
                const int i = i0 + (q + t) * KW;
                lo[q].tap(g.x, g.z, s_a[i]);
                lo[q].tap(g.y, g.w, s_h[i]);
                hi[q].tap(g.x, g.z, s_v[i]);
                hi[q].tap(g.y, g.w, s_d[i]);
            }
        }
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            s_lo[(tyg * R + 2 * q) * KW + kc] = lo[q].first();
            s_lo[(tyg * R + 2 * q + 1) * KW + kc] = lo[q].second();
            s_hi[(tyg * R + 2 * q) * KW + kc] = hi[q].first();
            s_hi[(tyg * R + 2 * q + 1) * KW + kc] = hi[q].second();
        }
    };
    if (tx < KWv && ty * R < nvh) undo_axis0(tx, ty);
    // the L/2 - 1 columns beyond the lanes' own (3 for db4): as (column, row group) items on the first lanes of the workgroup --
    // a second sweep `kc += LX` cost every wavefront a full pass for three active lanes
    {
        const int extra = KWv - LX;
        for (int it = threadIdx.x; it < extra * LY; it += NT) {
            const int tyg = it / extra, kc = LX + (it - tyg * extra);
            if (tyg * R < nvh) undo_axis0(kc, tyg);
        }
    }
    __syncthreads();
    // undo axis 1: row ml = ty + LY*i, outputs n = 2 tx and 2 tx + 1 share the coefficients k = tx .. tx + L/2 - 1
    constexpr int NR = OH / LY;
    T ve[NR], vo[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        Acc<T> eo;
        const T* ql = s_lo + (ty + LY * i) * KW + tx;
        const T* qh = s_hi + (ty + LY * i) * KW + tx;
        for (int t = 0; t < HL && ty + LY * i < nvh && 2 * tx < nvw; ++t) {
            const float4 g = s_tap[t];
            eo.tap(g.x, g.z, ql[t]);
            eo.tap(g.y, g.w, qh[t]);
        }
        ve[i] = eo.first();
        vo[i] = eo.second();
    }
    // per-thread partial sums in float (a thread adds at most 2 NR samples), combined in double: a convert and a double add per sample
    // were 3 % of the vector instructions of these issue-bound kernels, and the sums only feed the convergence test (POCS.py:622)
    float facc = 0.f;
    const int n = n0 + 2 * tx;
    if (!u.enabled) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = m0 + ty + LY * i;
            if (m >= RH) continue;
            T* o = rec + (size_t)s * rec_slice + (size_t)m * RW + n;
            if (n < RW) o[0] = ve[i];
            if (n + 1 < RW) o[1] = vo[i];
        }
    } else if (dn != 0 && !fin) {
        if (dn < 0 && u.zero_fill) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int m = m0 + ty + LY * i;
                if (m >= u.n1) continue;
                const size_t g = (size_t)s * u.n1 * u.n2 + (size_t)m * u.n2 + n;
                if (n < u.n2) store_out(u.out, u.dtype, g, zero_of<T>());
                if (n + 1 < u.n2) store_out(u.out, u.dtype, g + 1, zero_of<T>());
            }
        }
    } else {
        // crop to the slice + re-insertion (POCS.py:609, 616-619): all loads first, then the arithmetic and the stores
        const size_t per = (size_t)u.n1 * u.n2;
        T xo[NR][2];
        float mk[NR][2];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = m0 + ty + LY * i;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                xo[i][e] = zero_of<T>();
                mk[i][e] = 0.f;
                if (m < u.n1 && n + e < u.n2) {
                    const size_t li = (size_t)m * u.n2 + n + e;
                    xo[i][e] = load_x(u.x, u.dtype, (size_t)s * per + li, (T*)nullptr);
                    mk[i][e] = u.mask[li];
                }
            }
        }
        T* feed = reinterpret_cast<T*>(u.feed);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = m0 + ty + LY * i;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (m >= u.n1 || n + e >= u.n2) continue;
                const size_t g = (size_t)s * per + (size_t)m * u.n2 + n + e;
                const float wgt = 1.0f - u.alpha * mk[i][e];
                const T xn = cmulf(e ? vo[i] : ve[i], wgt) + cmulf(xo[i][e], u.alpha);
                if (u.write_out) store_out(u.out, u.dtype, g, xn);
                facc += mag(xn);
                if (fin) continue;   // (the slice is finished: nothing feeds on it)
                if (u.adaptive) feed[g] = (cmulf(xo[i][e], u.alpha) + cmulf(xn, wgt)) + cmulf(xo[i][e] - cmulf(xn, mk[i][e]), 1.0f - u.alpha);
                else feed[g] = xn;
            }
        }
    }
    if (u.enabled) {   // tile sum: shuffle tree per wavefront, then the four wavefronts through LDS
        double acc = (double)facc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0 && dn == 0) {
            double tot = (red[0] + red[1]) + (red[2] + red[3]);
            for (int w = 4; w < NT / 64; w += 4) tot += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
            atomicAdd(u.sums + s, tot);
        }
    }
}

// ---- level 1, steady state: synthesis of iteration k + re-insertion + analysis of iteration k + 1 in ONE kernel --------------------
// Between two iterations the level-1 synthesis writes the new iterate (crop + re-insertion fused, idwt2_tile_kernel) and the level-1
// analysis reads it straight back (dwt2_tile_kernel): 4 B/point written and 4.8 B/point read (halo) that exist only to cross a
// kernel boundary -- a third of the loop's traffic.  Here a workgroup owns a TILE x TILE tile of the NEXT decomposition: it
// reconstructs the (2 TILE + L - 2)^2 samples of the iterate that tile needs (its own 2 TILE x 2 TILE core plus the analysis halo,
// recomputed rather than exchanged: x 1.2 synthesis work), applies crop + re-insertion (POCS.py:609, 616-619), extends by the
// 'smooth' rule where the region leaves the slice, and filters both axes again; the iterate itself is never stored (`out` only when
// asked: last iteration / early exit).  Cost sums and `out` cover the core, every sample exactly once.  The arithmetic of every
// sample is that of the two kernels it replaces (same taps in the same order, same re-insertion expression): bit-identical results
// (tests/test_gpu_wavelet.py), P3D_WAVELET_NO_L1FUSE=1 keeps the two launches.
// The level-1 details are read (iteration k, with halo) and written (iteration k + 1) by different workgroups at the same time:
// they alternate between two buffers.
template <int TILE>
__host__ __device__ constexpr size_t wfuse1_lds_elems(int L) { return (size_t)4 * (TILE + L - 2) * (TILE + L - 2) + (size_t)2 * (2 * TILE + L - 2) * (TILE + L - 2); }

// Threads of a workgroup (a multiple of 256, at most 1024).  The tile's LDS image (44 KiB of float, 88 KiB of complex samples) allows three
// / one workgroups per CU whatever their size, so the size sets how many wavefronts cover each other's LDS and memory waits between the
// five barriers of a tile.  BASELINE configs[3] (float32), 256 / 512 / 1024 threads: 0.574 / 0.535 / 0.595 ms per iteration (12 / 24 / 32
// wavefronts per CU; at 1024 the barriers of sixteen wavefronts cost more than the occupancy returns); a 512 x 512 x 128 complex64
// cube (one workgroup per CU): 0.90 / 0.73 / 0.69 ms (profiles/r04_wavelet_workgroup_size.txt).
#ifndef P3D_WFUSE1_THREADS
#define P3D_WFUSE1_THREADS 512
#endif
#ifndef P3D_WFUSE1_THREADS_COMPLEX
#define P3D_WFUSE1_THREADS_COMPLEX 1024
#endif
template <typename T> constexpr int WFUSE1_NT = sizeof(T) == sizeof(float) ? P3D_WFUSE1_THREADS : P3D_WFUSE1_THREADS_COMPLEX;
// MBITS (LT != 0): the mask weights of the tile come as bits (Update::mask_bits: every weight is 0 or 1) -- 280 words per tile through LDS instead of ten
// float loads per thread, a quarter of the tile's load instructions
template <typename T, int TILE, int LT, bool MBITS>
__global__ __launch_bounds__(WFUSE1_NT<T>) void wfuse1_kernel(const T* a, size_t a_ld, size_t a_slice, const T* det_in, size_t det_in_slice, T* det_out, size_t det_out_slice,
                                                     int Ho, int Wo, T* cA, size_t cA_slice, Filters f, int tiles_x, int ntiles, int ns, Update u, Thresh th)
{
    extern __shared__ __align__(16) unsigned char w_smem[];
    constexpr int NT = WFUSE1_NT<T>;
    __shared__ double red[NT / 64];
    __shared__ float4 s_dec[MAXL / 2], s_rec[MAXL / 2];
    constexpr int MBW = 4;   // words of a tile row: 2 TILE + L - 2 <= 126 columns from any bit offset
    __shared__ unsigned s_mb[MBITS ? (2 * TILE + (LT ? LT : 2) - 2) * MBW : 1];
    constexpr int LX = TILE, LY = NT / TILE, R = TILE / LY;
    const int L = LT ? LT : f.len, HL = L / 2, IH = 2 * TILE + L - 2, IW = IH, KH = TILE + L - 2, KW = KH;
    T* s_a = reinterpret_cast<T*>(w_smem);           // synthesis: four coefficient arrays [KH][KW] ...
    T* s_h = s_a + (size_t)KH * KW;
    T* s_v = s_h + (size_t)KH * KW;
    T* s_d = s_v + (size_t)KH * KW;
    T* s_lo = s_d + (size_t)KH * KW;                 // ... and the column-reconstructed halves [IH][KW]; analysis: row-filtered [IH][TILE] x 2
    T* s_hi = s_lo + (size_t)IH * KW;
    T* s_in = s_a;                                   // the iterate of the tile [IH][IW] takes the place of the coefficients (IH * IW <= 4 KH KW)
    const int tid = threadIdx.x, tx = tid % LX, ty = tid / LX;
    int s, tile;
    if (!xcd_decode(ntiles, ns, s, tile)) return;
    const int dn = u.done ? u.done[s] : 0;
    if (dn != 0) return;                             // finished / empty slice: nothing feeds on it any more (uniform over the workgroup)
    const int by = tile / tiles_x, bx = tile - by * tiles_x;
    const int or0 = by * TILE, oc0 = bx * TILE, r0 = 2 * or0 - L + 2, c0 = 2 * oc0 - L + 2;
    const int vh = min(TILE, Ho - or0), vw = min(TILE, Wo - oc0), IHv = 2 * vh + L - 2, IWv = 2 * vw + L - 2;
    // the part of the region that lies inside the slice: rows m_lo ... m_hi (m_lo even: 0 or r0), columns n_lo ... n_hi
    const int m_lo = max(r0, 0), m_hi = min(r0 + IHv, u.n1) - 1, n_lo = max(c0, 0), n_hi = min(c0 + IWv, u.n2) - 1;
    const int nm = m_hi - m_lo + 1, nn = n_hi - n_lo + 1;            // (>= 2: a tile reaches at least L - 2 >= 2 samples into the slice)
    const int kr0 = m_lo / 2, kc0 = n_lo / 2, KHv = m_hi / 2 + HL - kr0, KWv = n_hi / 2 + HL - kc0;
    if (tid < HL) {
        const int j = 2 * tid, k = L - 2 - 2 * tid;
        s_dec[tid] = float4{f.dec_lo[j], f.dec_hi[j], f.dec_lo[j + 1], f.dec_hi[j + 1]};
        s_rec[tid] = float4{f.rec_lo[k], f.rec_hi[k], f.rec_lo[k + 1], f.rec_hi[k + 1]};
    }
    const size_t cnt = (size_t)Ho * Wo;
    const T* pa = a + (size_t)s * a_slice;
    const T* pd = det_in + (size_t)s * det_in_slice;
    // ---- coefficients of the region (zero beyond the arrays, as in idwt2_tile_kernel) and, right behind them, the observed samples
    //      and mask weights of the region: all requests of a thread are in flight together (a load per loop trip would pay the
    //      memory latency once per trip -- ten trips per tile) ----
    constexpr int IHc = 2 * TILE + (LT ? LT : 2) - 2, KHc = TILE + (LT ? LT : 2) - 2;
    constexpr int NE = LT ? (KHc * KHc + NT - 1) / NT : 1, NIT = LT ? (IHc * (IHc / 2) + NT - 1) / NT : 1;
    const size_t per = (size_t)u.n1 * u.n2;
    const T* const pd1 = pd + cnt;
    const T* const pd2 = pd + 2 * cnt;
    const void* const xs = reinterpret_cast<const char*>(u.x) + (size_t)s * per * (u.dtype == 0 ? sizeof(c32) : sizeof(float));   // this slice's observed samples
    T xo[NIT][2];
    float mk[NIT][2];
    if constexpr (LT != 0) {
        T va[NE], vh_[NE], vv[NE], vd[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + NT * i, kr = e / KHc, kc = e - kr * KHc, gr = kr0 + kr, gc = kc0 + kc;
            va[i] = vh_[i] = vv[i] = vd[i] = zero_of<T>();
            if (e < KHc * KHc && kr < KHv && kc < KWv && gr < Ho && gc < Wo) {
                // (32-bit element offsets from wave-uniform bases: one level of one slice is far below 2^32 samples)
                const unsigned o = (unsigned)gr * (unsigned)Wo + (unsigned)gc;
                va[i] = pa[(unsigned)gr * (unsigned)a_ld + (unsigned)gc];
                vh_[i] = pd[o];
                vv[i] = pd1[o];
                vd[i] = pd2[o];
            }
        }
        // observed samples (m, 2 ii), (m, 2 ii + 1) of a thread: ONE load of the pair where rows have an even number of samples (n_lo is even: the pair
        // is aligned then and never straddles the region's right edge) and the cube's samples have the kernel's own type; else sample by sample
        const bool pairs = MBITS && (u.n2 & 1) == 0 && (u.dtype == 0) == (sizeof(T) == sizeof(c32)) && (reinterpret_cast<size_t>(xs) & (2 * sizeof(T) - 1)) == 0;
        if (pairs) {
            struct alignas(2 * sizeof(T)) XPair { T a, b; };
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + NT * it, m = e / (IHc / 2), ii = e - m * (IHc / 2);
                xo[it][0] = xo[it][1] = zero_of<T>();
                mk[it][0] = mk[it][1] = 0.f;
                const int gm = m_lo + m, gn = n_lo + 2 * ii;
                if (e < IHc * (IHc / 2) && m < nm && gn <= n_hi) {
                    const unsigned li = (unsigned)gm * (unsigned)u.n2 + (unsigned)gn;
                    const XPair v = *reinterpret_cast<const XPair*>(reinterpret_cast<const T*>(xs) + li);
                    xo[it][0] = v.a;
                    xo[it][1] = v.b;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + NT * it, m = e / (IHc / 2), ii = e - m * (IHc / 2);
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {
                    xo[it][ee] = zero_of<T>();
                    mk[it][ee] = 0.f;
                    const int gm = m_lo + m, gn = n_lo + 2 * ii + ee;
                    if (e < IHc * (IHc / 2) && m < nm && gn <= n_hi) {
                        const unsigned li = (unsigned)gm * (unsigned)u.n2 + (unsigned)gn;
                        xo[it][ee] = load_xs(xs, u.dtype, li, (T*)nullptr);
                        if constexpr (!MBITS) mk[it][ee] = u.mask[li];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + NT * i, kr = e / KHc, kc = e - kr * KHc;
            if (e < KHc * KHc && kr < KHv && kc < KWv) { s_a[e] = va[i]; s_h[e] = vh_[i]; s_v[e] = vv[i]; s_d[e] = vd[i]; }
        }
        if constexpr (MBITS) {   // rows m_lo ... of the region, words (n_lo / 32) ... + 3 of each
            for (int e = tid; e < nm * MBW; e += NT) {
                const int row = e / MBW, wi = (n_lo >> 5) + (e - row * MBW);
                s_mb[e] = wi < u.mask_words ? u.mask_bits[(unsigned)(m_lo + row) * (unsigned)u.mask_words + (unsigned)wi] : 0u;
            }
        }
    } else {
        for (int e = tid; e < KH * KW; e += NT) {
            const int kr = e / KW, kc = e - kr * KW, gr = kr0 + kr, gc = kc0 + kc;
            if (kr < KHv && kc < KWv) {
                T va = zero_of<T>(), vh_ = zero_of<T>(), vv = zero_of<T>(), vd = zero_of<T>();
                if (gr < Ho && gc < Wo) {
                    const size_t o = (size_t)gr * Wo + gc;
                    va = pa[(size_t)gr * a_ld + gc];
                    vh_ = pd[o];
                    vv = pd[cnt + o];
                    vd = pd[2 * cnt + o];
                }
                s_a[e] = va; s_h[e] = vh_; s_v[e] = vv; s_d[e] = vd;
            }
        }
    }
    __syncthreads();
    // ---- undo axis 0: rows m_lo + 2 ip, + 1 from coefficient rows ip ... ip + L/2 - 1 (local) ----
    for (int e = tid; e < (IH / 2) * KW; e += NT) {
        const int ip = e / KW, kc = e - ip * KW;
        if (2 * ip < nm && kc < KWv) {
            Acc<T> lo, hi;
            for (int t = 0; t < HL; ++t) {
                const float4 g = s_rec[t];
                const int i = (ip + t) * KW + kc;
                lo.tap(g.x, g.z, s_a[i]);
                lo.tap(g.y, g.w, s_h[i]);
                hi.tap(g.x, g.z, s_v[i]);
                hi.tap(g.y, g.w, s_d[i]);
            }
            s_lo[(2 * ip) * KW + kc] = lo.first();
            s_lo[(2 * ip + 1) * KW + kc] = lo.second();
            s_hi[(2 * ip) * KW + kc] = hi.first();
            s_hi[(2 * ip + 1) * KW + kc] = hi.second();
        }
    }
    __syncthreads();
    // ---- undo axis 1 + crop + re-insertion: samples (m_lo + m, n_lo + 2 ii), + 1 -> the tile's image of the iterate ----
    float facc = 0.f;   // (float per thread, double across threads: see idwt2_tile_kernel)
    {
        T* feed_img = s_in + (m_lo - r0) * IW + (n_lo - c0);
        const int core_m0 = 2 * or0, core_n0 = 2 * oc0;   // the tile OWNS rows core_m0 ... + 2 TILE - 1 (cost sum, `out`)
        auto sample = [&](int m, int ii, const T (&xo2)[2], const float (&mk2)[2]) {
            const T* ql = s_lo + m * KW + ii;
            const T* qh = s_hi + m * KW + ii;
            Acc<T> eo;
            for (int t = 0; t < HL; ++t) {
                const float4 g = s_rec[t];
                eo.tap(g.x, g.z, ql[t]);
                eo.tap(g.y, g.w, qh[t]);
            }
            const int gm = m_lo + m;
            const bool mine_m = (unsigned)(gm - core_m0) < (unsigned)(2 * TILE);
#pragma unroll
            for (int ee = 0; ee < 2; ++ee) {
                const int gn = n_lo + 2 * ii + ee;
                if (gn > n_hi) continue;
                const float wgt = 1.0f - u.alpha * mk2[ee];
                const T xn = cmulf(ee ? eo.second() : eo.first(), wgt) + cmulf(xo2[ee], u.alpha);
                if (mine_m && (unsigned)(gn - core_n0) < (unsigned)(2 * TILE)) {   // the tile's own core: cost sum, `out`
                    if (u.write_out) store_out(u.out, u.dtype, (size_t)s * per + (size_t)gm * u.n2 + gn, xn);
                    facc += mag(xn);
                }
                T fd = xn;
                if (u.adaptive) fd = (cmulf(xo2[ee], u.alpha) + cmulf(xn, wgt)) + cmulf(xo2[ee] - cmulf(xn, mk2[ee]), 1.0f - u.alpha);
                feed_img[m * IW + 2 * ii + ee] = fd;
            }
        };
        if constexpr (LT != 0) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + NT * it, m = e / (IHc / 2), ii = e - m * (IHc / 2);
                if (e < IHc * (IHc / 2) && m < nm && 2 * ii < nn) {
                    if constexpr (MBITS) {   // samples (m, 2 ii), (m, 2 ii + 1): n_lo is even, so both bits sit in one word
                        const int rel = (n_lo & 31) + 2 * ii;
                        const unsigned wv = s_mb[m * MBW + (rel >> 5)] >> (rel & 31);
                        mk[it][0] = (float)(wv & 1u);
                        mk[it][1] = (float)((wv >> 1) & 1u);
                    }
                    sample(m, ii, xo[it], mk[it]);
                }
            }
        } else {
            for (int e = tid; e < IH * (IW / 2); e += NT) {
                const int m = e / (IW / 2), ii = e - m * (IW / 2);
                if (m >= nm || 2 * ii >= nn) continue;
                T x2[2];
                float m2[2];
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {
                    x2[ee] = zero_of<T>();
                    m2[ee] = 0.f;
                    const int gn = n_lo + 2 * ii + ee;
                    if (gn <= n_hi) {
                        const size_t li = (size_t)(m_lo + m) * u.n2 + gn;
                        x2[ee] = load_x(u.x, u.dtype, (size_t)s * per + li, (T*)nullptr);
                        m2[ee] = u.mask[li];
                    }
                }
                sample(m, ii, x2, m2);
            }
        }
    }
    __syncthreads();
    // ---- 'smooth' extension where the region leaves the slice (smooth2_at on the image: along axis 0, then along axis 1) ----
    if (r0 < 0 || c0 < 0 || r0 + IHv > u.n1 || c0 + IWv > u.n2) {
        for (int e = tid; e < IH * IW; e += NT) {
            const int lr = e / IW, lc = e - lr * IW;
            if (lr >= IHv || lc >= IWv) continue;
            const int r = r0 + lr, c = c0 + lc;
            if (r >= 0 && r < u.n1 && c >= 0 && c < u.n2) continue;
            const int H = u.n1, W = u.n2;
            const int rc = min(max(r, 0), H - 1), cc = min(max(c, 0), W - 1);
            const int rn = r < 0 ? min(1, H - 1) : max(H - 2, 0), cn = c < 0 ? min(1, W - 1) : max(W - 2, 0);
            const float tr = r < 0 ? (float)(-r) : (r >= H ? (float)(r - H + 1) : 0.f);
            const float tc = c < 0 ? (float)(-c) : (c >= W ? (float)(c - W + 1) : 0.f);
            // (a neighbour that carries weight 0 is not read: its place may lie outside the tile's image, and 0 x garbage is not 0)
            const int lrc = rc - r0, lcc = cc - c0, lrn = tr != 0.f ? rn - r0 : lrc, lcn = tc != 0.f ? cn - c0 : lcc;
            const T x00 = s_in[lrc * IW + lcc], x10 = s_in[lrn * IW + lcc];
            const T x01 = s_in[lrc * IW + lcn], x11 = s_in[lrn * IW + lcn];
            s_in[e] = extrapolate(extrapolate(x00, x10, tr), extrapolate(x01, x11, tr), tc);
        }
        __syncthreads();
    }
    // ---- analysis of the next iteration: dwt2_tile_kernel from here on (s_lo / s_hi re-used with pitch TILE) ----
    struct alignas(2 * sizeof(T)) Pair { T lo, hi; };
    for (int r = ty; r < IHv; r += LY) {
        if (tx >= vw) continue;
        const Pair* q = reinterpret_cast<const Pair*>(s_in + r * IW + 2 * tx + L - 2);
        Acc<T> ad;
        for (int jj = 0; jj < HL; ++jj) {
            const float4 g = s_dec[jj];
            const Pair v = q[-jj];
            ad.tap(g.x, g.y, v.hi);
            ad.tap(g.z, g.w, v.lo);
        }
        s_lo[r * TILE + tx] = ad.first();
        s_hi[r * TILE + tx] = ad.second();
    }
    __syncthreads();
    if (tx < vw && ty * R < vh) {
        Acc<T> fl[R], fh[R];
        const T* cl = s_lo + (2 * ty * R + L - 1) * TILE + tx;
        const T* ch = s_hi + (2 * ty * R + L - 1) * TILE + tx;
        for (int jj = 0; jj < HL; ++jj) {
            const float4 g = s_dec[jj];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int o = (2 * q - 2 * jj) * TILE;
                fl[q].tap(g.x, g.y, cl[o]);
                fh[q].tap(g.x, g.y, ch[o]);
                fl[q].tap(g.z, g.w, cl[o - TILE]);
                fh[q].tap(g.z, g.w, ch[o - TILE]);
            }
        }
        c32 t0{0.f, 0.f}, t1{0.f, 0.f}, t2{0.f, 0.f};
        if (th.tau) {
            const c32* t = th.tau + (((size_t)s * th.niter + th.iter) * th.nlev + th.lvl) * 3;
            t0 = t[0]; t1 = t[1]; t2 = t[2];
        }
        const ShrinkTile<T> sh0(t0, th.op), sh1(t1, th.op), sh2(t2, th.op);
        const int gc = oc0 + tx;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int go = or0 + ty * R + q;
            if (go >= Ho || gc >= Wo) continue;
            T da = fl[q].second(), ad = fh[q].first(), dd = fh[q].second();
            if (th.tau) { da = sh0(da); ad = sh1(ad); dd = sh2(dd); }
            const size_t o = (size_t)go * Wo + gc;
            cA[(size_t)s * cA_slice + o] = fl[q].first();
            T* dp = det_out + (size_t)s * det_out_slice + o;
            dp[0] = da;
            dp[cnt] = ad;
            dp[2 * cnt] = dd;
        }
    }
    // ---- cost sum of the core ----
    double acc = (double)facc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double tot = (red[0] + red[1]) + (red[2] + red[3]);
        for (int w = 4; w < NT / 64; w += 4) tot += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
        atomicAdd(u.sums + s, tot);
    }
}

// per (slice, level, detail): lexicographic max, max |d|, min |d| -> stats[((s*nlev + lvl)*3 + z)*4 ..]; one block each
__device__ __forceinline__ float re_of(c32 v) { return v.x; }
__device__ __forceinline__ float im_of(c32 v) { return v.y; }
__device__ __forceinline__ float re_of(float v) { return v; }
__device__ __forceinline__ float im_of(float) { return 0.f; }

// statistics of every detail array of every level in ONE launch (grid: level index coarse -> fine, slice, detail array)
constexpr int WSTAT_MAX_LEVELS = 16;
struct WStatLevels {
    size_t off[WSTAT_MAX_LEVELS], count[WSTAT_MAX_LEVELS];   // per level index (0 = coarsest): first detail array and samples per array
};
template <typename T>
__global__ void wstats_kernel(const T* coef, size_t coef_slice, const WStatLevels lv, float* stats, int nlev)
{
    __shared__ float sh[4 * 4];
    const int lvl = blockIdx.x, s = blockIdx.y, z = blockIdx.z;
    const size_t count = lv.count[lvl];
    const T* p = coef + (size_t)s * coef_slice + lv.off[lvl] + (size_t)z * count;
    float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY;
    for (size_t i = threadIdx.x; i < count; i += blockDim.x) {
        const T v = p[i];
        const float vr = re_of(v), vi = im_of(v), q = mag(v);
        if (vr > lr || (vr == lr && vi > li)) { lr = vr; li = vi; }
        mx = fmaxf(mx, q);
        mn = fminf(mn, q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
        if (orr > lr || (orr == lr && oi > li)) { lr = orr; li = oi; }
        mx = fmaxf(mx, __shfl_down(mx, o, 64));
        mn = fminf(mn, __shfl_down(mn, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        float* me = sh + (threadIdx.x >> 6) * 4;
        me[0] = lr; me[1] = li; me[2] = mx; me[3] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < (int)blockDim.x / 64; ++t) {
            const float* o = sh + t * 4;
            if (o[0] > lr || (o[0] == lr && o[1] > li)) { lr = o[0]; li = o[1]; }
            mx = fmaxf(mx, o[2]);
            mn = fminf(mn, o[3]);
        }
        float* q = stats + (((size_t)s * nlev + lvl) * 3 + z) * 4;
        q[0] = lr; q[1] = li; q[2] = mx; q[3] = mn;
    }
}

// mode 0: first input (feed = x or its APOCS mix; sums += |x|)
// mode 1: crop of the reconstruction + re-insertion (POCS.py:609, 616-619), sums += |x_new|, feed for the next iteration
// T = element type of the work buffers (float requires dtype == P3D_F32)
template <typename T>
__global__ void wupdate_kernel(const T* rec, size_t rec_ld, size_t rec_slice, T* feed, const void* x, int dtype, const float* mask, void* out, double* sums,
                               int mode, int adaptive, int write_out, float alpha, int n1, int n2, const int* done, int zero_fill)
{
    __shared__ double sh[256];
    const int s = blockIdx.y;
    const size_t per = (size_t)n1 * n2;
    const int dn = done ? done[s] : 0;
    if (zero_fill && dn < 0)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x)
            store_out(out, dtype, (size_t)s * per + i, zero_of<T>());
    double acc = 0.0;
    if (dn == 0) {
        // a block walks over rows (no division per sample: the 64-bit one of a flat index cost more than the sample's memory traffic)
        for (int r = blockIdx.x; r < n1; r += gridDim.x) {
            const size_t g0 = (size_t)s * per + (size_t)r * n2;
            const T* const rrow = rec ? rec + (size_t)s * rec_slice + (size_t)r * rec_ld : nullptr;
            const float* const mrow = mask ? mask + (size_t)r * n2 : nullptr;
            float racc = 0.f;
            for (int c = threadIdx.x; c < n2; c += blockDim.x) {
                const size_t g = g0 + c;
                const T xo = load_x(x, dtype, g, (T*)nullptr);
                const float m = mrow ? mrow[c] : 0.f;
                const float wgt = 1.0f - alpha * m;
                T xn;
                if (mode == 0) {
                    xn = xo;
                } else {
                    xn = cmulf(rrow[c], wgt) + cmulf(xo, alpha);
                    if (write_out) store_out(out, dtype, g, xn);
                }
                racc += mag(xn);
                if (adaptive) {
                    const T blend = cmulf(xo, alpha) + cmulf(xn, wgt);
                    feed[g] = blend + cmulf(xo - cmulf(xn, m), 1.0f - alpha);
                } else {
                    feed[g] = xn;
                }
            }
            acc += (double)racc;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && dn == 0) atomicAdd(sums + s, sh[0]);
}

__global__ void wconv_kernel(const double* sums, int* done, int nslices, int iter, double eps)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices || done[s] != 0) return;
    const double cur = sums[(size_t)(iter + 1) * nslices + s], prev = sums[(size_t)iter * nslices + s];
    const double d = cur - prev;
    if (iter > 2 && (d * d) / (cur * cur) < eps) done[s] = iter + 1;
}

int wfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    p3d::set_last_error(buf);
    return code;
}
#define W_TRY(expr)                                                                                     \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return wfail(P3D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline unsigned blocks_for(size_t n) { const size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

}  // namespace

struct p3d_wplan {
    int device = 0, nil = 0, nxl = 0, max_slices = 0, nlev = 0;
    Filters f{};
    std::vector<int> h, w;          // h[l], w[l], l = 0..nlev  (level 0 = the slice itself)
    std::vector<int> rh, rw;        // shape of the reconstruction OF level l (l = 0..nlev-1): 2*h[l+1] - L + 2
    std::vector<size_t> doff;       // offset of level-l details (l = 1..nlev) in the coefficient vector of one slice
    size_t ncoef = 0;               // cA first, then details of level nlev (coarsest) ... level 1: PyWavelets' order
    hipStream_t stream = nullptr;
    c32 *coef = nullptr, *feed = nullptr, *lo = nullptr, *hi = nullptr, *tau = nullptr;
    std::vector<c32*> approx;       // approx[l], l = 1..nlev-1 (approx[nlev] = coef, approx[0] = feed)
    std::vector<c32*> rec;          // rec[l], l = 0..nlev-1
    double* sums = nullptr;
    size_t sums_cap = 0, tau_cap = 0;
    int* done = nullptr;
    float *stats = nullptr, *mask = nullptr;
    unsigned* mask_bits = nullptr;   // one bit per sample of the mask, rows of mask_words words, + one int behind them: 1 = every weight is 0 or 1
    int mask_words = 0, mask_binary = 0;
    void *st_x = nullptr, *st_out = nullptr;
    // the observed cube and the result of the job in progress: the caller's own device buffers where it passed such (no staging
    // copies: 0.4 ms each way per 256 MiB), the staging buffers above otherwise
    const void* cur_x = nullptr;
    void* cur_out = nullptr;
    const int* loop_done = nullptr;   // the loop with the early exit: the per-slice states, for the passes that have no Update / Thresh of the caller's
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool fused = true;              // tile kernels (one launch per level and direction); P3D_WAVELET_UNFUSED=1 selects the per-axis kernels
    int tile_c = 0, tile_r = 0;     // coefficients per tile edge for complex64 / float32 work buffers
    // level 1 of the steady state in one kernel (wfuse1_kernel): synthesis of iteration k + re-insertion + analysis of iteration k + 1
    bool l1fuse_c = false, l1fuse_r = false;   // ... for complex64 / float32 work buffers
    c32* det1_alt = nullptr;        // second buffer of the level-1 details [max_slices][3 h[1] w[1]] (they alternate)
    // first level carried by the one-workgroup-per-slice kernel (wcoarse_kernel) for complex64 / float32 work buffers; 0: none
    int lc_c = 0, lc_r = 0;
    int cx_c = 0, cx_r = 0;         // ... elements of its LDS region X
    size_t clds_c = 0, clds_r = 0;  // ... its dynamic LDS bytes
    size_t per() const { return (size_t)nil * nxl; }
};

// LDS bytes of the tile kernels for `tile` coefficients per subband and edge (analysis) / 2*tile output samples per edge (synthesis)
static size_t tile_lds(int tile, int L, size_t esz)
{
    const size_t ih = 2 * (size_t)tile + L - 2, kh = (size_t)tile + L / 2 - 1;
    const size_t fwd = esz * (ih * ih + 2 * ih * tile);
    const size_t inv = esz * (4 * kh * kh + 2 * 2 * (size_t)tile * kh);
    return fwd > inv ? fwd : inv;
}
static int pick_tile(int L, size_t esz)
{
    if (L % 2) return 0;                                   // the pair loads of the tile kernels need an even tap count
    if (tile_lds(32, L, esz) <= 80 * 1024) return 32;      // two or more workgroups per CU
    return tile_lds(16, L, esz) <= 150 * 1024 ? 16 : 0;
}

// First level LC >= 2 from which the rest of a slice's pyramid fits one workgroup's LDS (see wcoarse_kernel); 0 if none does.
static int coarse_fit(const p3d_wplan* p, size_t esz, int* x_elems, size_t* lds)
{
    const int L = p->f.len;
    if (L % 2 || p->nlev > MAXLEV) return 0;
    const size_t budget = 160 * 1024 - 2 * sizeof(float4) * (MAXL / 2) - 256;
    for (int lc = 2; lc <= p->nlev; ++lc) {
        size_t x = (size_t)p->h[lc - 1] * p->w[lc - 1], y = 0;
        for (int l = lc; l <= p->nlev; ++l) {
            const size_t rh = 2 * (size_t)p->h[l] - L + 2, rw = 2 * (size_t)p->w[l] - L + 2;
            x = std::max(x, std::max((size_t)p->h[l] * p->w[l], rh * rw));
            y = std::max(y, 2 * std::max((2 * (size_t)p->h[l] + L - 2), rh) * (size_t)p->w[l]);
        }
        if ((x + y) * esz <= budget) { *x_elems = (int)x; *lds = (x + y) * esz; return lc; }
    }
    return 0;
}

extern "C" int p3d_wavelet_plan_destroy(p3d_wplan* p)
{
    if (!p) return P3D_OK;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    void* bufs[] = {p->det1_alt, p->coef, p->feed, p->lo, p->hi, p->tau, p->sums, p->done, p->stats, p->mask, p->mask_bits, p->st_x, p->st_out};
    for (void* b : bufs) if (b) hipFree(b);
    for (c32* b : p->approx) if (b) hipFree(b);
    for (c32* b : p->rec) if (b) hipFree(b);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return P3D_OK;
}

extern "C" int p3d_wavelet_plan_create(p3d_wplan** out, int device, int nil, int nxl, int max_slices, const double* dec_lo, const double* dec_hi,
                                       const double* rec_lo, const double* rec_hi, int flen, int level)
{
    if (!out || !dec_lo || !dec_hi || !rec_lo || !rec_hi) return wfail(P3D_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (nil < 1 || nxl < 1 || max_slices < 1 || max_slices > 65535) return wfail(P3D_ERR_INVALID, "bad shape / batch size");
    if (flen < 2 || flen > MAXL) return wfail(P3D_ERR_UNSUPPORTED, "filter length %d: 2..%d taps are supported", flen, MAXL);
    int ndev = 0;
    W_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return wfail(P3D_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    W_TRY(hipSetDevice(device));
    const int nmin = nil < nxl ? nil : nxl;
    int maxlev = 0;  // pywt.dwt_max_level(min(shape), flen)
    if (nmin >= flen - 1) maxlev = (int)std::floor(std::log2((double)nmin / (flen - 1.0)));
    if (maxlev < 0) maxlev = 0;
    if (level < 0) level = maxlev;
    if (level < 1) return wfail(P3D_ERR_UNSUPPORTED, "a %d x %d slice is too small for a %d-tap wavelet (0 levels)", nil, nxl, flen);

    p3d_wplan* p = new p3d_wplan;
    p->device = device; p->nil = nil; p->nxl = nxl; p->max_slices = max_slices; p->nlev = level;
    p->f.len = flen;
    for (int j = 0; j < flen; ++j) {
        p->f.dec_lo[j] = (float)dec_lo[j]; p->f.dec_hi[j] = (float)dec_hi[j];
        p->f.rec_lo[j] = (float)rec_lo[j]; p->f.rec_hi[j] = (float)rec_hi[j];
    }
    p->h.assign(level + 1, 0); p->w.assign(level + 1, 0);
    p->h[0] = nil; p->w[0] = nxl;
    for (int l = 1; l <= level; ++l) { p->h[l] = (p->h[l - 1] + flen - 1) / 2; p->w[l] = (p->w[l - 1] + flen - 1) / 2; }
    p->rh.assign(level, 0); p->rw.assign(level, 0);
    for (int l = 0; l < level; ++l) { p->rh[l] = 2 * p->h[l + 1] - flen + 2; p->rw[l] = 2 * p->w[l + 1] - flen + 2; }
    p->doff.assign(level + 1, 0);
    size_t off = (size_t)p->h[level] * p->w[level];
    for (int l = level; l >= 1; --l) { p->doff[l] = off; off += 3 * (size_t)p->h[l] * p->w[l]; }
    p->ncoef = off;

    auto bail = [&](const char* what, hipError_t e) {
        p3d_wavelet_plan_destroy(p);
        return wfail(P3D_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
#define ALLOC(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return bail(#ptr, e)
    if ((e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess) return bail("stream", e);
    if ((e = hipEventCreate(&p->ev0)) != hipSuccess) return bail("event", e);
    if ((e = hipEventCreate(&p->ev1)) != hipSuccess) return bail("event", e);
    const size_t S = (size_t)max_slices;
    ALLOC(p->coef, sizeof(c32) * p->ncoef * S);
    ALLOC(p->feed, sizeof(c32) * p->per() * S);
    // row-filtered intermediates of the forward step / column-reconstructed intermediates of the inverse step: the
    // largest are (h[0] x w[1]) and (rh[0] x w[1])
    const size_t tmp = (size_t)std::max(p->h[0], p->rh[0]) * p->w[1];
    ALLOC(p->lo, sizeof(c32) * tmp * S);
    ALLOC(p->hi, sizeof(c32) * tmp * S);
    p->approx.assign(level + 1, nullptr);
    for (int l = 1; l < level; ++l) ALLOC(p->approx[l], sizeof(c32) * (size_t)p->h[l] * p->w[l] * S);
    p->rec.assign(level, nullptr);
    for (int l = 0; l < level; ++l) ALLOC(p->rec[l], sizeof(c32) * (size_t)p->rh[l] * p->rw[l] * S);
    ALLOC(p->done, sizeof(int) * S);
    ALLOC(p->stats, sizeof(float) * 4 * 3 * (size_t)level * S);
    ALLOC(p->mask, sizeof(float) * p->per());
    p->mask_words = (nxl + 31) / 32;
    ALLOC(p->mask_bits, sizeof(unsigned) * ((size_t)nil * p->mask_words + 1));
    ALLOC(p->st_x, sizeof(c32) * p->per() * S);
    ALLOC(p->st_out, sizeof(c32) * p->per() * S);
    p->tile_c = pick_tile(flen, sizeof(c32));
    p->tile_r = pick_tile(flen, sizeof(float));
    const char* env = getenv("P3D_WAVELET_UNFUSED");
    p->fused = !(env && env[0] == '1') && p->tile_c > 0 && p->tile_r > 0;
    if (p->fused && !getenv("P3D_WAVELET_NO_COARSE")) {
        p->lc_c = coarse_fit(p, sizeof(c32), &p->cx_c, &p->clds_c);
        p->lc_r = coarse_fit(p, sizeof(float), &p->cx_r, &p->clds_r);
        const void* ck[] = {(const void*)wcoarse_kernel<c32, 0>, (const void*)wcoarse_kernel<c32, 4>, (const void*)wcoarse_kernel<c32, 8>,
                            (const void*)wcoarse_kernel<float, 0>, (const void*)wcoarse_kernel<float, 4>, (const void*)wcoarse_kernel<float, 8>};
        for (const void* k : ck)
            if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2 * (int)sizeof(float4) * (MAXL / 2) - 256)) != hipSuccess)
                return bail("hipFuncSetAttribute", e);
    }
    if (p->fused && p->nlev >= 2 && !getenv("P3D_WAVELET_NO_L1FUSE")) {
        // every tile of the level-1 decomposition must reach at least two samples into the slice along both axes (the 'smooth'
        // extension is evaluated on the tile's own image of the iterate)
        auto reaches = [&](int n, int no) { const int last = ((no - 1) / 32) * 32; return n >= 2 && 2 * last - flen + 2 <= n - 2; };
        const bool geom = reaches(nil, p->h[1]) && reaches(nxl, p->w[1]);
        p->l1fuse_c = geom && p->tile_c == 32 && wfuse1_lds_elems<32>(flen) * sizeof(c32) <= 150 * 1024;
        p->l1fuse_r = geom && p->tile_r == 32 && wfuse1_lds_elems<32>(flen) * sizeof(float) <= 150 * 1024;
        if (p->l1fuse_c || p->l1fuse_r) {
            ALLOC(p->det1_alt, sizeof(c32) * 3 * (size_t)p->h[1] * p->w[1] * S);
            const void* fk[] = {(const void*)wfuse1_kernel<c32, 32, 0, false>, (const void*)wfuse1_kernel<c32, 32, 4, false>, (const void*)wfuse1_kernel<c32, 32, 8, false>,
                                (const void*)wfuse1_kernel<float, 32, 0, false>, (const void*)wfuse1_kernel<float, 32, 4, false>, (const void*)wfuse1_kernel<float, 32, 8, false>,
                                (const void*)wfuse1_kernel<c32, 32, 4, true>, (const void*)wfuse1_kernel<c32, 32, 8, true>,
                                (const void*)wfuse1_kernel<float, 32, 4, true>, (const void*)wfuse1_kernel<float, 32, 8, true>};
            for (const void* k : fk)
                if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) != hipSuccess) return bail("hipFuncSetAttribute", e);
        }
    }
#undef ALLOC
    if (p->fused) {
        const int big = 150 * 1024;
#define P3D_W_KERNELS(LT) (const void*)dwt2_tile_kernel<c32, 32, LT>, (const void*)dwt2_tile_kernel<c32, 16, LT>, (const void*)dwt2_tile_kernel<float, 32, LT>, \
                         (const void*)dwt2_tile_kernel<float, 16, LT>, (const void*)idwt2_tile_kernel<c32, 32, LT>, (const void*)idwt2_tile_kernel<c32, 16, LT>, \
                         (const void*)idwt2_tile_kernel<float, 32, LT>, (const void*)idwt2_tile_kernel<float, 16, LT>
        const void* kernels[] = {P3D_W_KERNELS(0), P3D_W_KERNELS(4), P3D_W_KERNELS(8)};
#undef P3D_W_KERNELS
        for (const void* k : kernels)
            if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, big)) != hipSuccess) return bail("hipFuncSetAttribute", e);
    }
    *out = p;
    return P3D_OK;
}

extern "C" int p3d_wavelet_info(p3d_wplan* p, int* nlev, int64_t* ncoef, int32_t* shapes /* [(nlev+1)*2]: cA, then levels coarse -> fine */)
{
    if (!p) return wfail(P3D_ERR_INVALID, "NULL plan");
    if (nlev) *nlev = p->nlev;
    if (ncoef) *ncoef = (int64_t)p->ncoef;
    if (shapes) {
        shapes[0] = p->h[p->nlev]; shapes[1] = p->w[p->nlev];
        for (int l = p->nlev, i = 1; l >= 1; --l, ++i) { shapes[2 * i] = p->h[l]; shapes[2 * i + 1] = p->w[l]; }
    }
    return P3D_OK;
}

// ---- transforms on device buffers ------------------------------------------------------------------------------------------
// The work buffers are allocated for complex64; the real instantiation (float32 cubes with real thresholds: half the bytes)
// uses the same buffers with the same element counts.
template <typename T> static T* as(c32* p) { return reinterpret_cast<T*>(p); }

// feed (nil x nxl per slice) -> coefficient vectors (cA, details coarse -> fine).  With `th` (tau != nullptr) the details are
// thresholded as they are produced (threshold_wavelet, POCS.py:105-166) -- the approximation is never touched (POCS.py:586-587).
template <typename T> static int tile_of(const p3d_wplan* p) { return sizeof(T) == sizeof(float) ? p->tile_r : p->tile_c; }

template <typename T> static int coarse_level(const p3d_wplan* p) { return sizeof(T) == sizeof(float) ? p->lc_r : p->lc_c; }

// levels LC .. nlev of every slice in one workgroup each: analysis (approx[LC-1] -> thresholded details, cA) and / or synthesis
// (-> rec[LC-1])
template <typename T>
static int w_coarse(p3d_wplan* p, int ns, const Thresh* th, bool fwd, bool inv)
{
    const int lc = coarse_level<T>(p);
    CoarseArgs a{};
    a.in = p->approx[lc - 1]; a.in_slice = (size_t)p->h[lc - 1] * p->w[lc - 1];
    a.coef = p->coef; a.coef_slice = p->ncoef;
    a.rec = p->rec[lc - 1]; a.rec_slice = (size_t)p->rh[lc - 1] * p->rw[lc - 1];
    for (int l = 0; l <= p->nlev; ++l) { a.h[l] = p->h[l]; a.w[l] = p->w[l]; a.doff[l] = p->doff[l]; }
    a.nlev = p->nlev; a.LC = lc; a.do_fwd = fwd ? 1 : 0; a.do_inv = inv ? 1 : 0; a.ns = ns;
    a.x_elems = sizeof(T) == sizeof(float) ? p->cx_r : p->cx_c;
    const size_t lds = sizeof(T) == sizeof(float) ? p->clds_r : p->clds_c;
    Thresh t{nullptr, 0, 0, 0, 0, 0, -1, -1, nullptr};
    if (th) t = *th;
    t.done = p->loop_done;
    const int lt = p->f.len == 8 || p->f.len == 4 ? p->f.len : 0;
    if (lt == 8) wcoarse_kernel<T, 8><<<ns, COARSE_THREADS, lds, p->stream>>>(a, p->f, t);
    else if (lt == 4) wcoarse_kernel<T, 4><<<ns, COARSE_THREADS, lds, p->stream>>>(a, p->f, t);
    else wcoarse_kernel<T, 0><<<ns, COARSE_THREADS, lds, p->stream>>>(a, p->f, t);
    W_TRY(hipGetLastError());
    return P3D_OK;
}

// fuse_inverse: the coarse kernel also runs its synthesis (the loop; w_inverse_fused is then told so)
// l_from / l_to: only the levels l_from ... min(l_to, nlev) (the loop with the level-1 kernel: level 1 and the rest apart);
// det1 / det1_slice: where the level-1 details go (nullptr: into the coefficient vector)
template <typename T>
static int w_forward_fused(p3d_wplan* p, int ns, const Thresh* th, bool fuse_inverse = false, int l_from = 1, int l_to = 1 << 20, T* det1 = nullptr,
                           size_t det1_slice = 0)
{
    const int tile = tile_of<T>(p), ns8 = (ns + 7) / 8 * 8;
    const size_t lds = tile_lds(tile, p->f.len, sizeof(T));
    T* coef = as<T>(p->coef);
    const int lc = coarse_level<T>(p), last_tile_level = std::min(lc ? lc - 1 : p->nlev, l_to);
    for (int l = l_from; l <= last_tile_level; ++l) {
        const T* src = l == 1 ? as<T>(p->feed) : as<T>(p->approx[l - 1]);
        const int H = p->h[l - 1], W = p->w[l - 1], Ho = p->h[l], Wo = p->w[l];
        T* cA = l == p->nlev ? coef : as<T>(p->approx[l]);
        const size_t cA_slice = l == p->nlev ? p->ncoef : (size_t)Ho * Wo;
        Thresh t{nullptr, 0, 0, 0, 0, 0, -1, -1};
        if (th) { t = *th; t.lvl = p->nlev - l; }
        const int tx = (Wo + tile - 1) / tile, ty = (Ho + tile - 1) / tile;
        T* const det = (l == 1 && det1) ? det1 : coef + p->doff[l];
        const size_t det_slice = (l == 1 && det1) ? det1_slice : p->ncoef;
#define P3D_W_DWT(TL, LT) dwt2_tile_kernel<T, TL, LT><<<tx * ty * ns8, WTILE_NT<T, TL>, lds, p->stream>>>(src, (size_t)H * W, H, W, cA, cA_slice, det, det_slice, Ho, Wo, p->f, tx, tx * ty, ns, t)
        const int lt = p->f.len == 8 || p->f.len == 4 ? p->f.len : 0;   // db4 / sym4 and db2 have kernels of their own
        if (tile == 32) { if (lt == 8) P3D_W_DWT(32, 8); else if (lt == 4) P3D_W_DWT(32, 4); else P3D_W_DWT(32, 0); }
        else { if (lt == 8) P3D_W_DWT(16, 8); else if (lt == 4) P3D_W_DWT(16, 4); else P3D_W_DWT(16, 0); }
#undef P3D_W_DWT
    }
    W_TRY(hipGetLastError());
    if (lc && lc <= l_to) return w_coarse<T>(p, ns, th, true, fuse_inverse);
    return P3D_OK;
}

// `u`: what to do with the level-0 reconstruction (nullptr: store it in rec[0])
// l_from >= l >= l_to: only those levels (defaults: all of them); det1: where the level-1 details are (nullptr: the coefficient vector)
template <typename T>
static int w_inverse_fused(p3d_wplan* p, int ns, const Update* u, bool coarse_done = false, int l_from = 1 << 20, int l_to = 1, const T* det1 = nullptr,
                           size_t det1_slice = 0)
{
    const int tile = tile_of<T>(p), ns8 = (ns + 7) / 8 * 8;
    const size_t lds = tile_lds(tile, p->f.len, sizeof(T));
    T* coef = as<T>(p->coef);
    const int lc = coarse_level<T>(p);
    if (lc && !coarse_done && lc <= l_from) {
        const int rc = w_coarse<T>(p, ns, nullptr, false, true);
        if (rc) return rc;
    }
    for (int l = std::min(lc ? lc - 1 : p->nlev, l_from); l >= l_to; --l) {
        const int Ho = p->h[l], Wo = p->w[l], RH = p->rh[l - 1], RW = p->rw[l - 1];
        const T* a = l == p->nlev ? coef : as<T>(p->rec[l]);
        const size_t a_ld = l == p->nlev ? (size_t)Wo : (size_t)p->rw[l];
        const size_t a_slice = l == p->nlev ? p->ncoef : (size_t)p->rh[l] * p->rw[l];
        Update none{};
        none.done = p->loop_done;
        const Update& up = (l == 1 && u) ? *u : none;
        // with the re-insertion fused only the nil x nxl crop of the level-0 reconstruction is needed
        const int OHt = up.enabled ? up.n1 : RH, OWt = up.enabled ? up.n2 : RW;
        const int tx = (OWt + 2 * tile - 1) / (2 * tile), ty = (OHt + 2 * tile - 1) / (2 * tile);
        const T* const det = (l == 1 && det1) ? det1 : coef + p->doff[l];
        const size_t det_slice = (l == 1 && det1) ? det1_slice : p->ncoef;
#define P3D_W_IDWT(TL, LT) idwt2_tile_kernel<T, TL, LT><<<tx * ty * ns8, WTILE_NT<T, TL>, lds, p->stream>>>(a, a_ld, a_slice, det, det_slice, Ho, Wo, as<T>(p->rec[l - 1]), \
                                                                                     (size_t)RH * RW, RH, RW, p->f, tx, tx * ty, ns, up)
        const int lt = p->f.len == 8 || p->f.len == 4 ? p->f.len : 0;
        if (tile == 32) { if (lt == 8) P3D_W_IDWT(32, 8); else if (lt == 4) P3D_W_IDWT(32, 4); else P3D_W_IDWT(32, 0); }
        else { if (lt == 8) P3D_W_IDWT(16, 8); else if (lt == 4) P3D_W_IDWT(16, 4); else P3D_W_IDWT(16, 0); }
#undef P3D_W_IDWT
    }
    W_TRY(hipGetLastError());
    return P3D_OK;
}

template <typename T>
static int w_forward(p3d_wplan* p, int ns, const Thresh* th)
{
    if (p->fused) return w_forward_fused<T>(p, ns, th);
    const dim3 blk(256);
    const Thresh none{nullptr, 0, 0, 0, 0, 0, -1, -1};
    T *lo = as<T>(p->lo), *hi = as<T>(p->hi), *coef = as<T>(p->coef);
    for (int l = 1; l <= p->nlev; ++l) {
        const T* src = l == 1 ? as<T>(p->feed) : as<T>(p->approx[l - 1]);
        const int H = p->h[l - 1], W = p->w[l - 1], Ho = p->h[l], Wo = p->w[l];
        const size_t cnt = (size_t)Ho * Wo;
        // along axis 1 (rows are contiguous): (H x W) -> lo, hi (H x Wo)
        dwt_axis_kernel<T><<<dim3(blocks_for((size_t)H * Wo), ns), blk, 0, p->stream>>>(src, lo, hi, p->f, H, W, Wo, (size_t)W, 1, (size_t)H * W, (size_t)Wo, 1,
                                                                                    (size_t)H * Wo, (size_t)H * Wo, none);
        // along axis 0 (lines = columns): lo -> (aa, da = cH), hi -> (ad = cV, dd = cD), each (Ho x Wo)
        T* cA = l == p->nlev ? coef : as<T>(p->approx[l]);
        const size_t cA_slice = l == p->nlev ? p->ncoef : cnt;
        T* det = coef + p->doff[l];
        Thresh t1 = none, t2 = none;
        if (th) {
            t1 = t2 = *th;
            t1.lvl = t2.lvl = p->nlev - l;  // PyWavelets' order: coarsest level first
            t1.z_lo = -1; t1.z_hi = 0;
            t2.z_lo = 1; t2.z_hi = 2;
        }
        dwt_axis_kernel<T><<<dim3(blocks_for((size_t)Wo * Ho), ns), blk, 0, p->stream>>>(lo, cA, det, p->f, Wo, H, Ho, 1, (size_t)Wo, (size_t)H * Wo, 1, (size_t)Wo,
                                                                                     cA_slice, p->ncoef, t1);
        dwt_axis_kernel<T><<<dim3(blocks_for((size_t)Wo * Ho), ns), blk, 0, p->stream>>>(hi, det + cnt, det + 2 * cnt, p->f, Wo, H, Ho, 1, (size_t)Wo, (size_t)H * Wo,
                                                                                     1, (size_t)Wo, p->ncoef, p->ncoef, t2);
    }
    W_TRY(hipGetLastError());
    return P3D_OK;
}

// coefficient vectors -> rec[0] (rh[0] x rw[0] per slice; its top-left nil x nxl block is the slice)
template <typename T>
static int w_inverse(p3d_wplan* p, int ns)
{
    if (p->fused) return w_inverse_fused<T>(p, ns, nullptr);
    const dim3 blk(256);
    T *lo = as<T>(p->lo), *hi = as<T>(p->hi), *coef = as<T>(p->coef);
    for (int l = p->nlev; l >= 1; --l) {
        const int Ho = p->h[l], Wo = p->w[l];            // valid extent of the level-l arrays
        const int RH = p->rh[l - 1], RW = p->rw[l - 1];  // shape of the reconstruction of level l-1
        const size_t cnt = (size_t)Ho * Wo;
        // approximation of level l: cA itself at the coarsest level, otherwise the reconstruction of level l (which may be one
        // row / column larger than Ho x Wo: the extra samples are ignored, as in pywt.waverec2)
        const T* a = l == p->nlev ? coef : as<T>(p->rec[l]);
        const size_t a_ld = l == p->nlev ? (size_t)Wo : (size_t)p->rw[l];
        const size_t a_slice = l == p->nlev ? p->ncoef : (size_t)p->rh[l] * p->rw[l];
        const T* det = coef + p->doff[l];
        // undo axis 0: (a, cH) -> lo (RH x Wo);  (cV, cD) -> hi (RH x Wo)
        idwt_axis_kernel<T><<<dim3(blocks_for((size_t)Wo * RH), ns), blk, 0, p->stream>>>(a, det, lo, p->f, Wo, Ho, RH, 1, a_ld, a_slice, 1, (size_t)Wo, p->ncoef, 1,
                                                                                      (size_t)Wo, (size_t)RH * Wo);
        idwt_axis_kernel<T><<<dim3(blocks_for((size_t)Wo * RH), ns), blk, 0, p->stream>>>(det + cnt, det + 2 * cnt, hi, p->f, Wo, Ho, RH, 1, (size_t)Wo, p->ncoef, 1,
                                                                                      (size_t)Wo, p->ncoef, 1, (size_t)Wo, (size_t)RH * Wo);
        // undo axis 1: (lo, hi) (RH x Wo) -> rec[l-1] (RH x RW)
        idwt_axis_kernel<T><<<dim3(blocks_for((size_t)RH * RW), ns), blk, 0, p->stream>>>(lo, hi, as<T>(p->rec[l - 1]), p->f, RH, Wo, RW, (size_t)Wo, 1, (size_t)RH * Wo,
                                                                                      (size_t)Wo, 1, (size_t)RH * Wo, (size_t)RW, 1, (size_t)RH * RW);
    }
    W_TRY(hipGetLastError());
    return P3D_OK;
}

// a pointer into the memory of the plan's own device (the entry points take host or device pointers)
static bool on_plan_device(const p3d_wplan* p, const void* ptr)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();   // ordinary host memory
        return false;
    }
    return at.type == hipMemoryTypeDevice && at.device == p->device;
}

static int w_check(p3d_wplan* p, int nslices, int dtype)
{
    if (!p) return wfail(P3D_ERR_INVALID, "NULL plan");
    if (nslices < 1 || nslices > p->max_slices) return wfail(P3D_ERR_INVALID, "nslices = %d outside 1..max_slices (%d)", nslices, p->max_slices);
    if (dtype != P3D_C64 && dtype != P3D_F32) return wfail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    W_TRY(hipSetDevice(p->device));
    return P3D_OK;
}

template <typename T>
static int w_stats(p3d_wplan* p, int dtype, int nslices)
{
    wupdate_kernel<T><<<dim3(p->nil < 256 ? p->nil : 256, nslices), 256, 0, p->stream>>>(nullptr, 0, 0, as<T>(p->feed), p->cur_x, dtype, nullptr, nullptr, p->sums, 0, 0, 0, 1.0f,
                                                                                p->nil, p->nxl, nullptr, 0);
    int rc = w_forward<T>(p, nslices, nullptr);
    if (rc) return rc;
    if (p->nlev > WSTAT_MAX_LEVELS) return wfail(P3D_ERR_UNSUPPORTED, "more than %d levels", WSTAT_MAX_LEVELS);
    WStatLevels lv{};
    for (int l = p->nlev, i = 0; l >= 1; --l, ++i) { lv.off[i] = p->doff[l]; lv.count[i] = (size_t)p->h[l] * p->w[l]; }
    wstats_kernel<T><<<dim3(p->nlev, nslices, 3), 256, 0, p->stream>>>(as<T>(p->coef), p->ncoef, lv, p->stats, p->nlev);
    W_TRY(hipGetLastError());
    return P3D_OK;
}

template <typename T>
static int w_loop(p3d_wplan* p, int dtype, int nslices, const p3d_pocs_params* prm)
{
    const int niter = prm->niter;
    const bool early = prm->eps > 0.0, adaptive = prm->version == P3D_VER_ADAPTIVE;
    // early exit on the fused kernels: a finished slice's arrays are left alone by EVERY pass from the iteration after its last one, and its iterate is
    // rebuilt from them after the loop (below)
    struct LoopDone { p3d_wplan* p; ~LoopDone() { p->loop_done = nullptr; } } loop_done_guard{p};
    p->loop_done = (early && p->fused) ? p->done : nullptr;
    const dim3 ugrid(p->nil < 256 ? p->nil : 256, nslices);   // (a block walks over rows)
    wupdate_kernel<T><<<ugrid, 256, 0, p->stream>>>(nullptr, 0, 0, as<T>(p->feed), p->cur_x, dtype, p->mask, p->cur_out, p->sums, 0, adaptive ? 1 : 0, 0,
                                                   (float)prm->alpha, p->nil, p->nxl, p->done, 0);
    const bool l1fuse = p->fused && (sizeof(T) == sizeof(float) ? p->l1fuse_r : p->l1fuse_c);
    // level-1 details of iteration k live in buffer k % 2 (the level-1 kernel reads one while it writes the other)
    const size_t cnt1 = (size_t)p->h[1] * p->w[1];
    T* const det1[2] = {as<T>(p->coef) + p->doff[1], as<T>(p->det1_alt)};
    const size_t det1_slice[2] = {p->ncoef, 3 * cnt1};
    if (l1fuse) {
        const Thresh th0{p->tau, niter, 0, p->nlev, 0, prm->thresh_op, -1, -1, p->loop_done};
        const int rc = w_forward_fused<T>(p, nslices, &th0, false, 1, 1, det1[0], det1_slice[0]);
        if (rc) return rc;
    }
    for (int k = 0; k < niter; ++k) {
        const bool last = k + 1 == niter;
        const Thresh th{p->tau, niter, k, p->nlev, 0, prm->thresh_op, -1, -1, p->loop_done};
        int rc = l1fuse ? w_forward_fused<T>(p, nslices, &th, true, 2) : (p->fused ? w_forward_fused<T>(p, nslices, &th, true) : w_forward<T>(p, nslices, &th));
        if (rc) return rc;
        if (p->fused) {
            Update u{};
            u.enabled = 1; u.feed = p->feed; u.x = p->cur_x; u.dtype = dtype; u.mask = p->mask; u.out = p->cur_out;
            u.mask_bits = p->mask_binary ? p->mask_bits : nullptr; u.mask_words = p->mask_words;
            u.sums = p->sums + (size_t)(k + 1) * nslices;
            // (early exit: the iterate of a slice that converges is rebuilt once, after the loop, from the coefficients it stopped at -- storing every
            // iterate of every slice instead cost 7 % of a configs[3] job)
            u.adaptive = (adaptive && !last) ? 1 : 0; u.write_out = last ? 1 : 0; u.zero_fill = last ? 1 : 0;
            u.alpha = (float)prm->alpha; u.n1 = p->nil; u.n2 = p->nxl; u.done = p->done;
            if (!l1fuse) {
                if ((rc = w_inverse_fused<T>(p, nslices, &u, true))) return rc;
            } else {
                if ((rc = w_inverse_fused<T>(p, nslices, nullptr, true, 1 << 20, 2))) return rc;           // ... down to rec[1]
                if (last) {
                    if ((rc = w_inverse_fused<T>(p, nslices, &u, true, 1, 1, det1[k & 1], det1_slice[k & 1]))) return rc;
                } else {   // level-1 synthesis of this iteration + re-insertion + level-1 analysis of the next one
                    Thresh tn = th;
                    tn.iter = k + 1; tn.lvl = p->nlev - 1;
                    const int Ho = p->h[1], Wo = p->w[1], tx = (Wo + 31) / 32, ty = (Ho + 31) / 32, ns8 = (nslices + 7) / 8 * 8;
                    const size_t lds = wfuse1_lds_elems<32>(p->f.len) * sizeof(T);
                    // (both detail buffers are addressed with ONE slice stride inside the kernel: hand the larger-stride buffer its own launch form)
                    const T* din = det1[k & 1];
                    T* dout = det1[(k + 1) & 1];
#define P3D_W_FUSE1(LT, MB) wfuse1_kernel<T, 32, LT, MB><<<tx * ty * ns8, WFUSE1_NT<T>, lds, p->stream>>>(as<T>(p->rec[1]), (size_t)p->rw[1], (size_t)p->rh[1] * p->rw[1], din, det1_slice[k & 1], \
                                                                                       dout, det1_slice[(k + 1) & 1], Ho, Wo, as<T>(p->approx[1]), cnt1, p->f, tx, tx * ty, nslices, u, tn)
                    const int lt = p->f.len == 8 || p->f.len == 4 ? p->f.len : 0;
                    const bool mb = u.mask_bits != nullptr;
                    if (lt == 8) { if (mb) P3D_W_FUSE1(8, true); else P3D_W_FUSE1(8, false); }
                    else if (lt == 4) { if (mb) P3D_W_FUSE1(4, true); else P3D_W_FUSE1(4, false); }
                    else P3D_W_FUSE1(0, false);
#undef P3D_W_FUSE1
                }
            }
        } else {
            if ((rc = w_inverse<T>(p, nslices))) return rc;
            wupdate_kernel<T><<<ugrid, 256, 0, p->stream>>>(as<T>(p->rec[0]), (size_t)p->rw[0], (size_t)p->rh[0] * p->rw[0], as<T>(p->feed), p->cur_x, dtype, p->mask,
                                                           p->cur_out, p->sums + (size_t)(k + 1) * nslices, 1, (adaptive && !last) ? 1 : 0, (early || last) ? 1 : 0,
                                                           (float)prm->alpha, p->nil, p->nxl, p->done, last ? 1 : 0);
        }
        if (early) wconv_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
    }
    if (early && p->fused) {
        Update uf{};
        uf.enabled = 1; uf.feed = p->feed; uf.x = p->cur_x; uf.dtype = dtype; uf.mask = p->mask; uf.out = p->cur_out;
        uf.sums = p->sums;   // (not added to: the slices are finished)
        uf.write_out = 1; uf.alpha = (float)prm->alpha; uf.n1 = p->nil; uf.n2 = p->nxl; uf.done = p->done;
        uf.only_below = niter;
        for (int par = 0; par < (l1fuse ? 2 : 1); ++par) {
            uf.only_parity = l1fuse ? par + 1 : 3;
            const int rc = l1fuse ? w_inverse_fused<T>(p, nslices, &uf, true, 1, 1, det1[par], det1_slice[par]) : w_inverse_fused<T>(p, nslices, &uf, true, 1, 1);
            if (rc) return rc;
        }
    }
    W_TRY(hipGetLastError());
    return P3D_OK;
}

extern "C" {

// test hooks: multilevel decomposition of complex64 slices into coefficient vectors [nslices][ncoef] and back
int p3d_wavedec2_c64(p3d_wplan* p, const void* x, void* coef, int nslices)
{
    int rc = w_check(p, nslices, P3D_C64);
    if (rc) return rc;
    if (!x || !coef) return wfail(P3D_ERR_INVALID, "NULL buffer");
    W_TRY(hipMemcpy(p->feed, x, sizeof(c32) * p->per() * nslices, hipMemcpyHostToDevice));
    if ((rc = w_forward<c32>(p, nslices, nullptr))) return rc;
    W_TRY(hipStreamSynchronize(p->stream));
    W_TRY(hipMemcpy(coef, p->coef, sizeof(c32) * p->ncoef * nslices, hipMemcpyDeviceToHost));
    return P3D_OK;
}

int p3d_waverec2_c64(p3d_wplan* p, const void* coef, void* x, int nslices)
{
    int rc = w_check(p, nslices, P3D_C64);
    if (rc) return rc;
    if (!x || !coef) return wfail(P3D_ERR_INVALID, "NULL buffer");
    W_TRY(hipMemcpy(p->coef, coef, sizeof(c32) * p->ncoef * nslices, hipMemcpyHostToDevice));
    if ((rc = w_inverse<c32>(p, nslices))) return rc;
    // crop to the slice shape (POCS.py:513, 609)
    for (int s = 0; s < nslices; ++s)
        W_TRY(hipMemcpy2DAsync(p->feed + (size_t)s * p->per(), sizeof(c32) * p->nxl, p->rec[0] + (size_t)s * p->rh[0] * p->rw[0], sizeof(c32) * p->rw[0],
                               sizeof(c32) * p->nxl, (size_t)p->nil, hipMemcpyDeviceToDevice, p->stream));
    W_TRY(hipStreamSynchronize(p->stream));
    W_TRY(hipMemcpy(x, p->feed, sizeof(c32) * p->per() * nslices, hipMemcpyDeviceToHost));
    return P3D_OK;
}

// statistics of the detail arrays of transform(x) for the schedule (POCS.py:253-254, 281): stats [nslices][nlev][3][4] doubles =
// Re, Im of the lexicographic max, max |d|, min |d|; levels coarse -> fine (PyWavelets' order)
int p3d_wavelet_stats(p3d_wplan* p, const void* x, int dtype, int nslices, double* stats)
{
    int rc = w_check(p, nslices, dtype);
    if (rc) return rc;
    if (!x || !stats) return wfail(P3D_ERR_INVALID, "NULL buffer");
    const size_t esz = dtype == P3D_C64 ? sizeof(c32) : sizeof(float);
    if (on_plan_device(p, x)) {
        p->cur_x = x;
    } else {
        W_TRY(hipMemcpyAsync(p->st_x, x, esz * p->per() * nslices, hipMemcpyDefault, p->stream));   // (ordered with the plan's stream: p3d_wavelet_run)
        p->cur_x = p->st_x;
    }
    if (p->sums_cap < (size_t)nslices) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr; p->sums_cap = 0;
        W_TRY(hipMalloc((void**)&p->sums, sizeof(double) * 2 * p->max_slices));
        p->sums_cap = 2 * (size_t)p->max_slices;
    }
    W_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nslices, p->stream));
    if ((rc = dtype == P3D_F32 ? w_stats<float>(p, dtype, nslices) : w_stats<c32>(p, dtype, nslices))) return rc;
    std::vector<float> host((size_t)nslices * p->nlev * 12);
    W_TRY(hipMemcpyAsync(host.data(), p->stats, sizeof(float) * host.size(), hipMemcpyDeviceToHost, p->stream));
    W_TRY(hipStreamSynchronize(p->stream));
    for (size_t i = 0; i < host.size(); ++i) stats[i] = host[i];
    return P3D_OK;
}

// the loop (POCS.py:549-632 with the WAVELET branches); tau: HOST [nslices][niter][nlev][3][2] doubles, levels coarse -> fine
int p3d_wavelet_run(p3d_wplan* p, const void* x, int dtype, const float* mask, const double* tau, const uint8_t* active, const p3d_pocs_params* prm,
                    void* out, int nslices, int32_t* niter_done, double* sums, double* elapsed_ms)
{
    int rc = w_check(p, nslices, dtype);
    if (rc) return rc;
    if (!x || !mask || !tau || !prm || !out) return wfail(P3D_ERR_INVALID, "NULL argument");
    if (prm->niter < 1) return wfail(P3D_ERR_INVALID, "niter must be >= 1");
    if (prm->thresh_op < P3D_OP_HARD || prm->thresh_op > P3D_OP_GARROTE)
        return wfail(P3D_ERR_UNSUPPORTED, "thresh_op %d is not implemented for the wavelet transform", prm->thresh_op);
    const int niter = prm->niter;
    const size_t esz = dtype == P3D_C64 ? sizeof(c32) : sizeof(float);
    const size_t ntau = (size_t)nslices * niter * p->nlev * 3, nsum = (size_t)(niter + 1) * nslices;
    if (p->tau_cap < ntau) {
        if (p->tau) hipFree(p->tau);
        p->tau = nullptr; p->tau_cap = 0;
        W_TRY(hipMalloc((void**)&p->tau, sizeof(c32) * ntau));
        p->tau_cap = ntau;
    }
    if (p->sums_cap < nsum) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr; p->sums_cap = 0;
        W_TRY(hipMalloc((void**)&p->sums, sizeof(double) * nsum));
        p->sums_cap = nsum;
    }
    std::vector<c32> tau_f(ntau);
    bool real_tau = true;
    for (size_t i = 0; i < ntau; ++i) {
        tau_f[i] = p3d::tau_for_device(tau[2 * i], tau[2 * i + 1], prm->thresh_op == P3D_OP_HARD);
        real_tau = real_tau && tau[2 * i + 1] == 0.0;
    }
    // float32 cubes with real thresholds stay real through the whole loop (what PyWavelets does for real input)
    const bool real_path = dtype == P3D_F32 && real_tau;
    std::vector<int> done_h(nslices, 0);
    if (active) for (int s = 0; s < nslices; ++s) done_h[s] = active[s] ? 0 : -1;
    const size_t cube_bytes = esz * p->per() * nslices;
    if (on_plan_device(p, x)) {
        p->cur_x = x;
    } else {
        W_TRY(hipMemcpyAsync(p->st_x, x, cube_bytes, hipMemcpyDefault, p->stream));
        p->cur_x = p->st_x;
    }
    // (the loop reads the observed cube in every iteration: a result buffer that overlaps it goes through the staging buffer)
    const char* const xb = static_cast<const char*>(x);
    char* const ob = static_cast<char*>(out);
    const bool direct_out = on_plan_device(p, out) && (ob + cube_bytes <= xb || xb + cube_bytes <= ob);
    p->cur_out = direct_out ? out : p->st_out;
    // the mask may be a device pointer: a device-to-device hipMemcpy runs on the null stream, need not have finished when it returns, and the plan's
    // (non-blocking) stream does not wait for it -- every copy of this entry point goes onto the plan's stream
    W_TRY(hipMemcpyAsync(p->mask, mask, sizeof(float) * p->per(), hipMemcpyDefault, p->stream));
    p->mask_binary = 0;
    if ((p->l1fuse_c || p->l1fuse_r) && !getenv("P3D_WAVELET_NO_MASK_BITS")) {   // wfuse1_kernel reads a 0 / 1 mask as bits
        int* flag = reinterpret_cast<int*>(p->mask_bits + (size_t)p->nil * p->mask_words);
        const int one = 1, nw = p->nil * p->mask_words;
        W_TRY(hipMemcpyAsync(flag, &one, sizeof(int), hipMemcpyHostToDevice, p->stream));
        wmask_pack_kernel<<<(nw + 255) / 256, 256, 0, p->stream>>>(p->mask, p->mask_bits, p->nil, p->nxl, p->mask_words, flag);
        W_TRY(hipMemcpyAsync(&p->mask_binary, flag, sizeof(int), hipMemcpyDeviceToHost, p->stream));
        W_TRY(hipStreamSynchronize(p->stream));
    }
    W_TRY(hipMemcpyAsync(p->tau, tau_f.data(), sizeof(c32) * ntau, hipMemcpyHostToDevice, p->stream));
    W_TRY(hipMemcpyAsync(p->done, done_h.data(), sizeof(int) * nslices, hipMemcpyHostToDevice, p->stream));
    W_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nsum, p->stream));
    W_TRY(hipEventRecord(p->ev0, p->stream));
    if ((rc = real_path ? w_loop<float>(p, dtype, nslices, prm) : w_loop<c32>(p, dtype, nslices, prm))) return rc;
    W_TRY(hipEventRecord(p->ev1, p->stream));
    W_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
    if (sums) W_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
    W_TRY(hipStreamSynchronize(p->stream));
    if (!direct_out) {
        W_TRY(hipMemcpyAsync(out, p->st_out, cube_bytes, hipMemcpyDefault, p->stream));
        W_TRY(hipStreamSynchronize(p->stream));
    }
    if (niter_done) for (int s = 0; s < nslices; ++s) niter_done[s] = done_h[s] < 0 ? 0 : (done_h[s] > 0 ? done_h[s] : niter);
    if (elapsed_ms) {
        float ms = 0.f;
        W_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        *elapsed_ms = ms;
    }
    return P3D_OK;
}

}  // extern "C"
