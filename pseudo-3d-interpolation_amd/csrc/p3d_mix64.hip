// p3d_mix64.hip -- the two fused passes of the DOUBLE-PRECISION POCS loop (p3d_f64.hip) on the register-resident mixed-radix engine
// (p3d_mix_engine.hpp with complex128 elements).
//
// p3d_f64.hip's own passes (col64_kernel / row64_kernel) keep a tile of lines in LDS and run one butterfly per thread and pass with run-time
// factors: 16 barriers per tile, 0.165 of the pass's 56 B/point roofline (profiles/r04_f64_fused.txt).  Here a line of N = R0 R1 R2 points is
// held by N / PPT threads, 16-20 points each in registers, compile-time radices, ONE exchange through LDS between two passes -- the
// float32 engine of p3d_mix.hpp, instantiated for complex128 and for powers of two as well (p3d_mix64_plans.inc: tools/gen_mix_plans.py --f64
// --all-smooth: the powers of two 64 ... 4096 and every 7-smooth length 96 ... 4096, 8 points per thread).
// Same semantics as col64_kernel / row64_kernel, same row-major complex128 work buffer [slice][n1][n2], same partial-sum layouts; p3d_f64.hip
// picks these passes per axis where a plan exists (P3D_NO_MIX64=1: never).  The three fused passes of the double-precision SHEARLET loop
// (p3d_shearlet64.hip) live here too: shear_col_kernel / spread_row_kernel / gather_row_kernel, same engine, same two register layouts.
//
// Compiled once per part of the plan list (-DP3D_MIX64_PART=k, eight translation units); part 0 holds the registry.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <map>

#include "p3d.h"
#include "p3d_mix64.hpp"
#include "p3d_mix_engine.hpp"

#ifndef P3D_MIX64_PART
#error "compile with -DP3D_MIX64_PART=<k>"
#endif

namespace p3d {
namespace mix64 {

using mix::c64d;
static_assert(M64_COL_ITER == COL_ITER && M64_COL_STATS == COL_STATS && M64_COL_FWD == COL_FWD && M64_ROW_FIRST == ROW_FIRST && M64_ROW_MID == ROW_MID &&
              M64_ROW_LAST == ROW_LAST, "mode numbers");

namespace {

// ---- thresholds in double (threshold_operator.py:9-112; complex tau: lexicographic comparisons, as p3d_f64.hip's shrink64) -----------------
__device__ __forceinline__ c64d shrink(c64d X, c64d tau, int op)
{
    const double m = hypot(X.x, X.y);
    if (op == 0) {
        const bool below = m < tau.x || (m == tau.x && 0.0 < tau.y);
        return below ? c64d{0.0, 0.0} : X;
    }
    if (m == 0.0) return c64d{0.0, 0.0};
    double gr, gi;
    if (op == 1) {
        gr = 1.0 - tau.x / m;
        gi = -tau.y / m;
    } else {
        const double m2 = m * m;
        gr = 1.0 - (tau.x * tau.x - tau.y * tau.y) / m2;
        gi = -(2.0 * tau.x * tau.y) / m2;
    }
    const bool keep = gr > 0.0 || (gr == 0.0 && gi >= 0.0);
    return keep ? c64d{X.x * gr - X.y * gi, X.x * gi + X.y * gr} : c64d{0.0, 0.0};
}

__device__ __forceinline__ c64d load_x(const void* x, int dtype, size_t g)
{
    switch (dtype) {
        case P3D_C128: return reinterpret_cast<const c64d*>(x)[g];
        case P3D_F64: return c64d{reinterpret_cast<const double*>(x)[g], 0.0};
        case P3D_C64: { const float2 t = reinterpret_cast<const float2*>(x)[g]; return c64d{(double)t.x, (double)t.y}; }
        default: return c64d{(double)reinterpret_cast<const float*>(x)[g], 0.0};
    }
}
__device__ __forceinline__ void store_out(void* out, int dtype, size_t g, c64d v)
{
    switch (dtype) {
        case P3D_C128: reinterpret_cast<c64d*>(out)[g] = v; break;
        case P3D_F64: reinterpret_cast<double*>(out)[g] = v.x; break;
        case P3D_C64: reinterpret_cast<float2*>(out)[g] = float2{(float)v.x, (float)v.y}; break;
        default: reinterpret_cast<float*>(out)[g] = (float)v.x; break;
    }
}

// ---- column pass: a tile of T columns of one slice; modes as col64_kernel ---------------------------------------------------------------------------
template <class PL>
__global__ __launch_bounds__(PL::COLT* PL::TMAX) void col_kernel(const ColArgs64 a, int mode)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, TMAX = PL::TMAX, T = PL::COLT, THREADS = T * TMAX;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[((THREADS + 63) / 64) * 5];
    c64d* data = reinterpret_cast<c64d*>(smem_raw);
    const c64d* const twl = a.tab;   // the twiddle tables stay in memory (L1 / L2): 16 bytes per entry, ~1.03 N entries -- in LDS they cost the second workgroup per CU
    const int tid = threadIdx.x, c_lo = tid % T, tl = tid / T, s = blockIdx.y, tile = blockIdx.x;
    const int col = tile * T + c_lo;
    const bool valid = col < a.n2;
    if (a.done && a.done[s] != 0) return;
    c64d* const base = a.work + (size_t)s * N * a.n2 + (valid ? col : 0);
    const bool live_a = valid && tl < TPL_A, live_b = valid && tl < TPL_B;
    c64d v[VMAX];
#pragma unroll
    for (int q = 0; q < VMAX; ++q) v[q] = c64d{0.0, 0.0};
    {   // unconditional loads from clamped rows (a thread outside the layout re-reads row 0 ...), then the zeros where it does not take part
        const int r0 = tl < TPL_A ? tl : 0;
#pragma unroll
        for (int q = 0; q < PPT_A; ++q) v[q] = base[(size_t)(r0 + TPL_A * q) * a.n2];
        if (!live_a) {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) v[q] = c64d{0.0, 0.0};
        }
    }
    mix::line_fft<PL, FWD, T>(v, data + c_lo, twl, tl);   // layout A -> layout B
    if (mode == COL_STATS) {
        double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY, sq = 0.0;
        if (live_b) {
#pragma unroll
            for (int q = 0; q < PPT_B; ++q) {
                const double m = hypot(v[q].x, v[q].y);
                if (v[q].x > lr || (v[q].x == lr && v[q].y > li)) { lr = v[q].x; li = v[q].y; }
                mx = fmax(mx, m);
                mn = fmin(mn, m);
                sq += v[q].x * v[q].x + v[q].y * v[q].y;
            }
        }
        const int lane = tid & 63;
        const int nact = (THREADS - (tid & ~63)) < 64 ? (THREADS - (tid & ~63)) : 64;   // the last wavefront may be partial
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
            const double omx = __shfl_down(mx, o, 64), omn = __shfl_down(mn, o, 64), osq = __shfl_down(sq, o, 64);
            if (lane + o < nact) {
                if (orr > lr || (orr == lr && oi > li)) { lr = orr; li = oi; }
                mx = fmax(mx, omx);
                mn = fmin(mn, omn);
                sq += osq;
            }
        }
        if (lane == 0) { double* r = red + (tid >> 6) * 5; r[0] = lr; r[1] = li; r[2] = mx; r[3] = mn; r[4] = sq; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (THREADS + 63) / 64; ++w) {
                const double* r = red + w * 5;
                if (r[0] > lr || (r[0] == lr && r[1] > li)) { lr = r[0]; li = r[1]; }
                mx = fmax(mx, r[2]);
                mn = fmin(mn, r[3]);
                sq += r[4];
            }
            double* q = a.partial + ((size_t)s * gridDim.x + tile) * 8;
            q[0] = lr; q[1] = li; q[2] = mx; q[3] = mn; q[4] = sq;
        }
        return;
    }
    if (mode == COL_ITER) {
        const c64d t = a.tau[(size_t)s * a.niter + a.iter];
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) v[q] = shrink(v[q], t, a.op);
        if (a.nzflag != nullptr) {
            // sparse spectra (the premise of the method): a tile the threshold emptied is all zeros after the inverse transform too -- say so
            // instead of transforming and storing it; the row pass reads zeros for it
            bool any = false;
            if (live_b) {
#pragma unroll
                for (int q = 0; q < PPT_B; ++q) any = any || v[q].x != 0.0 || v[q].y != 0.0;
            }
            const int kept = __syncthreads_or(any ? 1 : 0);
            if (tid == 0) a.nzflag[(size_t)s * gridDim.x + tile] = kept ? 1 : 0;
            if (!kept) return;
        }
        mix::line_fft<PL, INV, T>(v, data + c_lo, twl, tl);   // layout B -> layout A
        if (live_a) {
            const double scale = 1.0 / N;
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) base[(size_t)(tl + TPL_A * q) * a.n2] = v[q] * scale;
        }
        return;
    }
    if (live_b) {   // COL_FWD
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) base[(size_t)(tl + TPL_B * q) * a.n2] = v[q];
    }
}

// ---- row pass: LB rows of one slice; modes as row64_kernel -----------------------------------------------------------------------------------------------
template <class PL>
__global__ __launch_bounds__(PL::ROWLB* PL::TMAX) void row_kernel(const RowArgs64 a, int mode)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, TMAX = PL::TMAX, LB = PL::ROWLB, THREADS = LB * TMAX;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[(THREADS + 63) / 64];
    __shared__ unsigned char nzl[N + 8];   // the slice's tile flags
    c64d* data = reinterpret_cast<c64d*>(smem_raw);
    const c64d* const twl = a.tab;
    const int tid = threadIdx.x, line = tid / TMAX, tl = tid % TMAX, s = blockIdx.y, row = blockIdx.x * LB + line;
    const bool valid = row < a.n1;
    const bool live_a = valid && tl < TPL_A, live_b = valid && tl < TPL_B;
    const int dn = a.done ? a.done[s] : 0;
    const size_t per = (size_t)a.n1 * N, rbase = (size_t)(valid ? row : 0) * N, g0 = (size_t)s * per + rbase;
    if (mode != ROW_FIRST && a.zero_fill && dn < 0 && live_a) {   // an empty slice is handed back untouched (zeros), POCS.py:515-521
#pragma unroll
        for (int q = 0; q < PPT_A; ++q) store_out(a.out, a.dtype, g0 + tl + TPL_A * q, c64d{0.0, 0.0});
    }
    if (dn != 0) {
        if (tid == 0) a.partial[(size_t)s * gridDim.x + blockIdx.x] = 0.0;
        return;
    }
    c64d* const wrow = a.work + g0;
    c64d* const image = data + (size_t)line * PL::LINE;
    c64d v[VMAX];
#pragma unroll
    for (int q = 0; q < VMAX; ++q) v[q] = c64d{0.0, 0.0};
    if (mode != ROW_FIRST) {
        // unconditional 16-byte buffer loads; a lane that must not take part (thread outside the layout, row past the edge, column tile the column
        // pass found empty and did not store) carries an offset beyond the descriptor's range and reads zeros
        const unsigned char* const nzf = a.nzflag ? a.nzflag + (size_t)s * a.nz_tiles : nullptr;
        if (nzf) {
            for (int i = tid; i < a.nz_tiles; i += THREADS) nzl[i] = nzf[i];
            __syncthreads();
        }
        const int tsh = 31 - __builtin_clz((unsigned)(a.nz_col_t > 0 ? a.nz_col_t : 1));
        const __amdgpu_buffer_rsrc_t wsrd = buf_srd(a.work + (size_t)s * per, (unsigned)(per * sizeof(c64d)));
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            const int i = tl + TPL_B * q;
            bool kept = live_b;
            if (nzf) kept = kept && nzl[i >> tsh] != 0;
            const p3d_u4 r = __builtin_amdgcn_raw_buffer_load_b128(wsrd, (int)(kept ? (unsigned)((rbase + i) * sizeof(c64d)) : BUF_OOB), 0, 0);
            v[q] = c64d{__hiloint2double((int)r.y, (int)r.x), __hiloint2double((int)r.w, (int)r.z)};
        }
        mix::line_fft<PL, INV, 1>(v, image, twl, tl);   // layout B -> layout A
    }
    // ---- re-insertion / first input, layout A ----
    double acc = 0.0;
    const double inv = 1.0 / N;
    const int i0 = tl < TPL_A ? tl : 0;
    const double* const mrow = a.mask ? a.mask + rbase : nullptr;
#pragma unroll
    for (int q = 0; q < PPT_A; ++q) {
        const int i = i0 + TPL_A * q;
        const c64d xo = load_x(a.x, a.dtype, g0 + i);
        const double m = mrow ? mrow[i] : 0.0;
        const double wgt = 1.0 - a.alpha * m;                       // POCS.py:616
        c64d xn;
        if (mode == ROW_FIRST) {
            xn = xo;
        } else {
            xn = (v[q] * inv) * wgt + xo * a.alpha;                  // POCS.py:617-619
            if (a.write_out && live_a) store_out(a.out, a.dtype, g0 + i, xn);
        }
        if (live_a) acc += hypot(xn.x, xn.y);
        if (a.adaptive) {   // POCS.py:574-575
            const c64d tmp = xo * a.alpha + xn * wgt;
            v[q] = tmp + (xo - xn * m) * (1.0 - a.alpha);
        } else {
            v[q] = xn;
        }
        if (!live_a) v[q] = c64d{0.0, 0.0};
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < (THREADS + 63) / 64; ++w) tot += red[w];
        a.partial[(size_t)s * gridDim.x + blockIdx.x] = tot;
    }
    if (mode == ROW_LAST) return;
    mix::line_fft<PL, FWD, 1>(v, image, twl, tl);   // layout A -> layout B
    if (live_b) {
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) wrow[tl + TPL_B * q] = v[q];
    }
}

// ---- the SHEARLET loop's passes (p3d_mix64.hpp) ----------------------------------------------------------------------------------------------------------
template <class PL>
__global__ __launch_bounds__(PL::COLT* PL::TMAX) void shear_col_kernel(const ShearCol64 a)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, T = PL::COLT;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c64d* data = reinterpret_cast<c64d*>(smem_raw);
    const c64d* const twl = a.tab;
    const int tid = threadIdx.x, c_lo = tid % T, tl = tid / T, u = blockIdx.y, tile = blockIdx.x;
    const int b = u / a.nsh, s = u - b * a.nsh;
    const int col = tile * T + c_lo;
    const bool valid = col < a.n2;
    if (a.done && a.done[b] != 0) return;
    c64d* const base = a.U + (size_t)u * N * a.n2 + (valid ? col : 0);
    const bool live_a = valid && tl < TPL_A, live_b = valid && tl < TPL_B;
    __shared__ unsigned char supl[N];   // per row of this shearlet's slice: 1 = the row exists (its spectrum does not vanish there)
    if (a.sup != nullptr) {
        const unsigned char* const sg = a.sup + (size_t)s * a.sup_groups;
        for (int r = tid; r < N; r += T * PL::TMAX) supl[r] = sg[r / a.sup_rows];
        __syncthreads();
    }
    c64d v[VMAX];
#pragma unroll
    for (int q = 0; q < VMAX; ++q) v[q] = c64d{0.0, 0.0};
    {
        const int r0 = tl < TPL_B ? tl : 0;
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            const int r = r0 + TPL_B * q;
            if (a.sup == nullptr || supl[r] != 0) v[q] = base[(size_t)r * a.n2];
        }
        if (!live_b) {
#pragma unroll
            for (int q = 0; q < PPT_B; ++q) v[q] = c64d{0.0, 0.0};
        }
    }
    mix::line_fft<PL, INV, T>(v, data + c_lo, twl, tl);   // layout B -> layout A
#pragma unroll
    for (int q = 0; q < PPT_A; ++q) {
        v[q] = v[q] * a.scale;
        if (a.real_only) v[q].y = 0.0;   // FFST hands back ST.real for real data
    }
    if (a.mode == 1) {
        if (live_a) {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) base[(size_t)(tl + TPL_A * q) * a.n2] = v[q];
        }
        return;
    }
    const c64d t = a.tau[((size_t)b * a.niter + a.iter) * a.nsh + s];
#pragma unroll
    for (int q = 0; q < PPT_A; ++q) v[q] = live_a ? shrink(v[q], t, a.op) : c64d{0.0, 0.0};
    mix::line_fft<PL, FWD, T>(v, data + c_lo, twl, tl);   // layout A -> layout B
    if (live_b) {
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            const int r = tl + TPL_B * q;
            if (a.sup == nullptr || supl[r] != 0) base[(size_t)r * a.n2] = v[q];   // (rows off the support: the gather pass multiplies them by zero -- it skips them)
        }
    }
}

// The same pass for REAL cubes on symmetric spectra (ShearCol64::pair): the coefficients c_s are real, so the slice U[b * nsh + s] -- the row-transformed
// spectrum, i.e. the column transform of c_s -- is Hermitian along its columns: U[N - k][x] = conj U[k][x].  Only rows 0 ... N / 2 exist (the row passes work on
// those alone), and two adjacent columns a, b go through ONE complex transform: Z[k] = Ua[k] + i Ub[k] (the rows beyond N / 2 from their mirror images) ->
// inverse transform -> Re = c_a, Im = c_b -> real threshold -> forward transform -> Ua'[k] = (Z'[k] + conj Z'[N - k]) / 2, Ub'[k] = (Z'[k] - conj Z'[N - k]) / 2i
// (the partner row comes through LDS).  A tile = COLT column pairs.  mode 1: c_a, c_b stored as samples (all N rows), for the statistics.
template <class PL>
__global__ __launch_bounds__(PL::COLT* PL::TMAX) void shear_col_pair_kernel(const ShearCol64 a)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, T = PL::COLT, H = N / 2;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c64d* data = reinterpret_cast<c64d*>(smem_raw);
    const c64d* const twl = a.tab;
    const int tid = threadIdx.x, c_lo = tid % T, tl = tid / T, u = blockIdx.y, tile = blockIdx.x;
    const int b = u / a.nsh, s = u - b * a.nsh;
    const int pcol = 2 * (tile * T + c_lo);   // columns pcol, pcol + 1
    const bool valid = pcol + 1 < a.n2;
    if (a.done && a.done[b] != 0) return;
    c64d* const base = a.U + (size_t)u * N * a.n2 + (valid ? pcol : 0);
    const bool live_a = valid && tl < TPL_A, live_b = valid && tl < TPL_B;
    __shared__ unsigned char supl[N];
    if (a.sup != nullptr) {
        const unsigned char* const sg = a.sup + (size_t)s * a.sup_groups;
        for (int r = tid; r < N; r += T * PL::TMAX) supl[r] = sg[r / a.sup_rows];
        __syncthreads();
    }
    c64d v[VMAX];
#pragma unroll
    for (int q = 0; q < VMAX; ++q) v[q] = c64d{0.0, 0.0};
    {
        const int r0 = tl < TPL_B ? tl : 0;
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            const int r = r0 + TPL_B * q, rr = r <= H ? r : N - r;
            if (a.sup == nullptr || supl[rr] != 0) {
                const c64d va = base[(size_t)rr * a.n2], vb = base[(size_t)rr * a.n2 + 1];
                // Z = Ua + i Ub; beyond N / 2: Ua = conj(mirror), Ub = conj(mirror)
                v[q] = r <= H ? c64d{va.x - vb.y, va.y + vb.x} : c64d{va.x + vb.y, vb.x - va.y};
            }
        }
        if (!live_b) {
#pragma unroll
            for (int q = 0; q < PPT_B; ++q) v[q] = c64d{0.0, 0.0};
        }
    }
    mix::line_fft<PL, INV, T>(v, data + c_lo, twl, tl);   // layout B -> layout A: Re = the samples of column pcol, Im = those of column pcol + 1
#pragma unroll
    for (int q = 0; q < PPT_A; ++q) v[q] = v[q] * a.scale;
    if (a.mode == 1) {
        if (live_a) {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) {
                c64d* const o = base + (size_t)(tl + TPL_A * q) * a.n2;
                o[0] = c64d{v[q].x, 0.0};
                o[1] = c64d{v[q].y, 0.0};
            }
        }
        return;
    }
    const c64d t = a.tau[((size_t)b * a.niter + a.iter) * a.nsh + s];
#pragma unroll
    for (int q = 0; q < PPT_A; ++q)
        v[q] = live_a ? c64d{shrink(c64d{v[q].x, 0.0}, t, a.op).x, shrink(c64d{v[q].y, 0.0}, t, a.op).x} : c64d{0.0, 0.0};
    mix::line_fft<PL, FWD, T>(v, data + c_lo, twl, tl);   // layout A -> layout B: Z'
    // Z'[N - k] of the rows this thread stores: once round through LDS in natural order (own addressing: [row][pair of the tile])
    __syncthreads();
    if (live_b) {
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) data[(size_t)(tl + TPL_B * q) * T + c_lo] = v[q];
    }
    __syncthreads();
    if (live_b) {
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            const int r = tl + TPL_B * q;
            if (r <= H && (a.sup == nullptr || supl[r] != 0)) {
                const c64d z = v[q], p = data[(size_t)(r == 0 ? 0 : N - r) * T + c_lo];
                base[(size_t)r * a.n2] = c64d{0.5 * (z.x + p.x), 0.5 * (z.y - p.y)};
                base[(size_t)r * a.n2 + 1] = c64d{0.5 * (z.y + p.y), 0.5 * (p.x - z.x)};
            }
        }
    }
}

template <class PL>
__global__ __launch_bounds__(PL::ROWLB* PL::TMAX) void spread_row_kernel(const SpreadRow64 a)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, TMAX = PL::TMAX, LB = PL::ROWLB;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c64d* data = reinterpret_cast<c64d*>(smem_raw);
    const c64d* const twl = a.tab;
    const int tid = threadIdx.x, line = tid / TMAX, tl = tid % TMAX, u = blockIdx.y, row = blockIdx.x * LB + line;
    const int b = u / a.nsh, s = u - b * a.nsh;
    if (a.done && a.done[b] != 0) return;
    if (a.sup != nullptr && a.sup[(size_t)s * a.sup_groups + blockIdx.x] == 0) return;   // the spectrum vanishes on this workgroup's rows: they are never read
    const bool valid = row < a.rows;
    const bool live_a = valid && tl < TPL_A, live_b = valid && tl < TPL_B;
    const size_t per = (size_t)a.n1 * N, rbase = (size_t)(valid ? row : 0) * N;
    const c64d* const frow = a.F + (size_t)b * per + rbase;
    const double* const prow = a.psi + (size_t)s * per + rbase;
    c64d* const image = data + (size_t)line * PL::LINE;
    c64d v[VMAX];
#pragma unroll
    for (int q = 0; q < VMAX; ++q) v[q] = c64d{0.0, 0.0};
    {
        const int i0 = tl < TPL_B ? tl : 0;
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) v[q] = frow[i0 + TPL_B * q] * prow[i0 + TPL_B * q];
        if (!live_b) {
#pragma unroll
            for (int q = 0; q < PPT_B; ++q) v[q] = c64d{0.0, 0.0};
        }
    }
    mix::line_fft<PL, INV, 1>(v, image, twl, tl);   // layout B -> layout A
    if (live_a) {
        c64d* const urow = a.U + (size_t)u * per + rbase;
#pragma unroll
        for (int q = 0; q < PPT_A; ++q) urow[tl + TPL_A * q] = v[q];
    }
}

template <class PL>
__global__ __launch_bounds__(PL::ROWLB* PL::TMAX) void gather_row_kernel(const GatherRow64 a)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, TMAX = PL::TMAX, LB = PL::ROWLB;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c64d* data = reinterpret_cast<c64d*>(smem_raw);
    const c64d* const twl = a.tab;
    const int tid = threadIdx.x, line = tid / TMAX, tl = tid % TMAX, b = blockIdx.y, row = blockIdx.x * LB + line;
    if (a.done && a.done[b] != 0) return;
    const bool valid = row < a.rows;
    const bool live_a = valid && tl < TPL_A, live_b = valid && tl < TPL_B;
    const size_t per = (size_t)a.n1 * N, rbase = (size_t)(valid ? row : 0) * N;
    c64d* const image = data + (size_t)line * PL::LINE;
    const int ia = tl < TPL_A ? tl : 0, ib = tl < TPL_B ? tl : 0;
    c64d acc[PPT_B];
#pragma unroll
    for (int q = 0; q < PPT_B; ++q) acc[q] = c64d{0.0, 0.0};
    for (int s = 0; s < a.nsh; ++s) {
        if (a.sup != nullptr && a.sup[(size_t)s * a.sup_groups + blockIdx.x] == 0) continue;   // (uniform over the workgroup)
        const c64d* const urow = a.U + ((size_t)b * a.nsh + s) * per + rbase;
        const double* const prow = a.psi + (size_t)s * per + rbase;
        c64d v[VMAX];
#pragma unroll
        for (int q = 0; q < VMAX; ++q) v[q] = c64d{0.0, 0.0};
#pragma unroll
        for (int q = 0; q < PPT_A; ++q) v[q] = urow[ia + TPL_A * q];
        if (!live_a) {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) v[q] = c64d{0.0, 0.0};
        }
        mix::line_fft<PL, FWD, 1>(v, image, twl, tl);   // layout A -> layout B
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            const double w = prow[ib + TPL_B * q];
            acc[q].x += v[q].x * w;
            acc[q].y += v[q].y * w;
        }
    }
    if (live_b) {
        c64d* const frow = a.F + (size_t)b * per + rbase;
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) frow[tl + TPL_B * q] = acc[q];
    }
}

constexpr size_t LDS_LIMIT = 160 * 1024;

template <class PL, class KERN>
hipError_t raise_lds(KERN kern, size_t lds, bool* done)
{
    if (*done) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) *done = true;
    return e;
}

template <class PL>
hipError_t launch_shear_col(const ShearCol64& a, hipStream_t st)
{
    constexpr size_t lds = sizeof(c64d) * ((size_t)PL::LINE * PL::COLT);
    static bool attr = false;
    const hipError_t e = raise_lds<PL>(shear_col_kernel<PL>, lds, &attr);
    if (e != hipSuccess) return e;
    if (a.pair) {
        if constexpr (PL::N % 2 == 0) {
            static bool attr2 = false;
            const hipError_t e2 = raise_lds<PL>(shear_col_pair_kernel<PL>, lds, &attr2);
            if (e2 != hipSuccess) return e2;
            shear_col_pair_kernel<PL><<<dim3((a.n2 / 2 + PL::COLT - 1) / PL::COLT, a.nslices), PL::COLT * PL::TMAX, lds, st>>>(a);
            return hipGetLastError();
        } else {
            return hipErrorNotSupported;
        }
    }
    shear_col_kernel<PL><<<dim3((a.n2 + PL::COLT - 1) / PL::COLT, a.nslices), PL::COLT * PL::TMAX, lds, st>>>(a);
    return hipGetLastError();
}

template <class PL>
hipError_t launch_spread_row(const SpreadRow64& a, hipStream_t st)
{
    constexpr size_t lds = sizeof(c64d) * ((size_t)PL::LINE * PL::ROWLB);
    static bool attr = false;
    const hipError_t e = raise_lds<PL>(spread_row_kernel<PL>, lds, &attr);
    if (e != hipSuccess) return e;
    spread_row_kernel<PL><<<dim3((a.rows + PL::ROWLB - 1) / PL::ROWLB, a.nb * a.nsh), PL::ROWLB * PL::TMAX, lds, st>>>(a);
    return hipGetLastError();
}

template <class PL>
hipError_t launch_gather_row(const GatherRow64& a, hipStream_t st)
{
    constexpr size_t lds = sizeof(c64d) * ((size_t)PL::LINE * PL::ROWLB);
    static bool attr = false;
    const hipError_t e = raise_lds<PL>(gather_row_kernel<PL>, lds, &attr);
    if (e != hipSuccess) return e;
    gather_row_kernel<PL><<<dim3((a.rows + PL::ROWLB - 1) / PL::ROWLB, a.nb), PL::ROWLB * PL::TMAX, lds, st>>>(a);
    return hipGetLastError();
}


template <class PL>
hipError_t launch_col(int mode, const ColArgs64& a, hipStream_t st)
{
    constexpr size_t lds = sizeof(c64d) * ((size_t)PL::LINE * PL::COLT);
    static_assert(lds + 2048 <= LDS_LIMIT, "column tile does not fit LDS");
    static_assert(PL::COLT * PL::TMAX <= 1024, "column tile needs more than 1024 threads");
    static bool attr = false;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(col_kernel<PL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    col_kernel<PL><<<dim3((a.n2 + PL::COLT - 1) / PL::COLT, a.nslices), PL::COLT * PL::TMAX, lds, st>>>(a, mode);
    return hipGetLastError();
}

template <class PL>
hipError_t launch_row(int mode, const RowArgs64& a, hipStream_t st)
{
    constexpr size_t lds = sizeof(c64d) * ((size_t)PL::LINE * PL::ROWLB);
    static_assert(lds + 2048 <= LDS_LIMIT, "row group does not fit LDS");
    static_assert(PL::ROWLB * PL::TMAX <= 1024, "row group needs more than 1024 threads");
    static bool attr = false;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(row_kernel<PL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    row_kernel<PL><<<dim3((a.n1 + PL::ROWLB - 1) / PL::ROWLB, a.nslices), PL::ROWLB * PL::TMAX, lds, st>>>(a, mode);
    return hipGetLastError();
}

#define P3D_MIX_PART P3D_MIX64_PART
#define P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3) mix::MixPlan<N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3>
#define X(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)                                                                             \
    {N, COLT, LB, P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)::TW_SLOTS,                                                  \
     &P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)::template build_tw<c64d>,                                               \
     &launch_col<P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)>, &launch_row<P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)>,                     \
     &launch_shear_col<P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)>, &launch_spread_row<P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)>,       \
     &launch_gather_row<P3D_PLAN64(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)>},
const Entry entries[] = {
#include "p3d_mix64_plans.inc"
    {0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
#undef X

}  // namespace

#define P3D_MIX64_CAT2(a, b) a##b
#define P3D_MIX64_CAT(a, b) P3D_MIX64_CAT2(a, b)
const Entry* P3D_MIX64_CAT(part_, P3D_MIX64_PART)() { return entries; }

#if P3D_MIX64_PART == 0
const Entry* part_1();
const Entry* part_2();
const Entry* part_3();
const Entry* part_4();
const Entry* part_5();
const Entry* part_6();
const Entry* part_7();

const Entry* find(int n)
{
    if (getenv("P3D_NO_MIX64")) return nullptr;   // experiment switch: the LDS-image passes of p3d_f64.hip for every length (read per call = per plan)
    static const std::map<int, const Entry*> table = [] {
        std::map<int, const Entry*> t;
        for (const Entry* (*part)() : {&part_0, &part_1, &part_2, &part_3, &part_4, &part_5, &part_6, &part_7})
            for (const Entry* e = part(); e->n != 0; ++e) t[e->n] = e;
        return t;
    }();
    const auto it = table.find(n);
    return it == table.end() ? nullptr : it->second;
}
#endif

}  // namespace mix64
}  // namespace p3d
