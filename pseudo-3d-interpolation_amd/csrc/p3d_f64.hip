// p3d_f64.hip -- the POCS loop in the REFERENCE's own precision: complex128 / float64 cubes, and complex64 / float32 cubes on request.
//
// The reference computes most of its configurations in double precision: under NumPy < 2 every path (np.fft returns complex128), under
// NumPy >= 2 everything but regular / hard on a complex64 cube -- the soft and garrote operators promote at the first threshold (tau is a
// complex128 scalar taken from an array: threshold_operator.py:37-39, 76-78), FPOCS through its float64 momentum scalar (POCS.py:566-571),
// APOCS and every alpha-weighted update through the float64 weight array 1 - alpha * mask (POCS.py:572-575, 616).  The float32 kernels of
// this library reproduce those runs to float32 rounding only, and the operators are discontinuous at |X| = Re tau (soft and garrote jump by
// |Im tau| there): 1e-4 instead of 1e-5 on ill-conditioned slices.  This file is the same loop in double precision.
// Two fused kernels per iteration like the float32 path (col64_kernel: forward transform, threshold, inverse transform of a tile of columns;
// row64_kernel: inverse transform, re-insertion, cost sum, forward transform of a tile of rows), on LDS-resident mixed-radix Stockham transforms
// with run-time factors -- a precision path, bound by its double-precision butterflies (rates in DESIGN.md section 5).  Any line length up to 5120;
// lines whose tile does not fit LDS (beyond 5088 points) take the first cut of this file, six plain passes over the cube per iteration built
// from the pieces of the any-length pipeline (p3d_generic.hip), which P3D_F64_UNFUSED=1 selects for every shape.
//
// Entry points (include/p3d.h): p3d_plan64_create / _destroy, p3d_pocs64_stats, p3d_pocs64_run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "p3d.h"
#include "p3d_generic.hpp"
#include "p3d_internal.hpp"
#include "p3d_mix64.hpp"

namespace {

using p3d::GenPlan;

struct __attribute__((aligned(16))) c64 {
    double x, y;
};
__device__ __forceinline__ c64 operator+(c64 a, c64 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c64 operator-(c64 a, c64 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c64 operator*(c64 a, c64 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c64 operator*(c64 a, double s) { return {a.x * s, a.y * s}; }

constexpr int F64_MAX_N = 5120;

int f64fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    p3d::set_last_error(buf);
    return code;
}
#define F_TRY(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return f64fail(P3D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <int R>
__device__ inline void butterfly64(const c64* A, c64* B, const c64* tw, int j, int m, int jm, int j0, int ns, int tstep, int rstep, int dir)
{
    c64 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        c64 w = tw[t * jm * tstep];
        if (dir > 0) w.y = -w.y;
        v[t] = A[j + t * m] * w;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        c64 acc = v[0];
#pragma unroll
        for (int t = 1; t < R; ++t) {
            c64 w = tw[((t * k) % R) * rstep];
            if (dir > 0) w.y = -w.y;
            acc = acc + v[t] * w;
        }
        B[j0 + k * ns] = acc;
    }
}

// one workgroup per line, the line in LDS (ping-pong); addressing as gen_line_fft (p3d_generic.hip)
__global__ void line_fft64(const c64* in, c64* out, const c64* tw, GenPlan pl, int dir, double scale, int es, int lpo, size_t outer, size_t inner,
                           const int* done, int lines_per_slice)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = pl.n;
    c64* A = reinterpret_cast<c64*>(smem_raw);
    c64* B = A + n;
    const size_t l = blockIdx.x;
    if (done && done[l / lines_per_slice] != 0) return;
    const size_t base = (l / lpo) * outer + (l % lpo) * inner;
    for (int i = threadIdx.x; i < n; i += blockDim.x) A[i] = in[base + (size_t)i * es];
    __syncthreads();
    int ns = 1;
    for (int p = 0; p < pl.nf; ++p) {
        const int R = pl.f[p];
        const int m = n / R, tstep = n / (ns * R), rstep = n / R;
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            const int jm = j % ns;
            const int j0 = (j / ns) * ns * R + jm;
            switch (R <= 8 ? R : 0) {
                case 2: butterfly64<2>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                case 3: butterfly64<3>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                case 4: butterfly64<4>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                case 5: butterfly64<5>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                case 7: butterfly64<7>(A, B, tw, j, m, jm, j0, ns, tstep, rstep, dir); break;
                default:   // a large prime factor: direct butterfly
                    for (int k = 0; k < R; ++k) {
                        c64 acc{0.0, 0.0};
                        for (int t = 0; t < R; ++t) {
                            const long idx = ((long)t * jm * tstep + (long)((long)t * k % R) * rstep) % n;
                            c64 w = tw[idx];
                            if (dir > 0) w.y = -w.y;
                            acc = acc + A[j + t * m] * w;
                        }
                        B[j0 + k * ns] = acc;
                    }
            }
        }
        __syncthreads();
        c64* t = A; A = B; B = t;
        ns *= R;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[base + (size_t)i * es] = A[i] * scale;
}

// threshold_operator.py:9-112 on one coefficient, NumPy's semantics for a complex tau (lexicographic comparisons)
__device__ __forceinline__ c64 shrink64(c64 X, c64 tau, int op)
{
    const double m = hypot(X.x, X.y);   // np.absolute
    if (op == 0) {                      // hard: where(|X| < tau, 0, X)
        const bool below = m < tau.x || (m == tau.x && 0.0 < tau.y);
        return below ? c64{0.0, 0.0} : X;
    }
    if (m == 0.0) return c64{0.0, 0.0};   // 1 - tau / 0 = -inf: clipped to 0
    double gr, gi;
    if (op == 1) {          // soft: X * clip(1 - tau / |X|, 0)
        gr = 1.0 - tau.x / m;
        gi = -tau.y / m;
    } else {                // garrote: X * clip(1 - tau^2 / |X|^2, 0)
        const double m2 = m * m;
        gr = 1.0 - (tau.x * tau.x - tau.y * tau.y) / m2;
        gi = -(2.0 * tau.x * tau.y) / m2;
    }
    const bool keep = (gr > 0.0) || (gr == 0.0 && gi >= 0.0);   // lexicographic max(g, 0)
    return keep ? X * c64{gr, gi} : c64{0.0, 0.0};
}

__global__ void shrink64_kernel(c64* w, const c64* tau, int niter, int iter, int op, size_t per_slice, const int* done)
{
    const int s = blockIdx.y;
    if (done && done[s] != 0) return;
    const c64 t = tau[(size_t)s * niter + iter];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x)
        w[(size_t)s * per_slice + i] = shrink64(w[(size_t)s * per_slice + i], t, op);
}

__device__ inline double block_sum64(double v, double* sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    return sh[0];
}

// per (slice, block) partial sums of |x| (fixed order: the host adds the blocks of a slice in order -- reproducible costs)
// mode 0: first input (w = x or its APOCS mix); mode 1: re-insertion (POCS.py:616-619), optional store of the iterate, APOCS mix
// dtype: P3D_C128 complex128 / P3D_F64 float64 / P3D_C64 complex64 / P3D_F32 float32 (of x and out; the arithmetic is double either way)
__global__ void update64_kernel(c64* w, const void* x, int dtype, const double* mask, void* out, double* partial, int mode, int adaptive, int write_out,
                                double alpha, size_t per_slice, const int* done, int zero_fill)
{
    __shared__ double sh[256];
    const int s = blockIdx.y;
    const int dn = done ? done[s] : 0;
    auto put = [&](size_t g, c64 v) {
        if (dtype == P3D_C128) reinterpret_cast<c64*>(out)[g] = v;
        else if (dtype == P3D_F64) reinterpret_cast<double*>(out)[g] = v.x;
        else if (dtype == P3D_C64) reinterpret_cast<float2*>(out)[g] = float2{(float)v.x, (float)v.y};
        else reinterpret_cast<float*>(out)[g] = (float)v.x;
    };
    if (zero_fill && dn < 0) {   // an empty slice is handed back untouched (zeros), POCS.py:515-521
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x) put((size_t)s * per_slice + i, c64{0.0, 0.0});
    }
    if (dn != 0) {
        if (threadIdx.x == 0) partial[(size_t)s * gridDim.x + blockIdx.x] = 0.0;
        return;
    }
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = (size_t)s * per_slice + i;
        c64 xo;
        if (dtype == P3D_C128) xo = reinterpret_cast<const c64*>(x)[g];
        else if (dtype == P3D_F64) xo = c64{reinterpret_cast<const double*>(x)[g], 0.0};
        else if (dtype == P3D_C64) { const float2 t = reinterpret_cast<const float2*>(x)[g]; xo = c64{(double)t.x, (double)t.y}; }
        else xo = c64{(double)reinterpret_cast<const float*>(x)[g], 0.0};
        const double m = mask ? mask[i] : 0.0;
        const double wgt = 1.0 - alpha * m;
        c64 xn;
        if (mode == 0) {
            xn = xo;
        } else {
            xn = w[g] * wgt + xo * alpha;
            if (write_out) put(g, xn);
        }
        acc += hypot(xn.x, xn.y);
        if (adaptive) {   // POCS.py:574-575
            const c64 tmp = xo * alpha + xn * wgt;
            w[g] = tmp + (xo - xn * m) * (1.0 - alpha);
        } else {
            w[g] = xn;
        }
    }
    const double tot = block_sum64(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)s * gridDim.x + blockIdx.x] = tot;
}

// per slice: sums[s] = the blocks' partial sums in a FIXED order (reproducible costs): one wavefront per slice, lane l adds blocks l, l + 64, ...
// in order, the 64 lane sums are combined by a fixed tree
__global__ void fold64_kernel(const double* partial, double* sums, int nslices, int blocks)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    double t = 0.0;
    for (int b = lane; b < blocks; b += 64) t += partial[(size_t)s * blocks + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if (lane == 0) sums[s] = t;
}

__global__ void conv64_kernel(const double* sums, int* done, int nslices, int iter, double eps)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices || done[s] != 0) return;
    const double cur = sums[(size_t)(iter + 1) * nslices + s], prev = sums[(size_t)iter * nslices + s];
    const double d = cur - prev;
    if (iter > 2 && (d * d) / (cur * cur) < eps) done[s] = iter + 1;   // POCS.py:622, 631
}

// per block: lexicographic max, max |X|, min |X|, sum |X|^2 -> partial[(s * blocks + b) * 8 ..]
__global__ void stats64_kernel(const c64* w, double* partial, size_t per_slice)
{
    __shared__ double sh[256 * 5];
    const int s = blockIdx.y;
    double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY, sq = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (size_t)gridDim.x * blockDim.x) {
        const c64 v = w[(size_t)s * per_slice + i];
        const double a = hypot(v.x, v.y);
        if (v.x > lr || (v.x == lr && v.y > li)) { lr = v.x; li = v.y; }
        mx = fmax(mx, a);
        mn = fmin(mn, a);
        sq += v.x * v.x + v.y * v.y;
    }
    double* me = sh + threadIdx.x * 5;
    me[0] = lr; me[1] = li; me[2] = mx; me[3] = mn; me[4] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < (int)blockDim.x; ++t) {
            const double* o = sh + t * 5;
            if (o[0] > lr || (o[0] == lr && o[1] > li)) { lr = o[0]; li = o[1]; }
            mx = fmax(mx, o[2]);
            mn = fmin(mn, o[3]);
            sq += o[4];
        }
        double* p = partial + ((size_t)s * gridDim.x + blockIdx.x) * 8;
        p[0] = lr; p[1] = li; p[2] = mx; p[3] = mn; p[4] = sq;
    }
}


// ==================================================================================================================================
// Fused passes (round 4): the loop as TWO kernels per iteration, like the float32 path -- a column pass (forward transform, threshold,
// inverse transform of a tile of adjacent columns) and a row pass (inverse transform, re-insertion, cost sum, forward transform of a
// few rows) -- instead of six passes over the cube with one line per workgroup (whose column transforms read 16 bytes per 16-KiB
// stride).  Lines live in LDS (two buffers, Stockham passes with in-register butterflies of 2, 3, 4, 5, 7, 8 and 9 points -- other
// prime factors as direct sums spread over the threads), the twiddle table exp(-2 pi i k / n) beside them when it fits.  The unfused
// kernels above remain the path of lines too long for that (two buffers + table > 160 KiB) and of P3D_F64_UNFUSED=1.
// ==================================================================================================================================
constexpr int F64_MAX_PASSES = 16;
struct Fft64 {
    int n, nf;
    int f[F64_MAX_PASSES];    // radices; 2, 3, 4, 5, 7, 8, 9: in-register butterflies, anything else: direct sums
    int ns[F64_MAX_PASSES];   // product of the earlier radices = distance of a butterfly's outputs
    unsigned mg[F64_MAX_PASSES];   // j / ns = umulhi(j, mg) for j < 2^16 (ns > 1)
};

Fft64 make_fft64(int n)
{
    Fft64 p{};
    p.n = n;
    int r = n, k = 0;
    auto push = [&](int R) { if (k < F64_MAX_PASSES) p.f[k] = R; ++k; };
    // odd radices first: the scattered writes of an early pass (small output stride) then have an odd stride
    while (r % 9 == 0) { push(9); r /= 9; }
    for (int q : {3, 5, 7}) while (r % q == 0) { push(q); r /= q; }
    for (int q = 11; (long)q * q <= r; q += 2) while (r % q == 0) { push(q); r /= q; }
    { int odd = r; while (odd % 2 == 0) odd /= 2; if (odd > 1) { push(odd); r /= odd; } }
    while (r % 8 == 0) { push(8); r /= 8; }
    while (r % 4 == 0) { push(4); r /= 4; }
    while (r % 2 == 0) { push(2); r /= 2; }
    p.nf = k <= F64_MAX_PASSES ? k : -1;
    int ns = 1;
    for (int i = 0; i < k && i < F64_MAX_PASSES; ++i) {
        p.ns[i] = ns;
        p.mg[i] = ns > 1 ? (unsigned)((0x100000000ull + (unsigned)ns - 1) / (unsigned)ns) : 0u;   // ceil(2^32 / ns): exact for j ns < 2^32
        ns *= p.f[i];
    }
    return p;
}

template <int R> struct Roots64;
template <> struct Roots64<3> {
    static constexpr double c[3] = {1.0, -0.5, -0.5};
    static constexpr double s[3] = {0.0, 0.86602540378443864676, -0.86602540378443864676};
};
template <> struct Roots64<5> {
    static constexpr double c[5] = {1.0, 0.3090169943749474241, -0.8090169943749474241, -0.8090169943749474241, 0.3090169943749474241};
    static constexpr double s[5] = {0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917, -0.95105651629515357212};
};
template <> struct Roots64<7> {
    static constexpr double c[7] = {1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624, -0.90096886790241912624, -0.22252093395631440429, 0.62348980185873353053};
    static constexpr double s[7] = {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048, -0.43388373911755812048, -0.97492791218182360702, -0.78183148246802980871};
};
template <> struct Roots64<8> {
    static constexpr double c[8] = {1.0, 0.7071067811865475244, 0.0, -0.7071067811865475244, -1.0, -0.7071067811865475244, 0.0, 0.7071067811865475244};
    static constexpr double s[8] = {0.0, 0.7071067811865475244, 1.0, 0.7071067811865475244, 0.0, -0.7071067811865475244, -1.0, -0.7071067811865475244};
};
template <> struct Roots64<9> {
    static constexpr double c[9] = {1.0, 0.7660444431189780352, 0.17364817766693034885, -0.5, -0.93969262078590838405, -0.93969262078590838405, -0.5, 0.17364817766693034885, 0.7660444431189780352};
    static constexpr double s[9] = {0.0, 0.64278760968653932632, 0.98480775301220805937, 0.86602540378443864676, 0.34202014332566873304, -0.34202014332566873304, -0.86602540378443864676, -0.98480775301220805937, -0.64278760968653932632};
};

// entry k of the twiddle table; `half` != 0: the table holds k < half = n / 2 only, exp(-2 pi i (k + n/2) / n) = -exp(-2 pi i k / n)
// (half the LDS: what lets two workgroups with a table each share a CU at 1024 points)
__device__ __forceinline__ c64 tw_at(const c64* tw, int k, int half)
{
    if (half == 0) return tw[k];
    const bool hi = k >= half;
    const c64 w = tw[hi ? k - half : k];
    return hi ? c64{-w.x, -w.y} : w;
}
// a * w (forward) or a * conj(w) (inverse): the table holds exp(-2 pi i k / n)
template <int DIR> __device__ __forceinline__ c64 twmul(c64 a, c64 w) { return DIR > 0 ? c64{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y} : a * w; }
// a +- i b
__device__ __forceinline__ c64 add_i(c64 a, c64 b) { return {a.x - b.y, a.y + b.x}; }
__device__ __forceinline__ c64 sub_i(c64 a, c64 b) { return {a.x + b.y, a.y - b.x}; }

// X[k] = sum_t x[t] W^(t k), W = exp(DIR 2 pi i / R), natural order in and out
template <int R, int DIR>
__device__ __forceinline__ void dft_small64(c64 (&x)[R])
{
    if constexpr (R == 2) {
        const c64 a = x[0] + x[1], b = x[0] - x[1];
        x[0] = a; x[1] = b;
    } else if constexpr (R == 4) {
        const c64 a = x[0] + x[2], b = x[0] - x[2], s = x[1] + x[3], t = x[1] - x[3];
        x[0] = a + s; x[2] = a - s;
        x[1] = DIR > 0 ? add_i(b, t) : sub_i(b, t);
        x[3] = DIR > 0 ? sub_i(b, t) : add_i(b, t);
    } else {
        // odd prime: X[k], X[R-k] = A_k +- i B_k, A_k = x0 + sum_q cos(2 pi q k / R) (x[q] + x[R-q]), B_k = DIR sum_q sin(2 pi q k / R) (x[q] - x[R-q])
        constexpr int H = (R - 1) / 2;
        c64 sp[H], dm[H];
        c64 x0 = x[0];
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            sp[q - 1] = x[q] + x[R - q];
            dm[q - 1] = x[q] - x[R - q];
            x0 = x0 + sp[q - 1];
        }
        const c64 xin = x[0];
        x[0] = x0;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            c64 A = xin, B{0.0, 0.0};
#pragma unroll
            for (int q = 1; q <= H; ++q) {
                const double cq = Roots64<R>::c[(q * k) % R], sq = DIR > 0 ? Roots64<R>::s[(q * k) % R] : -Roots64<R>::s[(q * k) % R];
                A = A + sp[q - 1] * cq;
                B = B + dm[q - 1] * sq;
            }
            x[k] = add_i(A, B);
            x[R - k] = sub_i(A, B);
        }
    }
}

// R = R1 R2 in registers (t = R2 t1 + t2, k = k1 + R1 k2)
template <int R1, int R2, int DIR>
__device__ __forceinline__ void radix64(c64 (&v)[R1 * R2])
{
    constexpr int R = R1 * R2;
    if constexpr (R2 == 1) {
        dft_small64<R, DIR>(v);
    } else {
        c64 u[R];
#pragma unroll
        for (int t2 = 0; t2 < R2; ++t2) {
            c64 a[R1];
#pragma unroll
            for (int t1 = 0; t1 < R1; ++t1) a[t1] = v[R2 * t1 + t2];
            dft_small64<R1, DIR>(a);
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) {
                constexpr int dummy = 0; (void)dummy;
                const int q = (k1 * t2) % R;
                u[k1 * R2 + t2] = q == 0 ? a[k1] : a[k1] * c64{Roots64<R>::c[q], DIR > 0 ? Roots64<R>::s[q] : -Roots64<R>::s[q]};
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            c64 b[R2];
#pragma unroll
            for (int t2 = 0; t2 < R2; ++t2) b[t2] = u[k1 * R2 + t2];
            dft_small64<R2, DIR>(b);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = b[k2];
        }
    }
}

// Addressing of a tile of LN lines in LDS.  COLS: the lines interleaved, element i of line l at X[i * LN + l] (a tile of adjacent
// columns: a row of the tile is contiguous, as in memory); else line after line, element i of line l at X[l * n + i] (rows).
// (one padding slot per 16 elements of a line, the recipe of the float32 engine against the strided Stockham writes, made these passes 30 %
// SLOWER -- 16-byte elements; profiles/r04_f64_fused.txt -- : the lines are not padded)
template <bool COLS> __device__ __forceinline__ int at64(int i, int l, int n, int LN) { return COLS ? i * LN + l : l * n + i; }

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }   // a wave-uniform value: into a scalar register

// one Stockham pass of radix R = R1 R2 over the tile: butterfly j of line l reads in[j + t nb], multiplies by w^(t jm ts), writes
// out[j0 + k ns] (jq = j / ns, jm = j mod ns, j0 = jq ns R + jm).  A, B: the two buffers of the tile.
// (LN and the thread count are powers of two: a thread's line and its first butterfly come from shifts -- lsh = log2 LN --, the
// quotient by ns from a multiplication; what a pass derives from its run-time constants -- the strides of a butterfly's inputs and
// outputs -- is pinned to scalar registers: as vector values they cost a quarter-rate 32-bit multiplication per LDS access)
template <int R1, int R2, int DIR, bool COLS>
__device__ __forceinline__ void pass64(const c64* A, c64* B, const c64* tw, int half, int n, int ns, unsigned mg, int LN, int lsh, int tid, int nthr)
{
    constexpr int R = R1 * R2;
    n = uni(n); ns = uni(ns); LN = uni(LN); lsh = uni(lsh); half = uni(half);
    const int nb = uni(n / R), ts = uni(n / (ns * R));
    const int tpl = uni(nthr >> lsh);                              // threads per line
    const int sa = COLS ? nb << lsh : nb, sb = COLS ? ns << lsh : ns;   // element distance of a butterfly's inputs / outputs
    const int l = COLS ? tid & (LN - 1) : tid >> (31 - __clz(tpl));
    const int lb = COLS ? l : l * n;                               // the thread's line
    for (int j = COLS ? tid >> lsh : tid & (tpl - 1); j < nb; j += tpl) {
        const int jq = ns > 1 ? (int)__umulhi((unsigned)j, mg) : j, jm = j - jq * ns, j0 = jq * ns * R + jm;
        const c64* in = A + (lb + (COLS ? j << lsh : j));
        c64* out = B + (lb + (COLS ? j0 << lsh : j0));
        c64 v[R];
        v[0] = in[0];
        if (ns == 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = in[t * sa];
        } else {
            const int twi = jm * ts;
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = twmul<DIR>(in[t * sa], tw_at(tw, t * twi, half));
        }
        radix64<R1, R2, DIR>(v);
#pragma unroll
        for (int k = 0; k < R; ++k) out[k * sb] = v[k];
    }
}

// any other (prime) radix: every OUTPUT of every butterfly is a direct sum, spread over the threads
template <int DIR, bool COLS>
__device__ __forceinline__ void pass64_direct(const c64* A, c64* B, const c64* tw, int half, int n, int R, int ns, int LN, int tid, int nthr)
{
    const int nb = n / R, ts = n / (ns * R), total = n * LN;
    for (int o = tid; o < total; o += nthr) {
        int rest = o;
        const int l = COLS ? rest % LN : rest / n;
        rest = COLS ? rest / LN : rest % n;          // (k, j): output k of butterfly j
        const int k = rest / nb, j = rest - k * nb;
        const int jq = j / ns, jm = j - jq * ns, j0 = jq * ns * R + jm;
        c64 acc{0.0, 0.0};
        for (int t = 0; t < R; ++t) {
            const long idx = ((long)t * jm * ts + (long)((long)t * k % R) * nb) % n;
            acc = acc + twmul<DIR>(A[at64<COLS>(j + t * nb, l, n, LN)], tw_at(tw, (int)idx, half));
        }
        B[at64<COLS>(j0 + k * ns, l, n, LN)] = acc;
    }
}

// all passes of the tile; returns the buffer that holds the result (every thread of the workgroup calls this)
template <int DIR, bool COLS>
__device__ __forceinline__ c64* tile_fft64(c64* A, c64* B, const c64* tw, int half, const Fft64& pl, int LN, int tid, int nthr)
{
    const int lsh = 31 - __clz(LN);
    for (int p = 0; p < pl.nf; ++p) {
        const int ns = pl.ns[p];
        switch (pl.f[p]) {
            case 2: pass64<2, 1, DIR, COLS>(A, B, tw, half, pl.n, ns, pl.mg[p], LN, lsh, tid, nthr); break;
            case 3: pass64<3, 1, DIR, COLS>(A, B, tw, half, pl.n, ns, pl.mg[p], LN, lsh, tid, nthr); break;
            case 4: pass64<4, 1, DIR, COLS>(A, B, tw, half, pl.n, ns, pl.mg[p], LN, lsh, tid, nthr); break;
            case 5: pass64<5, 1, DIR, COLS>(A, B, tw, half, pl.n, ns, pl.mg[p], LN, lsh, tid, nthr); break;
            case 7: pass64<7, 1, DIR, COLS>(A, B, tw, half, pl.n, ns, pl.mg[p], LN, lsh, tid, nthr); break;
            case 8: pass64<2, 4, DIR, COLS>(A, B, tw, half, pl.n, ns, pl.mg[p], LN, lsh, tid, nthr); break;
            case 9: pass64<3, 3, DIR, COLS>(A, B, tw, half, pl.n, ns, pl.mg[p], LN, lsh, tid, nthr); break;
            default: pass64_direct<DIR, COLS>(A, B, tw, half, pl.n, pl.f[p], ns, LN, tid, nthr);
        }
        __syncthreads();
        c64* t = A; A = B; B = t;
    }
    return A;
}

constexpr int F64_THREADS = 512;
enum { C64_ITER = 0, C64_STATS = 1, C64_FWD = 2 };

// Column pass of one tile of LN adjacent columns of a slice.  C64_ITER: forward transform, threshold (tau of the slice and iteration),
// inverse transform, 1 / n1, in place on the work buffer.  C64_STATS: forward transform, then the tile's statistics (lexicographic
// maximum, max |X|, min |X|, sum |X|^2) -> partial[(slice, tile)][8].  C64_FWD: forward transform only.
// grid (tiles, slices); dynamic LDS: [table n1 if tw_lds][2 x LN n1]
// TWL: the twiddle table (or its first half) sits in LDS -- a template parameter, so that its loads are LDS instructions and not flat ones
template <int MODE, bool TWL>
__global__ __launch_bounds__(F64_THREADS) void col64_kernel(c64* work, const c64* tw_g, Fft64 pl, int n2, int LN, int tw_lds, const c64* tau, int niter, int iter, int op,
                                                            double* partial, const int* done)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[(F64_THREADS / 64) * 5];
    const int tid = threadIdx.x, nthr = blockDim.x, n = pl.n, s = blockIdx.y, c0 = blockIdx.x * LN;
    if (done && done[s] != 0) return;
    c64* twl = reinterpret_cast<c64*>(smem_raw);
    const int half = TWL && tw_lds == 2 ? n / 2 : 0, tw_n = !TWL ? 0 : (tw_lds == 2 ? n / 2 : n);   // tw_lds: 0 the table stays in memory, 1 in LDS, 2 its first half in LDS
    c64* A = twl + tw_n;
    c64* B = A + (size_t)LN * n;
    c64* const base = work + (size_t)s * n * n2;
    const int lsh = 31 - __clz(LN);
    const int lv = min(LN, n2 - c0);   // valid columns of the tile
    for (int e = tid; e < n * LN; e += nthr) {
        const int i = e >> lsh, l = e & (LN - 1);
        A[at64<true>(i, l, n, LN)] = l < lv ? base[(size_t)i * n2 + c0 + l] : c64{0.0, 0.0};
    }
    for (int k = tid; k < tw_n; k += nthr) twl[k] = tw_g[k];
    const c64* tw = TWL ? twl : tw_g;
    __syncthreads();
    c64* X = tile_fft64<-1, true>(A, B, tw, half, pl, LN, tid, nthr);
    c64* Y = X == A ? B : A;
    if (MODE == C64_STATS) {
        double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY, sq = 0.0;
        for (int e = tid; e < n * LN; e += nthr) {
            if ((e & (LN - 1)) >= lv) continue;
            const c64 v = X[at64<true>(e >> lsh, e & (LN - 1), n, LN)];
            const double a = hypot(v.x, v.y);
            if (v.x > lr || (v.x == lr && v.y > li)) { lr = v.x; li = v.y; }
            mx = fmax(mx, a);
            mn = fmin(mn, a);
            sq += v.x * v.x + v.y * v.y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
            if (orr > lr || (orr == lr && oi > li)) { lr = orr; li = oi; }
            mx = fmax(mx, __shfl_down(mx, o, 64));
            mn = fmin(mn, __shfl_down(mn, o, 64));
            sq += __shfl_down(sq, o, 64);
        }
        if ((tid & 63) == 0) { double* r = red + (tid >> 6) * 5; r[0] = lr; r[1] = li; r[2] = mx; r[3] = mn; r[4] = sq; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (nthr + 63) / 64; ++w) {
                const double* r = red + w * 5;
                if (r[0] > lr || (r[0] == lr && r[1] > li)) { lr = r[0]; li = r[1]; }
                mx = fmax(mx, r[2]);
                mn = fmin(mn, r[3]);
                sq += r[4];
            }
            double* q = partial + ((size_t)s * gridDim.x + blockIdx.x) * 8;
            q[0] = lr; q[1] = li; q[2] = mx; q[3] = mn; q[4] = sq;
        }
        return;
    }
    double scale = 1.0;
    if (MODE == C64_ITER) {
        const c64 t = tau[(size_t)s * niter + iter];
        for (int e = tid; e < n * LN; e += nthr) {
            c64& v = X[at64<true>(e >> lsh, e & (LN - 1), n, LN)];
            v = shrink64(v, t, op);
        }
        __syncthreads();
        X = tile_fft64<+1, true>(X, Y, tw, half, pl, LN, tid, nthr);
        scale = 1.0 / n;
    }
    for (int e = tid; e < n * LN; e += nthr) {
        const int i = e >> lsh, l = e & (LN - 1);
        if (l < lv) base[(size_t)i * n2 + c0 + l] = X[at64<true>(i, l, n, LN)] * scale;
    }
}

enum { R64_FIRST = 0, R64_MID = 1, R64_LAST = 2 };

// Row pass of LN consecutive rows of a slice (n = n2 samples each; n1 rows per slice).
//   R64_FIRST: w = x (adaptive: the APOCS mix of the first input, POCS.py:574-575 with x_old = x), sum |x| -> partial, forward transform.
//   R64_MID:   inverse transform, 1 / n2, re-insertion (POCS.py:616-619), sum |x_new| -> partial, the iterate to `out` if asked, APOCS mix,
//              forward transform.     R64_LAST: the same without the forward transform (nothing feeds on the iterate any more).
// partial[(slice, workgroup)]: the host-side fold adds a slice's workgroups in order (reproducible costs).
// dynamic LDS as in col64_kernel
template <int MODE, bool TWL>
__global__ __launch_bounds__(F64_THREADS) void row64_kernel(c64* work, const c64* tw_g, Fft64 pl, int n1, int LN, int tw_lds, const void* x, int dtype, const double* mask,
                                                            void* out, double* partial, int adaptive, int write_out, double alpha, const int* done, int zero_fill)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[F64_THREADS / 64];
    const int tid = threadIdx.x, nthr = blockDim.x, n = pl.n, s = blockIdx.y, r0 = blockIdx.x * LN;
    const int dn = done ? done[s] : 0;
    const size_t per = (size_t)n1 * n;
    const int lv = min(LN, n1 - r0);   // valid rows of the tile
    auto put = [&](size_t g, c64 v) {
        if (dtype == P3D_C128) reinterpret_cast<c64*>(out)[g] = v;
        else if (dtype == P3D_F64) reinterpret_cast<double*>(out)[g] = v.x;
        else if (dtype == P3D_C64) reinterpret_cast<float2*>(out)[g] = float2{(float)v.x, (float)v.y};
        else reinterpret_cast<float*>(out)[g] = (float)v.x;
    };
    if (MODE != R64_FIRST && zero_fill && dn < 0) {   // an empty slice is handed back untouched (zeros), POCS.py:515-521
        for (int e = tid; e < n * lv; e += nthr) put((size_t)s * per + (size_t)r0 * n + e, c64{0.0, 0.0});
    }
    if (dn != 0) {
        if (tid == 0) partial[(size_t)s * gridDim.x + blockIdx.x] = 0.0;
        return;
    }
    c64* twl = reinterpret_cast<c64*>(smem_raw);
    const int half = TWL && tw_lds == 2 ? n / 2 : 0, tw_n = !TWL ? 0 : (tw_lds == 2 ? n / 2 : n);
    c64* A = twl + tw_n;
    c64* B = A + (size_t)LN * n;
    c64* const base = work + (size_t)s * per + (size_t)r0 * n;   // the tile's rows are one contiguous range
    for (int k = tid; k < tw_n; k += nthr) twl[k] = tw_g[k];
    const c64* tw = TWL ? twl : tw_g;
    c64* X = A;
    if (MODE != R64_FIRST) {
        for (int l = 0; l < LN; ++l)
            for (int i = tid; i < n; i += nthr) A[at64<false>(i, l, n, LN)] = l < lv ? base[(size_t)l * n + i] : c64{0.0, 0.0};
        __syncthreads();
        X = tile_fft64<+1, false>(A, B, tw, half, pl, LN, tid, nthr);
    } else {
        __syncthreads();   // (the table)
    }
    c64* Y = X == A ? B : A;
    // ---- re-insertion / first input, element by element; the tile's samples are one contiguous range of the slice ----
    double acc = 0.0;
    const double inv = 1.0 / n;
    for (int l = 0; l < LN; ++l) {
        for (int i = tid; i < n; i += nthr) {
            c64& slot = X[at64<false>(i, l, n, LN)];
            if (l >= lv) { slot = c64{0.0, 0.0}; continue; }
            const size_t li = (size_t)(r0 + l) * n + i, g = (size_t)s * per + li;
            c64 xo;
            if (dtype == P3D_C128) xo = reinterpret_cast<const c64*>(x)[g];
            else if (dtype == P3D_F64) xo = c64{reinterpret_cast<const double*>(x)[g], 0.0};
            else if (dtype == P3D_C64) { const float2 t = reinterpret_cast<const float2*>(x)[g]; xo = c64{(double)t.x, (double)t.y}; }
            else xo = c64{(double)reinterpret_cast<const float*>(x)[g], 0.0};
            const double m = mask ? mask[li] : 0.0;
            const double wgt = 1.0 - alpha * m;
            c64 xn;
            if (MODE == R64_FIRST) {
                xn = xo;
            } else {
                xn = (slot * inv) * wgt + xo * alpha;
                if (write_out) put(g, xn);
            }
            acc += hypot(xn.x, xn.y);
            if (adaptive) {   // POCS.py:574-575
                const c64 tmp = xo * alpha + xn * wgt;
                slot = tmp + (xo - xn * m) * (1.0 - alpha);
            } else {
                slot = xn;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();   // (also: the tile is complete for the forward transform)
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < (nthr + 63) / 64; ++w) tot += red[w];
        partial[(size_t)s * gridDim.x + blockIdx.x] = tot;
    }
    if (MODE == R64_LAST) return;
    X = tile_fft64<-1, false>(X, Y, tw, half, pl, LN, tid, nthr);
    for (int l = 0; l < lv; ++l)
        for (int i = tid; i < n; i += nthr) base[(size_t)l * n + i] = X[at64<false>(i, l, n, LN)];
}

}  // namespace

struct p3d_plan64 {
    int device = 0, nil = 0, nxl = 0, max_slices = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    GenPlan gcol{}, grow{};
    c64 *tw_col = nullptr, *tw_row = nullptr, *work = nullptr, *tau = nullptr;
    void *st_x = nullptr, *st_out = nullptr;
    double *mask = nullptr, *partial = nullptr, *sums = nullptr, *spart = nullptr;
    int* done = nullptr;
    size_t tau_cap = 0, sums_cap = 0;
    static constexpr int BLOCKS = 64;
    size_t per() const { return (size_t)nil * nxl; }
    // fused passes (col64_kernel / row64_kernel): tiles of ln_col columns / ln_row rows, the twiddle table in LDS where it fits
    bool fused = false;
    Fft64 fcol{}, frow{};
    int ln_col = 0, ln_row = 0, tw_col_lds = 0, tw_row_lds = 0, thr_col = F64_THREADS, thr_row = F64_THREADS;
    size_t lds_col = 0, lds_row = 0;
    // passes on the mixed-radix register engine (p3d_mix64.hip) where the axis' length has a plan: chosen per axis, same work buffer
    const p3d::mix64::Entry *mcol = nullptr, *mrow = nullptr;
    c64 *tw_mcol = nullptr, *tw_mrow = nullptr;
    unsigned char* nzflag = nullptr;   // both passes on the register engine: [max_slices][tiles_col] tile flags of the sparse shortcut
    bool sparse = false;               // ... in use for the job in progress
    // the observed cube and the result of the job in progress: the caller's own device buffers where it passed such, else the staging buffers
    const void* cur_x = nullptr;
    void* cur_out = nullptr;
    int tiles_col() const { return (nxl + ln_col - 1) / ln_col; }
    int tiles_row() const { return (nil + ln_row - 1) / ln_row; }
};

namespace {

int fft_pass(p3d_plan64* p, const c64* in, c64* out, int nslices, bool rows, int dir, double scale, const int* done, int group = 1)
{
    const GenPlan& pl = rows ? p->grow : p->gcol;
    const int n = pl.n;
    const size_t lines = (size_t)nslices * (rows ? p->nil : p->nxl);
    const size_t lds = sizeof(c64) * 2 * (size_t)n;
    if (lds > 64 * 1024) F_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(line_fft64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int threads = 64;
    while (threads < 256 && threads * 4 < n) threads *= 2;
    if (rows) line_fft64<<<dim3((unsigned)lines), threads, lds, p->stream>>>(in, out, p->tw_row, pl, dir, scale, 1, p->nil, p->per(), (size_t)p->nxl, done, p->nil * group);
    else line_fft64<<<dim3((unsigned)lines), threads, lds, p->stream>>>(in, out, p->tw_col, pl, dir, scale, p->nxl, p->nxl, p->per(), (size_t)1, done, p->nxl * group);
    F_TRY(hipGetLastError());
    return P3D_OK;
}

int fft2_64(p3d_plan64* p, int nslices, bool inverse, const int* done)
{
    int rc;
    if (!inverse) {
        if ((rc = fft_pass(p, p->work, p->work, nslices, true, -1, 1.0, done))) return rc;
        return fft_pass(p, p->work, p->work, nslices, false, -1, 1.0, done);
    }
    if ((rc = fft_pass(p, p->work, p->work, nslices, false, +1, 1.0 / p->nil, done))) return rc;
    return fft_pass(p, p->work, p->work, nslices, true, +1, 1.0 / p->nxl, done);
}

// tile of the fused passes for lines of n points: the largest of 4, 2, 1 lines whose two buffers (+ the table, where another 32 KiB allow it)
// stay below `budget` bytes of LDS; 0: none
int pick_tile64(int n, size_t budget, int want, int* tw_lds, size_t* lds)
{
    for (int ln : {8, 4, 2, 1}) {
        if (ln > want) continue;
        const size_t buf = sizeof(c64) * 2 * (size_t)ln * n, tab = sizeof(c64) * (size_t)n;
        if (buf > budget) continue;
        *tw_lds = 0;
        if (!getenv("P3D_F64_TW_GLOBAL")) {
            if (tab <= 32 * 1024 && buf + tab <= budget) *tw_lds = 1;
            else if (n % 2 == 0 && tab / 2 <= 32 * 1024 && buf + tab / 2 <= budget && !getenv("P3D_F64_NO_HALF_TABLE")) *tw_lds = 2;
        }
        *lds = buf + (*tw_lds == 1 ? tab : (*tw_lds == 2 ? tab / 2 : 0));
        return ln;
    }
    return 0;
}

template <int MODE>
int col_pass64(p3d_plan64* p, int nslices, int niter, int iter, int op, const int* done)
{
    if (p->mcol) {
        p3d::mix64::ColArgs64 a{};
        a.work = reinterpret_cast<p3d::mix::c64d*>(p->work); a.tab = reinterpret_cast<const p3d::mix::c64d*>(p->tw_mcol);
        a.n2 = p->nxl; a.nslices = nslices; a.tau = reinterpret_cast<const p3d::mix::c64d*>(p->tau); a.niter = niter; a.iter = iter; a.op = op;
        a.partial = p->partial; a.done = done;
        a.nzflag = (MODE == C64_ITER && p->sparse) ? p->nzflag : nullptr;
        F_TRY(p->mcol->col(MODE, a, p->stream));
        return P3D_OK;
    }
    auto kern = p->tw_col_lds ? col64_kernel<MODE, true> : col64_kernel<MODE, false>;
    if (p->lds_col > 64 * 1024) F_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_col));
    kern<<<dim3(p->tiles_col(), nslices), p->thr_col, p->lds_col, p->stream>>>(p->work, p->tw_col, p->fcol, p->nxl, p->ln_col, p->tw_col_lds, p->tau, niter, iter, op, p->partial, done);
    F_TRY(hipGetLastError());
    return P3D_OK;
}

template <int MODE>
int row_pass64(p3d_plan64* p, int dtype, double* sums_row, int adaptive, int write_out, double alpha, int nslices, const int* done, int zero_fill)
{
    if (p->mrow) {
        p3d::mix64::RowArgs64 a{};
        a.work = reinterpret_cast<p3d::mix::c64d*>(p->work); a.tab = reinterpret_cast<const p3d::mix::c64d*>(p->tw_mrow);
        a.n1 = p->nil; a.nslices = nslices; a.x = p->cur_x; a.dtype = dtype; a.mask = (MODE == R64_FIRST && !adaptive) ? nullptr : p->mask; a.out = p->cur_out;
        a.partial = p->spart; a.adaptive = adaptive; a.write_out = write_out; a.alpha = alpha; a.done = done; a.zero_fill = zero_fill;
        a.nzflag = (MODE != R64_FIRST && p->sparse) ? p->nzflag : nullptr; a.nz_tiles = p->tiles_col(); a.nz_col_t = p->ln_col;
        F_TRY(p->mrow->row(MODE, a, p->stream));
        fold64_kernel<<<nslices, 64, 0, p->stream>>>(p->spart, sums_row, nslices, p->tiles_row());
        F_TRY(hipGetLastError());
        return P3D_OK;
    }
    auto kern = p->tw_row_lds ? row64_kernel<MODE, true> : row64_kernel<MODE, false>;
    if (p->lds_row > 64 * 1024) F_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_row));
    kern<<<dim3(p->tiles_row(), nslices), p->thr_row, p->lds_row, p->stream>>>(p->work, p->tw_row, p->frow, p->nil, p->ln_row, p->tw_row_lds, p->cur_x, dtype,
                                                                              (MODE == R64_FIRST && !adaptive) ? nullptr : p->mask, p->cur_out, p->spart, adaptive, write_out, alpha,
                                                                              done, zero_fill);
    fold64_kernel<<<nslices, 64, 0, p->stream>>>(p->spart, sums_row, nslices, p->tiles_row());
    F_TRY(hipGetLastError());
    return P3D_OK;
}

size_t esize(int dtype) { return dtype == P3D_C128 ? 16 : (dtype == P3D_F64 || dtype == P3D_C64 ? 8 : 4); }

// a pointer into the memory of the plan's own device (the entry points take host or device pointers)
bool on_plan_device(const p3d_plan64* p, const void* ptr)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();   // ordinary host memory
        return false;
    }
    return at.type == hipMemoryTypeDevice && at.device == p->device;
}

int check64(p3d_plan64* p, int nslices, int dtype)
{
    if (!p) return f64fail(P3D_ERR_INVALID, "NULL plan");
    if (nslices < 1 || nslices > p->max_slices) return f64fail(P3D_ERR_INVALID, "nslices = %d outside 1..max_slices (%d)", nslices, p->max_slices);
    if (dtype != P3D_C128 && dtype != P3D_F64 && dtype != P3D_C64 && dtype != P3D_F32) return f64fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    F_TRY(hipSetDevice(p->device));
    return P3D_OK;
}

int update(p3d_plan64* p, int dtype, double* sums_row, int mode, int adaptive, int write_out, double alpha, int nslices, const int* done, int zero_fill)
{
    update64_kernel<<<dim3(p3d_plan64::BLOCKS, nslices), 256, 0, p->stream>>>(p->work, p->cur_x, dtype, mode == 0 && !adaptive ? nullptr : p->mask, p->cur_out, p->spart, mode,
                                                                              adaptive, write_out, alpha, p->per(), done, zero_fill);
    fold64_kernel<<<nslices, 64, 0, p->stream>>>(p->spart, sums_row, nslices, p3d_plan64::BLOCKS);
    F_TRY(hipGetLastError());
    return P3D_OK;
}

}  // namespace

extern "C" {

int p3d_plan64_destroy(p3d_plan64* p)
{
    if (!p) return P3D_OK;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    void* bufs[] = {p->nzflag, p->tw_mcol, p->tw_mrow, p->tw_col, p->tw_row, p->work, p->tau, p->st_x, p->st_out, p->mask, p->partial, p->sums, p->spart, p->done};
    for (void* b : bufs)
        if (b) hipFree(b);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return P3D_OK;
}

// bare: twiddles and the work buffer only (a plan whose transforms another loop borrows, p3d_shearlet64.hip)
static int create64(p3d_plan64** out, int device, int nil, int nxl, int max_slices, bool bare)
{
    if (!out) return f64fail(P3D_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (nil < 1 || nxl < 1 || max_slices < 1 || max_slices > 65535) return f64fail(P3D_ERR_INVALID, "nil, nxl, max_slices must be positive (max_slices <= 65535)");
    if (nil > F64_MAX_N || nxl > F64_MAX_N) return f64fail(P3D_ERR_UNSUPPORTED, "double-precision path: extents up to %d (got %d x %d)", F64_MAX_N, nil, nxl);
    int ndev = 0;
    F_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return f64fail(P3D_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    F_TRY(hipSetDevice(device));
    p3d_plan64* p = new p3d_plan64;
    p->device = device; p->nil = nil; p->nxl = nxl; p->max_slices = max_slices;
    p->gcol = p3d::gen_make_plan(nil);
    p->grow = p3d::gen_make_plan(nxl);
    if (p->gcol.nf < 0 || p->grow.nf < 0) { delete p; return f64fail(P3D_ERR_UNSUPPORTED, "slice shape %d x %d cannot be factorised", nil, nxl); }
    auto bail = [&](const char* what, hipError_t e) {
        f64fail(P3D_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        std::string keep = p3d_last_error();
        p3d_plan64_destroy(p);
        p3d::set_last_error(keep.c_str());
        return P3D_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess) return bail("stream", e);
    if ((e = hipEventCreate(&p->ev0)) != hipSuccess || (e = hipEventCreate(&p->ev1)) != hipSuccess) return bail("event", e);
    for (int which = 0; which < 2; ++which) {
        const int n = which ? nxl : nil;
        std::vector<c64> host(n);
        for (int k = 0; k < n; ++k) {
            const long double ang = -6.283185307179586476925286766559005768L * (long double)k / (long double)n;
            host[k] = c64{(double)cosl(ang), (double)sinl(ang)};
        }
        c64** dst = which ? &p->tw_row : &p->tw_col;
        if ((e = hipMalloc((void**)dst, sizeof(c64) * n)) != hipSuccess) return bail("twiddles", e);
        if ((e = hipMemcpy(*dst, host.data(), sizeof(c64) * n, hipMemcpyHostToDevice)) != hipSuccess) return bail("twiddles", e);
    }
    // the fused passes where two buffers of a tile fit LDS (rows: two workgroups per CU if that leaves a tile of two rows)
    p->fcol = make_fft64(nil);
    p->frow = make_fft64(nxl);
    {
        const size_t whole = 160 * 1024 - 1024, half = 80 * 1024 - 1024;
        auto knob = [](const char* name, int dflt) { const char* v = getenv(name); return v && atoi(v) > 0 ? atoi(v) : dflt; };
        const size_t first = getenv("P3D_F64_ONE_PER_CU") ? whole : half;
        p->ln_col = pick_tile64(nil, first, knob("P3D_F64_COL_TILE", 2), &p->tw_col_lds, &p->lds_col);
        if (p->ln_col == 0) p->ln_col = pick_tile64(nil, whole, knob("P3D_F64_COL_TILE", 2), &p->tw_col_lds, &p->lds_col);
        p->ln_row = pick_tile64(nxl, first, knob("P3D_F64_ROW_TILE", 2), &p->tw_row_lds, &p->lds_row);
        if (p->ln_row == 0) p->ln_row = pick_tile64(nxl, whole, knob("P3D_F64_ROW_TILE", 2), &p->tw_row_lds, &p->lds_row);
        auto pow2 = [](int v) { int t = 64; while (2 * t <= v && 2 * t <= F64_THREADS) t *= 2; return t; };   // (the passes split threads by shifts)
        p->thr_col = pow2(knob("P3D_F64_COL_THREADS", F64_THREADS));
        p->thr_row = pow2(knob("P3D_F64_ROW_THREADS", F64_THREADS));
        p->fused = p->ln_col > 0 && p->ln_row > 0 && p->fcol.nf > 0 && p->frow.nf > 0 && !getenv("P3D_F64_UNFUSED");
        if (!p->fused) p->ln_col = p->ln_row = 1;
    }
    if (p->fused) {
        // the register-engine passes where the axis has a plan (p3d_mix64.hip): their tiles define the partial-sum layouts
        p->mcol = p3d::mix64::find(nil);
        p->mrow = p3d::mix64::find(nxl);
        for (int which = 0; which < 2; ++which) {
            const p3d::mix64::Entry* en = which ? p->mrow : p->mcol;
            if (!en) continue;
            std::vector<c64> host((size_t)en->tw_slots + 1);
            en->build_tw(reinterpret_cast<p3d::mix::c64d*>(host.data()));
            c64** dst = which ? &p->tw_mrow : &p->tw_mcol;
            if ((e = hipMalloc((void**)dst, sizeof(c64) * host.size())) != hipSuccess) return bail("twiddles", e);
            if ((e = hipMemcpy(*dst, host.data(), sizeof(c64) * host.size(), hipMemcpyHostToDevice)) != hipSuccess) return bail("twiddles", e);
            if (which) p->ln_row = en->row_lines; else p->ln_col = en->col_tile;
        }
    }
    const size_t S = (size_t)max_slices, per = p->per();
    const size_t nparts = (size_t)std::max(std::max(p3d_plan64::BLOCKS, p->tiles_col()), p->tiles_row());
#define ALLOC64(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return bail(#ptr, e)
    ALLOC64(p->work, sizeof(c64) * per * S);
    if (!bare) {
        ALLOC64(p->st_x, sizeof(c64) * per * S);
        ALLOC64(p->st_out, sizeof(c64) * per * S);
    }
    ALLOC64(p->mask, sizeof(double) * per);
    ALLOC64(p->partial, sizeof(double) * 8 * nparts * S);
    ALLOC64(p->spart, sizeof(double) * nparts * S);
    ALLOC64(p->done, sizeof(int) * S);
    if (p->mcol && p->mrow && !getenv("P3D_F64_NO_SPARSE")) ALLOC64(p->nzflag, (size_t)p->tiles_col() * S);
#undef ALLOC64
    *out = p;
    return P3D_OK;
}

int p3d_plan64_create(p3d_plan64** out, int device, int nil, int nxl, int max_slices) { return create64(out, device, nil, nxl, max_slices, false); }

// statistics of fft2(x) for the schedule: stats[nslices][P3D_STATS_PER_SLICE] as p3d_pocs_stats (x: host or device pointer)
int p3d_pocs64_stats(p3d_plan64* p, const void* x, int dtype, int nslices, double* stats)
{
    int rc = check64(p, nslices, dtype);
    if (rc) return rc;
    if (!x || !stats) return f64fail(P3D_ERR_INVALID, "NULL buffer");
    if (on_plan_device(p, x)) {
        p->cur_x = x;
    } else {
        F_TRY(hipMemcpyAsync(p->st_x, x, esize(dtype) * p->per() * nslices, hipMemcpyDefault, p->stream));
        p->cur_x = p->st_x;
    }
    p->cur_out = p->st_out;
    p->sparse = false;
    const int nblocks = p->fused ? p->tiles_col() : p3d_plan64::BLOCKS;
    if (p->fused) {
        if ((rc = row_pass64<R64_FIRST>(p, dtype, p->partial, 0, 0, 1.0, nslices, nullptr, 0))) return rc;   // rows of x (its sums go to a scratch row)
        if ((rc = col_pass64<C64_STATS>(p, nslices, 0, 0, 0, nullptr))) return rc;
    } else {
        if ((rc = update(p, dtype, p->partial, 0, 0, 0, 1.0, nslices, nullptr, 0))) return rc;   // w = x (its sums go to a scratch row)
        if ((rc = fft2_64(p, nslices, false, nullptr))) return rc;
        stats64_kernel<<<dim3(p3d_plan64::BLOCKS, nslices), 256, 0, p->stream>>>(p->work, p->partial, p->per());
        F_TRY(hipGetLastError());
    }
    std::vector<double> host((size_t)nslices * nblocks * 8);
    F_TRY(hipMemcpyAsync(host.data(), p->partial, sizeof(double) * host.size(), hipMemcpyDeviceToHost, p->stream));
    F_TRY(hipStreamSynchronize(p->stream));
    for (int s = 0; s < nslices; ++s) {
        double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY, sq = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            const double* q = &host[((size_t)s * nblocks + b) * 8];
            if (q[0] > lr || (q[0] == lr && q[1] > li)) { lr = q[0]; li = q[1]; }
            mx = std::fmax(mx, q[2]);
            mn = std::fmin(mn, q[3]);
            sq += q[4];
        }
        double* o = stats + (size_t)s * P3D_STATS_PER_SLICE;
        o[0] = lr; o[1] = li; o[2] = mx; o[3] = mn; o[4] = sq; o[5] = 0.0;
    }
    return P3D_OK;
}

// the loop of POCS_algorithm (POCS.py:560-632) in double precision.  x, out: host or device, dtype as above (x and out alike); mask: host or
// device double [nil][nxl]; tau: host double [nslices][niter][2] (Re, Im); sums: host double [niter + 1][nslices] or NULL
int p3d_pocs64_run(p3d_plan64* p, const void* x, int dtype, const double* mask, const double* tau, const uint8_t* active, const p3d_pocs_params* prm, void* out,
                   int nslices, int32_t* niter_done, double* sums, double* elapsed_ms)
{
    int rc = check64(p, nslices, dtype);
    if (rc) return rc;
    if (!x || !mask || !tau || !prm || !out) return f64fail(P3D_ERR_INVALID, "NULL argument");
    if (prm->niter < 1) return f64fail(P3D_ERR_INVALID, "niter must be >= 1");
    if (prm->thresh_op < P3D_OP_HARD || prm->thresh_op > P3D_OP_GARROTE) return f64fail(P3D_ERR_UNSUPPORTED, "thresh_op %d: the double-precision path has hard, soft and garrote", prm->thresh_op);
    if (prm->version < P3D_VER_REGULAR || prm->version > P3D_VER_ADAPTIVE) return f64fail(P3D_ERR_INVALID, "unknown version %d", prm->version);
    const int niter = prm->niter;
    const bool early = prm->eps > 0.0, adaptive = prm->version == P3D_VER_ADAPTIVE;
    const size_t ntau = (size_t)nslices * niter, nsum = (size_t)(niter + 1) * nslices, per = p->per();
    if (p->tau_cap < ntau) {
        if (p->tau) hipFree(p->tau);
        p->tau = nullptr; p->tau_cap = 0;
        F_TRY(hipMalloc((void**)&p->tau, sizeof(c64) * ntau));
        p->tau_cap = ntau;
    }
    if (p->sums_cap < nsum) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr; p->sums_cap = 0;
        F_TRY(hipMalloc((void**)&p->sums, sizeof(double) * nsum));
        p->sums_cap = nsum;
    }
    std::vector<int> done_h(nslices, 0);
    bool any_off = early;
    if (active) for (int s = 0; s < nslices; ++s) { done_h[s] = active[s] ? 0 : -1; any_off = any_off || !active[s]; }
    const int* done_d = any_off ? p->done : nullptr;
    // the caller's own device buffers where it passed such (no staging copies: 80 bytes per point and two passes less for a resident batch);
    // a result buffer that overlaps the observed cube -- read in every iteration -- goes through the staging buffer
    const size_t cube_bytes = esize(dtype) * per * nslices;
    if (on_plan_device(p, x)) {
        p->cur_x = x;
    } else {
        F_TRY(hipMemcpyAsync(p->st_x, x, cube_bytes, hipMemcpyDefault, p->stream));
        p->cur_x = p->st_x;
    }
    const char* const xb = static_cast<const char*>(x);
    char* const ob = static_cast<char*>(out);
    const bool direct_out = on_plan_device(p, out) && (ob + cube_bytes <= xb || xb + cube_bytes <= ob);
    p->cur_out = direct_out ? out : p->st_out;
    F_TRY(hipMemcpyAsync(p->mask, mask, sizeof(double) * per, hipMemcpyDefault, p->stream));
    F_TRY(hipMemcpyAsync(p->tau, tau, sizeof(c64) * ntau, hipMemcpyHostToDevice, p->stream));   // (Re, Im) pairs of doubles: c64's layout
    F_TRY(hipMemcpyAsync(p->done, done_h.data(), sizeof(int) * nslices, hipMemcpyHostToDevice, p->stream));
    F_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nsum, p->stream));
    F_TRY(hipEventRecord(p->ev0, p->stream));
    p->sparse = p->nzflag != nullptr;   // (tiles of the spectrum that the threshold empties are neither transformed back, stored nor read again: exact)
    if (p->fused) {
        // two kernels per iteration: rows (inverse transform, re-insertion, forward transform), columns (forward, threshold, inverse)
        if ((rc = row_pass64<R64_FIRST>(p, dtype, p->sums, adaptive ? 1 : 0, 0, prm->alpha, nslices, done_d, 0))) return rc;
        for (int k = 0; k < niter; ++k) {
            const bool last = k + 1 == niter;
            if ((rc = col_pass64<C64_ITER>(p, nslices, niter, k, prm->thresh_op, done_d))) return rc;
            double* srow = p->sums + (size_t)(k + 1) * nslices;
            if (last) rc = row_pass64<R64_LAST>(p, dtype, srow, 0, 1, prm->alpha, nslices, p->done, 1);
            else rc = row_pass64<R64_MID>(p, dtype, srow, adaptive ? 1 : 0, early ? 1 : 0, prm->alpha, nslices, p->done, 0);
            if (rc) return rc;
            if (early) conv64_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
        }
    } else {
    if ((rc = update(p, dtype, p->sums, 0, adaptive ? 1 : 0, 0, prm->alpha, nslices, done_d, 0))) return rc;
    for (int k = 0; k < niter; ++k) {
        const bool last = k + 1 == niter;
        if ((rc = fft2_64(p, nslices, false, done_d))) return rc;
        shrink64_kernel<<<dim3(p3d_plan64::BLOCKS * 4, nslices), 256, 0, p->stream>>>(p->work, p->tau, niter, k, prm->thresh_op, per, done_d);
        if ((rc = fft2_64(p, nslices, true, done_d))) return rc;
        if ((rc = update(p, dtype, p->sums + (size_t)(k + 1) * nslices, 1, (adaptive && !last) ? 1 : 0, (early || last) ? 1 : 0, prm->alpha, nslices, p->done,
                         last ? 1 : 0)))
            return rc;
        if (early) conv64_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
    }
    }
    F_TRY(hipGetLastError());
    F_TRY(hipEventRecord(p->ev1, p->stream));
    F_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
    if (sums) F_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
    if (!direct_out) F_TRY(hipMemcpyAsync(out, p->st_out, cube_bytes, hipMemcpyDefault, p->stream));
    F_TRY(hipStreamSynchronize(p->stream));
    if (niter_done) for (int s = 0; s < nslices; ++s) niter_done[s] = done_h[s] < 0 ? 0 : (done_h[s] > 0 ? done_h[s] : niter);
    if (elapsed_ms) {
        float ms = 0.f;
        F_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        *elapsed_ms = ms;
    }
    return P3D_OK;
}

// test hook: batched fft2 / ifft2 (numpy.fft conventions) of HOST complex128 slices THROUGH THE LOOP'S OWN PASSES -- forward = the first row pass
// (rows of x) + the forward column pass; inverse = the inverse column pass + the last row pass on an all-zero observed cube with an all-zero
// mask (its re-insertion then adds nothing) -- so that every axis length is checked element by element on the engine the loop uses for it
int p3d_fft2_c128(p3d_plan64* p, const void* in_host, void* out_host, int nslices, int inverse)
{
    int rc = check64(p, nslices, P3D_C128);
    if (rc) return rc;
    if (!in_host || !out_host) return f64fail(P3D_ERR_INVALID, "NULL buffer");
    const size_t bytes = sizeof(c64) * p->per() * nslices;
    p->sparse = false;
    if (!inverse) {
        F_TRY(hipMemcpyAsync(p->st_x, in_host, bytes, hipMemcpyHostToDevice, p->stream));
        p->cur_x = p->st_x;
        p->cur_out = p->st_out;
        if (p->fused) {
            if ((rc = row_pass64<R64_FIRST>(p, P3D_C128, p->partial, 0, 0, 1.0, nslices, nullptr, 0))) return rc;
            if ((rc = col_pass64<C64_FWD>(p, nslices, 0, 0, 0, nullptr))) return rc;
        } else {
            if ((rc = update(p, P3D_C128, p->partial, 0, 0, 0, 1.0, nslices, nullptr, 0))) return rc;
            if ((rc = fft2_64(p, nslices, false, nullptr))) return rc;
        }
        F_TRY(hipMemcpyAsync(out_host, p->work, bytes, hipMemcpyDeviceToHost, p->stream));
    } else {
        F_TRY(hipMemcpyAsync(p->work, in_host, bytes, hipMemcpyHostToDevice, p->stream));
        if (p->fused) {
            F_TRY(hipMemsetAsync(p->st_x, 0, bytes, p->stream));
            F_TRY(hipMemsetAsync(p->mask, 0, sizeof(double) * p->per(), p->stream));
            p->cur_x = p->st_x;
            p->cur_out = p->st_out;
            if (p->mcol) rc = p3d::plan64_shear_cols(p, nullptr, nullptr, nslices, 1, 0, 0, 0, 0, 1, 1.0 / p->nil, nullptr, nullptr, 0);
            else rc = fft_pass(p, p->work, p->work, nslices, false, +1, 1.0 / p->nil, nullptr);
            if (rc) return rc;
            if ((rc = row_pass64<R64_LAST>(p, P3D_C128, p->partial, 0, 1, 1.0, nslices, nullptr, 0))) return rc;
            F_TRY(hipMemcpyAsync(out_host, p->st_out, bytes, hipMemcpyDeviceToHost, p->stream));
        } else {
            if ((rc = fft2_64(p, nslices, true, nullptr))) return rc;
            F_TRY(hipMemcpyAsync(out_host, p->work, bytes, hipMemcpyDeviceToHost, p->stream));
        }
    }
    F_TRY(hipStreamSynchronize(p->stream));
    return P3D_OK;
}

}  // extern "C"

// ---- what p3d_shearlet64.hip borrows (p3d_internal.hpp) ------------------------------------------------------------------------------------------------
namespace p3d {

int plan64_create_bare(p3d_plan64** out, int device, int nil, int nxl, int max_slices) { return create64(out, device, nil, nxl, max_slices, true); }
hipStream_t plan64_stream(p3d_plan64* p) { return p->stream; }
void* plan64_work(p3d_plan64* p) { return p->work; }

bool plan64_shear_supported(p3d_plan64* p) { return p->fused && p->mcol && p->mrow && p->st_x != nullptr; }
bool plan64_engine_shape(int nil, int nxl)
{
    return nil >= 1 && nxl >= 1 && nil <= F64_MAX_N && nxl <= F64_MAX_N && !getenv("P3D_F64_UNFUSED") && p3d::mix64::find(nil) != nullptr && p3d::mix64::find(nxl) != nullptr;
}
double* plan64_mask(p3d_plan64* p) { return p->mask; }
void* plan64_stage_x(p3d_plan64* p) { return p->st_x; }
void* plan64_stage_out(p3d_plan64* p) { return p->st_out; }
void plan64_bind(p3d_plan64* p, const void* x, void* out)
{
    p->cur_x = x;
    p->cur_out = out;
    p->sparse = false;
}

int plan64_shear_first(p3d_plan64* p, int dtype, double* sums_row, int adaptive, double alpha, int nslices, const int* done)
{
    int rc = row_pass64<R64_FIRST>(p, dtype, sums_row, adaptive, 0, alpha, nslices, done, 0);
    if (rc) return rc;
    return col_pass64<C64_FWD>(p, nslices, 0, 0, 0, done);
}

int plan64_shear_row_group(p3d_plan64* p) { return p->mrow ? p->mrow->row_lines : 0; }

int plan64_shear_spread(p3d_plan64* p, const double* psi, void* U, int nb, int nsh, const int* done, const unsigned char* sup, int rows)
{
    p3d::mix64::SpreadRow64 a{};
    a.F = reinterpret_cast<const p3d::mix::c64d*>(p->work); a.psi = psi; a.U = reinterpret_cast<p3d::mix::c64d*>(U);
    a.tab = reinterpret_cast<const p3d::mix::c64d*>(p->tw_mrow); a.n1 = p->nil; a.nb = nb; a.nsh = nsh; a.done = done; a.sup = sup;
    a.sup_groups = (p->nil + p->mrow->row_lines - 1) / p->mrow->row_lines; a.rows = rows > 0 ? rows : p->nil;
    F_TRY(p->mrow->spread_row(a, p->stream));
    return P3D_OK;
}

int plan64_shear_cols(p3d_plan64* p, void* U, const void* tau, int nb, int nsh, int niter, int iter, int op, int real_only, int mode, double scale, const int* done,
                      const unsigned char* sup, int pair)
{
    p3d::mix64::ShearCol64 a{};
    a.U = reinterpret_cast<p3d::mix::c64d*>(U ? U : p->work); a.tab = reinterpret_cast<const p3d::mix::c64d*>(p->tw_mcol);
    a.n2 = p->nxl; a.nslices = nb * nsh; a.nsh = nsh; a.tau = reinterpret_cast<const p3d::mix::c64d*>(tau); a.niter = niter; a.iter = iter; a.op = op;
    a.real_only = real_only; a.mode = mode; a.scale = scale; a.done = done;
    a.sup = sup; a.sup_rows = p->mrow->row_lines; a.sup_groups = (p->nil + a.sup_rows - 1) / a.sup_rows; a.pair = pair;
    F_TRY(p->mcol->shear_col(a, p->stream));
    return P3D_OK;
}

int plan64_shear_gather(p3d_plan64* p, const void* U, const double* psi, int nb, int nsh, const int* done, const unsigned char* sup, int rows)
{
    p3d::mix64::GatherRow64 a{};
    a.U = reinterpret_cast<const p3d::mix::c64d*>(U); a.psi = psi; a.F = reinterpret_cast<p3d::mix::c64d*>(p->work);
    a.tab = reinterpret_cast<const p3d::mix::c64d*>(p->tw_mrow); a.n1 = p->nil; a.nb = nb; a.nsh = nsh; a.done = done; a.sup = sup;
    a.sup_groups = (p->nil + p->mrow->row_lines - 1) / p->mrow->row_lines; a.rows = rows > 0 ? rows : p->nil;
    F_TRY(p->mrow->gather_row(a, p->stream));
    return P3D_OK;
}

int plan64_shear_back(p3d_plan64* p, int dtype, double* sums_row, bool last, int adaptive, int write_out, double alpha, int nslices, const int* done, int zero_fill)
{
    int rc = plan64_shear_cols(p, nullptr, nullptr, nslices, 1, 0, 0, 0, 0, 1, 1.0 / p->nil, done, nullptr, 0);   // (the row pass divides by nxl)
    if (rc) return rc;
    if (last) return row_pass64<R64_LAST>(p, dtype, sums_row, 0, 1, alpha, nslices, done, zero_fill);
    if ((rc = row_pass64<R64_MID>(p, dtype, sums_row, adaptive, write_out, alpha, nslices, done, zero_fill))) return rc;
    return col_pass64<C64_FWD>(p, nslices, 0, 0, 0, done);
}

int plan64_fft2(p3d_plan64* p, void* buf, int nslices, bool inverse, const int* done, int done_group)
{
    c64* w = reinterpret_cast<c64*>(buf);
    int rc;
    if (!inverse) {
        if ((rc = fft_pass(p, w, w, nslices, true, -1, 1.0, done, done_group))) return rc;
        return fft_pass(p, w, w, nslices, false, -1, 1.0, done, done_group);
    }
    if ((rc = fft_pass(p, w, w, nslices, false, +1, 1.0 / p->nil, done, done_group))) return rc;
    return fft_pass(p, w, w, nslices, true, +1, 1.0 / p->nxl, done, done_group);
}

}  // namespace p3d
