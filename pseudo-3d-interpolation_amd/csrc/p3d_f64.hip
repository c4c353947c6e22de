// p3d_f64.hip -- the POCS loop in the REFERENCE's own precision: complex128 / float64 cubes, and complex64 / float32 cubes on request.
//
// The reference computes most of its configurations in double precision: under NumPy < 2 every path (np.fft returns complex128), under
// NumPy >= 2 everything but regular / hard on a complex64 cube -- the soft and garrote operators promote at the first threshold (tau is a
// complex128 scalar taken from an array: threshold_operator.py:37-39, 76-78), FPOCS through its float64 momentum scalar (POCS.py:566-571),
// APOCS and every alpha-weighted update through the float64 weight array 1 - alpha * mask (POCS.py:572-575, 616).  The float32 kernels of
// this library reproduce those runs to float32 rounding only, and the operators are discontinuous at |X| = Re tau (soft and garrote jump by
// |Im tau| there): 1e-4 instead of 1e-5 on ill-conditioned slices.  This file is the same loop in double precision.
// It is built from the plain pieces of the any-length pipeline (p3d_generic.hip) -- an LDS-resident mixed-radix Stockham line transform with
// run-time factors and element-wise passes, six passes over the cube per iteration instead of two fused ones: a precision path, not a
// fast one (rates in DESIGN.md section 5).  Any line length up to 5120 (two double-precision copies of a line in 160 KiB of LDS).
//
// Entry points (include/p3d.h): p3d_plan64_create / _destroy, p3d_pocs64_stats, p3d_pocs64_run.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "p3d.h"
#include "p3d_generic.hpp"
#include "p3d_internal.hpp"

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

// per slice: sums[s] = the blocks' partial sums added in order
__global__ void fold64_kernel(const double* partial, double* sums, int nslices, int blocks)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    double t = 0.0;
    for (int b = 0; b < blocks; ++b) t += partial[(size_t)s * blocks + b];
    sums[s] = t;
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
};

namespace {

int fft_pass(p3d_plan64* p, const c64* in, c64* out, int nslices, bool rows, int dir, double scale, const int* done)
{
    const GenPlan& pl = rows ? p->grow : p->gcol;
    const int n = pl.n;
    const size_t lines = (size_t)nslices * (rows ? p->nil : p->nxl);
    const size_t lds = sizeof(c64) * 2 * (size_t)n;
    if (lds > 64 * 1024) F_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(line_fft64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int threads = 64;
    while (threads < 256 && threads * 4 < n) threads *= 2;
    if (rows) line_fft64<<<dim3((unsigned)lines), threads, lds, p->stream>>>(in, out, p->tw_row, pl, dir, scale, 1, p->nil, p->per(), (size_t)p->nxl, done, p->nil);
    else line_fft64<<<dim3((unsigned)lines), threads, lds, p->stream>>>(in, out, p->tw_col, pl, dir, scale, p->nxl, p->nxl, p->per(), (size_t)1, done, p->nxl);
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

size_t esize(int dtype) { return dtype == P3D_C128 ? 16 : (dtype == P3D_F64 || dtype == P3D_C64 ? 8 : 4); }

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
    update64_kernel<<<dim3(p3d_plan64::BLOCKS, nslices), 256, 0, p->stream>>>(p->work, p->st_x, dtype, mode == 0 && !adaptive ? nullptr : p->mask, p->st_out, p->spart, mode,
                                                                              adaptive, write_out, alpha, p->per(), done, zero_fill);
    fold64_kernel<<<(nslices + 63) / 64, 64, 0, p->stream>>>(p->spart, sums_row, nslices, p3d_plan64::BLOCKS);
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
    void* bufs[] = {p->tw_col, p->tw_row, p->work, p->tau, p->st_x, p->st_out, p->mask, p->partial, p->sums, p->spart, p->done};
    for (void* b : bufs)
        if (b) hipFree(b);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return P3D_OK;
}

int p3d_plan64_create(p3d_plan64** out, int device, int nil, int nxl, int max_slices)
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
    const size_t S = (size_t)max_slices, per = p->per();
#define ALLOC64(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return bail(#ptr, e)
    ALLOC64(p->work, sizeof(c64) * per * S);
    ALLOC64(p->st_x, sizeof(c64) * per * S);
    ALLOC64(p->st_out, sizeof(c64) * per * S);
    ALLOC64(p->mask, sizeof(double) * per);
    ALLOC64(p->partial, sizeof(double) * 8 * p3d_plan64::BLOCKS * S);
    ALLOC64(p->spart, sizeof(double) * p3d_plan64::BLOCKS * S);
    ALLOC64(p->done, sizeof(int) * S);
#undef ALLOC64
    *out = p;
    return P3D_OK;
}

// statistics of fft2(x) for the schedule: stats[nslices][P3D_STATS_PER_SLICE] as p3d_pocs_stats (x: host or device pointer)
int p3d_pocs64_stats(p3d_plan64* p, const void* x, int dtype, int nslices, double* stats)
{
    int rc = check64(p, nslices, dtype);
    if (rc) return rc;
    if (!x || !stats) return f64fail(P3D_ERR_INVALID, "NULL buffer");
    F_TRY(hipMemcpyAsync(p->st_x, x, esize(dtype) * p->per() * nslices, hipMemcpyDefault, p->stream));
    if ((rc = update(p, dtype, p->partial, 0, 0, 0, 1.0, nslices, nullptr, 0))) return rc;   // w = x (its sums go to a scratch row)
    if ((rc = fft2_64(p, nslices, false, nullptr))) return rc;
    stats64_kernel<<<dim3(p3d_plan64::BLOCKS, nslices), 256, 0, p->stream>>>(p->work, p->partial, p->per());
    F_TRY(hipGetLastError());
    std::vector<double> host((size_t)nslices * p3d_plan64::BLOCKS * 8);
    F_TRY(hipMemcpyAsync(host.data(), p->partial, sizeof(double) * host.size(), hipMemcpyDeviceToHost, p->stream));
    F_TRY(hipStreamSynchronize(p->stream));
    for (int s = 0; s < nslices; ++s) {
        double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY, sq = 0.0;
        for (int b = 0; b < p3d_plan64::BLOCKS; ++b) {
            const double* q = &host[((size_t)s * p3d_plan64::BLOCKS + b) * 8];
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
    F_TRY(hipMemcpyAsync(p->st_x, x, esize(dtype) * per * nslices, hipMemcpyDefault, p->stream));
    F_TRY(hipMemcpyAsync(p->mask, mask, sizeof(double) * per, hipMemcpyDefault, p->stream));
    F_TRY(hipMemcpyAsync(p->tau, tau, sizeof(c64) * ntau, hipMemcpyHostToDevice, p->stream));   // (Re, Im) pairs of doubles: c64's layout
    F_TRY(hipMemcpyAsync(p->done, done_h.data(), sizeof(int) * nslices, hipMemcpyHostToDevice, p->stream));
    F_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nsum, p->stream));
    F_TRY(hipEventRecord(p->ev0, p->stream));
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
    F_TRY(hipGetLastError());
    F_TRY(hipEventRecord(p->ev1, p->stream));
    F_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
    if (sums) F_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
    F_TRY(hipMemcpyAsync(out, p->st_out, esize(dtype) * per * nslices, hipMemcpyDefault, p->stream));
    F_TRY(hipStreamSynchronize(p->stream));
    if (niter_done) for (int s = 0; s < nslices; ++s) niter_done[s] = done_h[s] < 0 ? 0 : (done_h[s] > 0 ? done_h[s] : niter);
    if (elapsed_ms) {
        float ms = 0.f;
        F_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        *elapsed_ms = ms;
    }
    return P3D_OK;
}

}  // extern "C"
