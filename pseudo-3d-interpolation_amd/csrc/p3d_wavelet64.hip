// p3d_wavelet64.hip -- the WAVELET variant of the POCS loop in the REFERENCE's double precision.
//
// pywt.wavedec2 / waverec2 keep float64 for float64 input and the reference's loop never narrows (functions/POCS.py:585-588, 596-597,
// 608-609; threshold_wavelet POCS.py:105-166; the driver casts to the input dtype only at the end, cube_POCS_interpolation_3D.py:324).
// The float32 kernels of p3d_wavelet.hip follow PyWavelets' float32 arithmetic for float32 cubes; on long schedules the 'smooth'
// boundary extension makes the iteration expansive (the iterate grows by orders of magnitude: DESIGN.md section 4), and float32 rounding
// grows with it -- in the reference's own float32 run as much as on the device.  This file is the loop for complex128 / float64 cubes and
// for complex64 / float32 cubes on request (precision='reference'): the same published algorithm (p3d_wavelet.hip's header; restated in
// NumPy in oracle/wavelet_oracle.py, pinned on PyWavelets 1.1.1 and on reference runs) with every sample, tap, threshold, weight and
// statistic in double precision.
//
// A precision path: one thread per output sample and axis (the per-axis kernels of p3d_wavelet.hip), no LDS tiles -- three launches per
// level and direction.  Rate and roofline: DESIGN.md section 5 / bench.py `reference_precision`.
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

namespace {

constexpr int MAXL = 128;   // longest filter (PyWavelets: coif17 has 102 taps)

struct c64 {
    double x, y;
};
struct Filters64 {
    double dec_lo[MAXL], dec_hi[MAXL], rec_lo[MAXL], rec_hi[MAXL];
    int len;
};

__device__ __forceinline__ c64 operator+(c64 a, c64 b) { return c64{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c64 operator-(c64 a, c64 b) { return c64{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c64 scale(c64 a, double s) { return c64{a.x * s, a.y * s}; }
__device__ __forceinline__ double scale(double a, double s) { return a * s; }
__device__ __forceinline__ void acc_tap(c64& acc, double f, c64 v) { acc.x = fma(f, v.x, acc.x); acc.y = fma(f, v.y, acc.y); }
__device__ __forceinline__ void acc_tap(double& acc, double f, double v) { acc = fma(f, v, acc); }
__device__ __forceinline__ double mag(c64 v) { return hypot(v.x, v.y); }   // numpy's abs() of a complex128
__device__ __forceinline__ double mag(double v) { return fabs(v); }
__device__ __forceinline__ double re_of(c64 v) { return v.x; }
__device__ __forceinline__ double im_of(c64 v) { return v.y; }
__device__ __forceinline__ double re_of(double v) { return v; }
__device__ __forceinline__ double im_of(double) { return 0.0; }
template <typename T> __device__ __forceinline__ T zero_of() { return T{}; }
// straight line through the edge pair (e = edge sample, f = its neighbour), t samples beyond the edge ('smooth')
__device__ __forceinline__ c64 extrapolate(c64 e, c64 f, double t) { return c64{e.x + (e.x - f.x) * t, e.y + (e.y - f.y) * t}; }
__device__ __forceinline__ double extrapolate(double e, double f, double t) { return e + (e - f) * t; }

// ---- thresholds (threshold_operator.py:9-112 / pywt._thresholding; tau complex because the schedule is scaled by numpy's lexicographic
// complex max, POCS.py:281 -- comparisons and clipping against it are lexicographic) ---------------------------------------------------
__device__ __forceinline__ c64 shrink(c64 X, c64 tau, int op)
{
    const double m = mag(X);
    if (op == 0) {   // hard: where(|X| < tau, 0, X)
        const bool below = m < tau.x || (m == tau.x && 0.0 < tau.y);
        return below ? c64{0.0, 0.0} : X;
    }
    if (m == 0.0) return c64{0.0, 0.0};   // 1 - tau / 0 = -inf -> clipped to 0
    double gr, gi;
    if (op == 1) {   // soft: X * clip(1 - tau / |X|, 0)
        gr = 1.0 - tau.x / m;
        gi = -tau.y / m;
    } else {         // garrote: X * clip(1 - tau^2 / |X|^2, 0)
        const double m2 = m * m;
        gr = 1.0 - (tau.x * tau.x - tau.y * tau.y) / m2;
        gi = -(2.0 * tau.x * tau.y) / m2;
    }
    const bool keep = gr > 0.0 || (gr == 0.0 && gi >= 0.0);   // lexicographic max(g, 0)
    return keep ? c64{X.x * gr - X.y * gi, X.x * gi + X.y * gr} : c64{0.0, 0.0};
}
__device__ __forceinline__ double shrink(double x, c64 tau, int op)   // real data, real tau
{
    const double m = fabs(x);
    if (op == 0) return m < tau.x ? 0.0 : x;
    if (m == 0.0) return 0.0;
    const double g = op == 1 ? 1.0 - tau.x / m : 1.0 - (tau.x * tau.x) / (m * m);
    return g > 0.0 ? x * g : 0.0;
}

template <typename T>
__device__ __forceinline__ T smooth_at(const T* line, int n, size_t st, int k)
{
    if (k >= 0 && k < n) return line[(size_t)k * st];
    if (n == 1) return line[0];
    if (k < 0) return extrapolate(line[0], line[st], (double)(-k));
    return extrapolate(line[(size_t)(n - 1) * st], line[(size_t)(n - 2) * st], (double)(k - n + 1));
}

// thresholds fused into the last analysis step of a level: z < 0 = leave that output alone
struct Thresh64 {
    const c64* tau;   // [slice][niter][nlev][3]
    int niter, iter, nlev, lvl, op, z_lo, z_hi;
    const int* done;  // per-slice state, != 0: the slice is finished / empty -- leave its arrays alone
};

// forward step along one axis (p3d_wavelet.hip dwt_axis_kernel): out[o] = sum_j f[j] xe[2 o + 1 - j]
template <typename T>
__global__ void dwt_axis64_kernel(const T* in, T* lo, T* hi, const Filters64* fp, int nlines, int n, int nout, size_t in_lin, size_t in_el, size_t in_slice,
                                  size_t out_lin, size_t out_el, size_t lo_slice, size_t hi_slice, Thresh64 th)
{
    __shared__ double f_lo[MAXL], f_hi[MAXL];
    const int s = blockIdx.y, L = fp->len;
    if (th.done && th.done[s] != 0) return;
    for (int j = threadIdx.x; j < L; j += blockDim.x) { f_lo[j] = fp->dec_lo[j]; f_hi[j] = fp->dec_hi[j]; }
    __syncthreads();
    const size_t total = (size_t)nlines * nout;
    c64 t_lo{0.0, 0.0}, t_hi{0.0, 0.0};
    if (th.z_lo >= 0) t_lo = th.tau[(((size_t)s * th.niter + th.iter) * th.nlev + th.lvl) * 3 + th.z_lo];
    if (th.z_hi >= 0) t_hi = th.tau[(((size_t)s * th.niter + th.iter) * th.nlev + th.lvl) * 3 + th.z_hi];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int line, o;   // neighbouring threads walk along the contiguous direction of the data
        if (in_el == 1) { line = (int)(i / nout); o = (int)(i - (size_t)line * nout); }
        else { o = (int)(i / nlines); line = (int)(i - (size_t)o * nlines); }
        const T* src = in + (size_t)s * in_slice + (size_t)line * in_lin;
        T a = zero_of<T>(), d = zero_of<T>();
        const int top = 2 * o + 1;
        if (top - (L - 1) >= 0 && top < n) {   // interior: no extension
            const T* q = src + (size_t)top * in_el;
            for (int j = 0; j < L; ++j) {
                const T v = q[-(ptrdiff_t)((size_t)j * in_el)];
                acc_tap(a, f_lo[j], v);
                acc_tap(d, f_hi[j], v);
            }
        } else {
            for (int j = 0; j < L; ++j) {
                const T v = smooth_at(src, n, in_el, top - j);
                acc_tap(a, f_lo[j], v);
                acc_tap(d, f_hi[j], v);
            }
        }
        if (th.z_lo >= 0) a = shrink(a, t_lo, th.op);
        if (th.z_hi >= 0) d = shrink(d, t_hi, th.op);
        const size_t dst = (size_t)line * out_lin + (size_t)o * out_el;
        lo[(size_t)s * lo_slice + dst] = a;
        hi[(size_t)s * hi_slice + dst] = d;
    }
}

// inverse step along one axis: out[m] = sum_k a[k] rec_lo[m + L - 2 - 2k] + d[k] rec_hi[m + L - 2 - 2k]
template <typename T>
__global__ void idwt_axis64_kernel(const T* a, const T* d, T* out, const Filters64* fp, int nlines, int n, int nout, size_t a_lin, size_t a_el, size_t a_slice,
                                   size_t d_lin, size_t d_el, size_t d_slice, size_t out_lin, size_t out_el, size_t out_slice, const int* done)
{
    __shared__ double r_lo[MAXL], r_hi[MAXL];
    const int s = blockIdx.y, L = fp->len;
    if (done && done[s] != 0) return;
    for (int j = threadIdx.x; j < L; j += blockDim.x) { r_lo[j] = fp->rec_lo[j]; r_hi[j] = fp->rec_hi[j]; }
    __syncthreads();
    const size_t total = (size_t)nlines * nout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int line, m;
        if (out_el == 1) { line = (int)(i / nout); m = (int)(i - (size_t)line * nout); }
        else { m = (int)(i / nlines); line = (int)(i - (size_t)m * nlines); }
        const T* pa = a + (size_t)s * a_slice + (size_t)line * a_lin;
        const T* pd = d + (size_t)s * d_slice + (size_t)line * d_lin;
        const int k0 = m / 2;
        int k1 = (m + L - 2) / 2;
        if (k1 > n - 1) k1 = n - 1;
        T acc = zero_of<T>();
        for (int k = k0; k <= k1; ++k) {
            const int j = m + L - 2 - 2 * k;
            acc_tap(acc, r_lo[j], pa[(size_t)k * a_el]);
            acc_tap(acc, r_hi[j], pd[(size_t)k * d_el]);
        }
        out[(size_t)s * out_slice + (size_t)line * out_lin + (size_t)m * out_el] = acc;
    }
}

// ---- cube element access: the cube (x, out) has the caller's dtype, the work arrays are T (double or c64) ------------------------------------
__device__ __forceinline__ c64 load_x(const void* x, int dtype, size_t g, c64*)
{
    switch (dtype) {
        case P3D_C128: return reinterpret_cast<const c64*>(x)[g];
        case P3D_F64: return c64{reinterpret_cast<const double*>(x)[g], 0.0};
        case P3D_C64: { const p3d::c32 v = reinterpret_cast<const p3d::c32*>(x)[g]; return c64{(double)v.x, (double)v.y}; }
        default: return c64{(double)reinterpret_cast<const float*>(x)[g], 0.0};
    }
}
__device__ __forceinline__ double load_x(const void* x, int dtype, size_t g, double*)   // real dtypes only
{
    return dtype == P3D_F64 ? reinterpret_cast<const double*>(x)[g] : (double)reinterpret_cast<const float*>(x)[g];
}
__device__ __forceinline__ void store_out(void* out, int dtype, size_t g, c64 v)   // a real cube gets np.real() of the iterate (POCS.py:656)
{
    switch (dtype) {
        case P3D_C128: reinterpret_cast<c64*>(out)[g] = v; break;
        case P3D_F64: reinterpret_cast<double*>(out)[g] = v.x; break;
        case P3D_C64: reinterpret_cast<p3d::c32*>(out)[g] = p3d::c32{(float)v.x, (float)v.y}; break;
        default: reinterpret_cast<float*>(out)[g] = (float)v.x; break;
    }
}
__device__ __forceinline__ void store_out(void* out, int dtype, size_t g, double v)
{
    if (dtype == P3D_F64) reinterpret_cast<double*>(out)[g] = v;
    else reinterpret_cast<float*>(out)[g] = (float)v;
}

// statistics of the detail arrays for the schedule (POCS.py:253-254, 281): per (slice, level, detail) Re, Im of the lexicographic max, max |d|, min |d|
constexpr int WSTAT_MAX_LEVELS = 16;
struct WStatLevels {
    size_t off[WSTAT_MAX_LEVELS], count[WSTAT_MAX_LEVELS];
};
template <typename T>
__global__ void wstats64_kernel(const T* coef, size_t coef_slice, const WStatLevels lv, double* stats, int nlev)
{
    __shared__ double sh[4 * 4];
    const int lvl = blockIdx.x, s = blockIdx.y, z = blockIdx.z;
    const size_t count = lv.count[lvl];
    const T* p = coef + (size_t)s * coef_slice + lv.off[lvl] + (size_t)z * count;
    double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY;
    for (size_t i = threadIdx.x; i < count; i += blockDim.x) {
        const T v = p[i];
        const double vr = re_of(v), vi = im_of(v), q = mag(v);
        if (vr > lr || (vr == lr && vi > li)) { lr = vr; li = vi; }
        mx = fmax(mx, q);
        mn = fmin(mn, q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
        if (orr > lr || (orr == lr && oi > li)) { lr = orr; li = oi; }
        mx = fmax(mx, __shfl_down(mx, o, 64));
        mn = fmin(mn, __shfl_down(mn, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        double* me = sh + (threadIdx.x >> 6) * 4;
        me[0] = lr; me[1] = li; me[2] = mx; me[3] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < (int)blockDim.x / 64; ++t) {
            const double* o = sh + t * 4;
            if (o[0] > lr || (o[0] == lr && o[1] > li)) { lr = o[0]; li = o[1]; }
            mx = fmax(mx, o[2]);
            mn = fmin(mn, o[3]);
        }
        double* q = stats + (((size_t)s * nlev + lvl) * 3 + z) * 4;
        q[0] = lr; q[1] = li; q[2] = mx; q[3] = mn;
    }
}

// mode 0: first input (feed = x or its APOCS mix; rowsum = sum |x| per row)
// mode 1: crop of the reconstruction + re-insertion (POCS.py:609, 616-619), rowsum = sum |x_new| per row, feed for the next iteration
// One workgroup per row; the row sums are added up in a fixed order by wrowsum64_kernel (the cost, POCS.py:622, is a difference of two
// nearly equal sums).
template <typename T>
__global__ __launch_bounds__(256) void wupdate64_kernel(const T* rec, size_t rec_ld, size_t rec_slice, T* feed, const void* x, int dtype, const double* mask, void* out,
                                                        double* rowsum, int mode, int adaptive, int write_out, double alpha, int n1, int n2, const int* done, int zero_fill)
{
    __shared__ double sh[256];
    const int s = blockIdx.y, r = blockIdx.x;
    const size_t per = (size_t)n1 * n2, g0 = (size_t)s * per + (size_t)r * n2;
    const int dn = done ? done[s] : 0;
    if (dn != 0) {
        if (zero_fill && dn < 0)   // an all-zero slice is handed back untouched (POCS.py:515-521)
            for (int c = threadIdx.x; c < n2; c += blockDim.x) store_out(out, dtype, g0 + c, zero_of<T>());
        if (threadIdx.x == 0) rowsum[(size_t)s * n1 + r] = 0.0;
        return;
    }
    const T* const rrow = rec ? rec + (size_t)s * rec_slice + (size_t)r * rec_ld : nullptr;
    const double* const mrow = mask ? mask + (size_t)r * n2 : nullptr;
    double acc = 0.0;
    for (int c = threadIdx.x; c < n2; c += blockDim.x) {
        const size_t g = g0 + c;
        const T xo = load_x(x, dtype, g, (T*)nullptr);
        const double m = mrow ? mrow[c] : 0.0;
        const double wgt = 1.0 - alpha * m;                    // POCS.py:616
        T xn;
        if (mode == 0) {
            xn = xo;
        } else {
            xn = scale(rrow[c], wgt) + scale(xo, alpha);       // POCS.py:617-619
            if (write_out) store_out(out, dtype, g, xn);
        }
        acc += mag(xn);
        if (adaptive) {   // x_input of the next iteration (POCS.py:572-575)
            const T blend = scale(xo, alpha) + scale(xn, wgt);
            feed[g] = blend + scale(xo - scale(xn, m), 1.0 - alpha);
        } else {
            feed[g] = xn;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) rowsum[(size_t)s * n1 + r] = sh[0];
}

__global__ __launch_bounds__(256) void wrowsum64_kernel(const double* rowsum, double* sums, int n1, const int* done)
{
    __shared__ double sh[256];
    const int s = blockIdx.x;
    double acc = 0.0;
    for (int r = threadIdx.x; r < n1; r += 256) acc += rowsum[(size_t)s * n1 + r];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && !(done && done[s] != 0)) sums[s] = sh[0];
}

__global__ void wconv64_kernel(const double* sums, int* done, int nslices, int iter, double eps)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices || done[s] != 0) return;
    const double cur = sums[(size_t)(iter + 1) * nslices + s], prev = sums[(size_t)iter * nslices + s];
    const double d = cur - prev;
    if (iter > 2 && (d * d) / (cur * cur) < eps) done[s] = iter + 1;   // POCS.py:622, 631
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

inline unsigned blocks_for(size_t n) { const size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

struct p3d_wplan64 {
    int device = 0, nil = 0, nxl = 0, max_slices = 0, nlev = 0, flen = 0;
    Filters64* f = nullptr;         // device
    std::vector<int> h, w;          // h[l], w[l], l = 0..nlev (level 0 = the slice itself)
    std::vector<int> rh, rw;        // shape of the reconstruction OF level l (l = 0..nlev-1): 2 h[l+1] - L + 2
    std::vector<size_t> doff;       // offset of level-l details in the coefficient vector of one slice (PyWavelets' order: cA, coarsest ... finest)
    size_t ncoef = 0;
    hipStream_t stream = nullptr;
    c64 *coef = nullptr, *feed = nullptr, *lo = nullptr, *hi = nullptr, *tau = nullptr;   // sized for complex128 elements; the real instantiation uses half
    std::vector<c64*> approx, rec;
    double *sums = nullptr, *rowsum = nullptr, *stats = nullptr, *mask = nullptr;
    size_t sums_cap = 0, tau_cap = 0;
    int* done = nullptr;
    void *st_x = nullptr, *st_out = nullptr;
    const void* cur_x = nullptr;
    void* cur_out = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t per() const { return (size_t)nil * nxl; }
};

extern "C" int p3d_wavelet64_plan_destroy(p3d_wplan64* p)
{
    if (!p) return P3D_OK;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    void* bufs[] = {p->f, p->coef, p->feed, p->lo, p->hi, p->tau, p->sums, p->rowsum, p->done, p->stats, p->mask, p->st_x, p->st_out};
    for (void* b : bufs) if (b) hipFree(b);
    for (c64* b : p->approx) if (b) hipFree(b);
    for (c64* b : p->rec) if (b) hipFree(b);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return P3D_OK;
}

extern "C" int p3d_wavelet64_plan_create(p3d_wplan64** out, int device, int nil, int nxl, int max_slices, const double* dec_lo, const double* dec_hi,
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
    int maxlev = 0;   // pywt.dwt_max_level(min(shape), flen)
    if (nmin >= flen - 1) maxlev = (int)std::floor(std::log2((double)nmin / (flen - 1.0)));
    if (maxlev < 0) maxlev = 0;
    if (level < 0) level = maxlev;
    if (level < 1) return wfail(P3D_ERR_UNSUPPORTED, "a %d x %d slice is too small for a %d-tap wavelet (0 levels)", nil, nxl, flen);
    if (level > WSTAT_MAX_LEVELS) return wfail(P3D_ERR_UNSUPPORTED, "more than %d levels", WSTAT_MAX_LEVELS);

    p3d_wplan64* p = new p3d_wplan64;
    p->device = device; p->nil = nil; p->nxl = nxl; p->max_slices = max_slices; p->nlev = level; p->flen = flen;
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
        p3d_wavelet64_plan_destroy(p);
        return wfail(P3D_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
#define ALLOC(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return bail(#ptr, e)
    if ((e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess) return bail("stream", e);
    if ((e = hipEventCreate(&p->ev0)) != hipSuccess) return bail("event", e);
    if ((e = hipEventCreate(&p->ev1)) != hipSuccess) return bail("event", e);
    Filters64 host{};
    host.len = flen;
    for (int j = 0; j < flen; ++j) { host.dec_lo[j] = dec_lo[j]; host.dec_hi[j] = dec_hi[j]; host.rec_lo[j] = rec_lo[j]; host.rec_hi[j] = rec_hi[j]; }
    ALLOC(p->f, sizeof(Filters64));
    if ((e = hipMemcpy(p->f, &host, sizeof(Filters64), hipMemcpyHostToDevice)) != hipSuccess) return bail("filters", e);
    const size_t S = (size_t)max_slices;
    ALLOC(p->coef, sizeof(c64) * p->ncoef * S);
    ALLOC(p->feed, sizeof(c64) * p->per() * S);
    const size_t tmp = (size_t)std::max(p->h[0], p->rh[0]) * p->w[1];
    ALLOC(p->lo, sizeof(c64) * tmp * S);
    ALLOC(p->hi, sizeof(c64) * tmp * S);
    p->approx.assign(level + 1, nullptr);
    for (int l = 1; l < level; ++l) ALLOC(p->approx[l], sizeof(c64) * (size_t)p->h[l] * p->w[l] * S);
    p->rec.assign(level, nullptr);
    for (int l = 0; l < level; ++l) ALLOC(p->rec[l], sizeof(c64) * (size_t)p->rh[l] * p->rw[l] * S);
    ALLOC(p->done, sizeof(int) * S);
    ALLOC(p->stats, sizeof(double) * 4 * 3 * (size_t)level * S);
    ALLOC(p->rowsum, sizeof(double) * (size_t)nil * S);
    ALLOC(p->mask, sizeof(double) * p->per());
    ALLOC(p->st_x, sizeof(c64) * p->per() * S);
    ALLOC(p->st_out, sizeof(c64) * p->per() * S);
#undef ALLOC
    *out = p;
    return P3D_OK;
}

extern "C" int p3d_wavelet64_info(p3d_wplan64* p, int* nlev, int64_t* ncoef)
{
    if (!p) return wfail(P3D_ERR_INVALID, "NULL plan");
    if (nlev) *nlev = p->nlev;
    if (ncoef) *ncoef = (int64_t)p->ncoef;
    return P3D_OK;
}

template <typename T> static T* as(c64* p) { return reinterpret_cast<T*>(p); }

// feed (nil x nxl per slice) -> coefficient vectors (cA, details coarse -> fine); with `th` the details are thresholded as they are produced
// (threshold_wavelet, POCS.py:105-166) -- the approximation is never touched (POCS.py:586-587)
template <typename T>
static int w_forward(p3d_wplan64* p, int ns, const Thresh64* th, const int* done)
{
    const dim3 blk(256);
    const Thresh64 none{nullptr, 0, 0, 0, 0, 0, -1, -1, done};
    T *lo = as<T>(p->lo), *hi = as<T>(p->hi), *coef = as<T>(p->coef);
    for (int l = 1; l <= p->nlev; ++l) {
        const T* src = l == 1 ? as<T>(p->feed) : as<T>(p->approx[l - 1]);
        const int H = p->h[l - 1], W = p->w[l - 1], Ho = p->h[l], Wo = p->w[l];
        const size_t cnt = (size_t)Ho * Wo;
        // along axis 1 (rows are contiguous): (H x W) -> lo, hi (H x Wo)
        dwt_axis64_kernel<T><<<dim3(blocks_for((size_t)H * Wo), ns), blk, 0, p->stream>>>(src, lo, hi, p->f, H, W, Wo, (size_t)W, 1, (size_t)H * W, (size_t)Wo, 1,
                                                                                      (size_t)H * Wo, (size_t)H * Wo, none);
        // along axis 0 (lines = columns): lo -> (aa, da = cH), hi -> (ad = cV, dd = cD), each (Ho x Wo)
        T* cA = l == p->nlev ? coef : as<T>(p->approx[l]);
        const size_t cA_slice = l == p->nlev ? p->ncoef : cnt;
        T* det = coef + p->doff[l];
        Thresh64 t1 = none, t2 = none;
        if (th) {
            t1 = t2 = *th;
            t1.lvl = t2.lvl = p->nlev - l;   // PyWavelets' order: coarsest level first
            t1.z_lo = -1; t1.z_hi = 0;
            t2.z_lo = 1; t2.z_hi = 2;
        }
        dwt_axis64_kernel<T><<<dim3(blocks_for((size_t)Wo * Ho), ns), blk, 0, p->stream>>>(lo, cA, det, p->f, Wo, H, Ho, 1, (size_t)Wo, (size_t)H * Wo, 1, (size_t)Wo,
                                                                                       cA_slice, p->ncoef, t1);
        dwt_axis64_kernel<T><<<dim3(blocks_for((size_t)Wo * Ho), ns), blk, 0, p->stream>>>(hi, det + cnt, det + 2 * cnt, p->f, Wo, H, Ho, 1, (size_t)Wo, (size_t)H * Wo,
                                                                                       1, (size_t)Wo, p->ncoef, p->ncoef, t2);
    }
    W_TRY(hipGetLastError());
    return P3D_OK;
}

// coefficient vectors -> rec[0] (rh[0] x rw[0] per slice; its top-left nil x nxl block is the slice, POCS.py:513, 609)
template <typename T>
static int w_inverse(p3d_wplan64* p, int ns, const int* done)
{
    const dim3 blk(256);
    T *lo = as<T>(p->lo), *hi = as<T>(p->hi), *coef = as<T>(p->coef);
    for (int l = p->nlev; l >= 1; --l) {
        const int Ho = p->h[l], Wo = p->w[l];
        const int RH = p->rh[l - 1], RW = p->rw[l - 1];
        const size_t cnt = (size_t)Ho * Wo;
        // approximation of level l: cA itself at the coarsest level, otherwise the reconstruction of level l (one row / column larger than
        // Ho x Wo at times: the extra samples are ignored, as in pywt.waverec2)
        const T* a = l == p->nlev ? coef : as<T>(p->rec[l]);
        const size_t a_ld = l == p->nlev ? (size_t)Wo : (size_t)p->rw[l];
        const size_t a_slice = l == p->nlev ? p->ncoef : (size_t)p->rh[l] * p->rw[l];
        const T* det = coef + p->doff[l];
        idwt_axis64_kernel<T><<<dim3(blocks_for((size_t)Wo * RH), ns), blk, 0, p->stream>>>(a, det, lo, p->f, Wo, Ho, RH, 1, a_ld, a_slice, 1, (size_t)Wo, p->ncoef, 1,
                                                                                        (size_t)Wo, (size_t)RH * Wo, done);
        idwt_axis64_kernel<T><<<dim3(blocks_for((size_t)Wo * RH), ns), blk, 0, p->stream>>>(det + cnt, det + 2 * cnt, hi, p->f, Wo, Ho, RH, 1, (size_t)Wo, p->ncoef, 1,
                                                                                        (size_t)Wo, p->ncoef, 1, (size_t)Wo, (size_t)RH * Wo, done);
        idwt_axis64_kernel<T><<<dim3(blocks_for((size_t)RH * RW), ns), blk, 0, p->stream>>>(lo, hi, as<T>(p->rec[l - 1]), p->f, RH, Wo, RW, (size_t)Wo, 1, (size_t)RH * Wo,
                                                                                        (size_t)Wo, 1, (size_t)RH * Wo, (size_t)RW, 1, (size_t)RH * RW, done);
    }
    W_TRY(hipGetLastError());
    return P3D_OK;
}

static bool on_plan_device(const p3d_wplan64* p, const void* ptr)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();   // ordinary host memory
        return false;
    }
    return at.type == hipMemoryTypeDevice && at.device == p->device;
}

static size_t elem_bytes(int dtype) { return dtype == P3D_C128 ? 16 : (dtype == P3D_F64 || dtype == P3D_C64) ? 8 : 4; }
static bool real_dtype(int dtype) { return dtype == P3D_F64 || dtype == P3D_F32; }

static int w_check(p3d_wplan64* p, int nslices, int dtype)
{
    if (!p) return wfail(P3D_ERR_INVALID, "NULL plan");
    if (nslices < 1 || nslices > p->max_slices) return wfail(P3D_ERR_INVALID, "nslices = %d outside 1..max_slices (%d)", nslices, p->max_slices);
    if (dtype != P3D_C64 && dtype != P3D_F32 && dtype != P3D_C128 && dtype != P3D_F64) return wfail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    W_TRY(hipSetDevice(p->device));
    return P3D_OK;
}

static int ensure_sums(p3d_wplan64* p, size_t n)
{
    if (p->sums_cap < n) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr; p->sums_cap = 0;
        W_TRY(hipMalloc((void**)&p->sums, sizeof(double) * n));
        p->sums_cap = n;
    }
    return P3D_OK;
}

template <typename T>
static int w_stats(p3d_wplan64* p, int dtype, int nslices)
{
    wupdate64_kernel<T><<<dim3(p->nil, nslices), 256, 0, p->stream>>>(nullptr, 0, 0, as<T>(p->feed), p->cur_x, dtype, nullptr, nullptr, p->rowsum, 0, 0, 0, 1.0, p->nil,
                                                                    p->nxl, nullptr, 0);
    int rc = w_forward<T>(p, nslices, nullptr, nullptr);
    if (rc) return rc;
    WStatLevels lv{};
    for (int l = p->nlev, i = 0; l >= 1; --l, ++i) { lv.off[i] = p->doff[l]; lv.count[i] = (size_t)p->h[l] * p->w[l]; }
    wstats64_kernel<T><<<dim3(p->nlev, nslices, 3), 256, 0, p->stream>>>(as<T>(p->coef), p->ncoef, lv, p->stats, p->nlev);
    W_TRY(hipGetLastError());
    return P3D_OK;
}

template <typename T>
static int w_loop(p3d_wplan64* p, int dtype, int nslices, const p3d_pocs_params* prm)
{
    const int niter = prm->niter;
    const bool early = prm->eps > 0.0, adaptive = prm->version == P3D_VER_ADAPTIVE;
    const dim3 ugrid(p->nil, nslices);
    wupdate64_kernel<T><<<ugrid, 256, 0, p->stream>>>(nullptr, 0, 0, as<T>(p->feed), p->cur_x, dtype, p->mask, p->cur_out, p->rowsum, 0, adaptive ? 1 : 0, 0, prm->alpha,
                                                     p->nil, p->nxl, p->done, 0);
    wrowsum64_kernel<<<nslices, 256, 0, p->stream>>>(p->rowsum, p->sums, p->nil, p->done);
    for (int k = 0; k < niter; ++k) {
        const bool last = k + 1 == niter;
        const Thresh64 th{p->tau, niter, k, p->nlev, 0, prm->thresh_op, -1, -1, p->done};
        int rc = w_forward<T>(p, nslices, &th, p->done);
        if (rc) return rc;
        if ((rc = w_inverse<T>(p, nslices, p->done))) return rc;
        // (early exit: every iterate is stored, so that a slice that converges keeps its last one -- wconv64_kernel switches it off afterwards)
        wupdate64_kernel<T><<<ugrid, 256, 0, p->stream>>>(as<T>(p->rec[0]), (size_t)p->rw[0], (size_t)p->rh[0] * p->rw[0], as<T>(p->feed), p->cur_x, dtype, p->mask,
                                                         p->cur_out, p->rowsum, 1, (adaptive && !last) ? 1 : 0, (early || last) ? 1 : 0, prm->alpha, p->nil, p->nxl, p->done,
                                                         last ? 1 : 0);
        wrowsum64_kernel<<<nslices, 256, 0, p->stream>>>(p->rowsum, p->sums + (size_t)(k + 1) * nslices, p->nil, p->done);
        if (early) wconv64_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
    }
    W_TRY(hipGetLastError());
    return P3D_OK;
}

extern "C" {

// statistics of the detail arrays of transform(x) for the schedule: stats [nslices][nlev][3][4] doubles = Re, Im of the lexicographic max, max |d|,
// min |d|; levels coarse -> fine (PyWavelets' order).  x: host or device pointer, dtype P3D_C128 / P3D_F64 / P3D_C64 / P3D_F32.
int p3d_wavelet64_stats(p3d_wplan64* p, const void* x, int dtype, int nslices, double* stats)
{
    int rc = w_check(p, nslices, dtype);
    if (rc) return rc;
    if (!x || !stats) return wfail(P3D_ERR_INVALID, "NULL buffer");
    if (on_plan_device(p, x)) {
        p->cur_x = x;
    } else {
        W_TRY(hipMemcpyAsync(p->st_x, x, elem_bytes(dtype) * p->per() * nslices, hipMemcpyDefault, p->stream));
        p->cur_x = p->st_x;
    }
    if ((rc = real_dtype(dtype) ? w_stats<double>(p, dtype, nslices) : w_stats<c64>(p, dtype, nslices))) return rc;
    W_TRY(hipMemcpyAsync(stats, p->stats, sizeof(double) * (size_t)nslices * p->nlev * 12, hipMemcpyDeviceToHost, p->stream));
    W_TRY(hipStreamSynchronize(p->stream));
    return P3D_OK;
}

// the loop (POCS.py:549-632 with the WAVELET branches) in double precision; tau: HOST [nslices][niter][nlev][3][2] doubles, levels coarse -> fine;
// mask: DOUBLE [nil][nxl], host or device; x / out: host or device, dtype as above (complex64 / float32 cubes are converted on load / store)
int p3d_wavelet64_run(p3d_wplan64* p, const void* x, int dtype, const double* mask, const double* tau, const uint8_t* active, const p3d_pocs_params* prm,
                      void* out, int nslices, int32_t* niter_done, double* sums, double* elapsed_ms)
{
    int rc = w_check(p, nslices, dtype);
    if (rc) return rc;
    if (!x || !mask || !tau || !prm || !out) return wfail(P3D_ERR_INVALID, "NULL argument");
    if (prm->niter < 1) return wfail(P3D_ERR_INVALID, "niter must be >= 1");
    if (prm->thresh_op < P3D_OP_HARD || prm->thresh_op > P3D_OP_GARROTE)
        return wfail(P3D_ERR_UNSUPPORTED, "thresh_op %d is not implemented for the wavelet transform", prm->thresh_op);
    const int niter = prm->niter;
    const size_t ntau = (size_t)nslices * niter * p->nlev * 3, nsum = (size_t)(niter + 1) * nslices;
    if (p->tau_cap < ntau) {
        if (p->tau) hipFree(p->tau);
        p->tau = nullptr; p->tau_cap = 0;
        W_TRY(hipMalloc((void**)&p->tau, sizeof(c64) * ntau));
        p->tau_cap = ntau;
    }
    if ((rc = ensure_sums(p, nsum))) return rc;
    bool real_tau = true;
    for (size_t i = 0; i < ntau; ++i) real_tau = real_tau && tau[2 * i + 1] == 0.0;
    // real cubes with real thresholds stay real through the whole loop (what PyWavelets does for real input)
    const bool real_path = real_dtype(dtype) && real_tau;
    std::vector<int> done_h(nslices, 0);
    if (active) for (int s = 0; s < nslices; ++s) done_h[s] = active[s] ? 0 : -1;
    const size_t cube_bytes = elem_bytes(dtype) * p->per() * nslices;
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
    // every copy goes onto the plan's (non-blocking) stream: a device-to-device hipMemcpy on the null stream need not have finished when it returns
    W_TRY(hipMemcpyAsync(p->mask, mask, sizeof(double) * p->per(), hipMemcpyDefault, p->stream));
    W_TRY(hipMemcpyAsync(p->tau, tau, sizeof(c64) * ntau, hipMemcpyHostToDevice, p->stream));
    W_TRY(hipMemcpyAsync(p->done, done_h.data(), sizeof(int) * nslices, hipMemcpyHostToDevice, p->stream));
    W_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nsum, p->stream));
    W_TRY(hipEventRecord(p->ev0, p->stream));
    if ((rc = real_path ? w_loop<double>(p, dtype, nslices, prm) : w_loop<c64>(p, dtype, nslices, prm))) return rc;
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
