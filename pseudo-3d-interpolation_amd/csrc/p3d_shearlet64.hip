// p3d_shearlet64.hip -- the SHEARLET variant of the POCS path in the REFERENCE's double precision.
//
// The reference's cubes are float32 but NumPy's FFTs (FFST's shearletTransformSpect / inverseShearletTransformSpect are np.fft.fft2 / ifft2
// around a product with the spectra, see p3d_shearlet.hip and oracle/shearlet_oracle.py) compute in double and hand back complex128 / float64
// coefficients: POCS_algorithm's whole SHEARLET loop (functions/POCS.py:526-527, 549-632) therefore runs in double, and only the final store
// (cube_POCS_interpolation_3D.py:324) narrows.  p3d_shearlet.hip runs that loop in float32 (the fast path); this file runs it in double, for
// complex128 / float64 cubes and for complex64 / float32 cubes on request (precision='reference'): results within 1e-10 of the NumPy restatement
// instead of 1e-4.
//
// One iteration of a batch of nb slices:  fft2(feed) -> x Psi_s into the coefficient buffer (nb * nsh slices) -> batched ifft2 -> threshold
// (real cubes: of the real part, FFST returns ST.real) -> batched fft2 -> sum over s of x Psi_s -> ifft2 -> re-insertion, cost.
//   fused form (both extents have a plan on the double-precision register engine, p3d_mix64.hip): three passes over the coefficients --
//     x Psi_s folded into the inverse row pass, the threshold into the column pass between its two transforms, x Psi_s and the sum over s into
//     the forward row pass (80 B per coefficient instead of ~210), row groups on which a shearlet's spectrum vanishes skipped by all three
//     (p3d_splan64::sup; P3D_SHEARLET64_NO_SUPPORT=1 moves every row); REAL cubes on symmetric spectra have real coefficients, so the coefficient
//     slices are Hermitian along their columns: rows 0 ... nil/2 only, two columns per transform (p3d_splan64::pair; P3D_SHEARLET64_NO_PAIR=1) --
//     and the slice-sized passes of p3d_f64.hip's own loop around them;
//   unfused form (any other shape; P3D_SHEARLET64_UNFUSED=1): separate kernels around p3d_f64.hip's line transforms (plan64_fft2), the
//     coefficient buffer being that plan's work buffer.
// Costs are sums of partial sums added in a fixed order (reproducible).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "p3d.h"
#include "p3d_internal.hpp"

namespace {

struct __attribute__((aligned(16))) c64 {
    double x, y;
};

int s64fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    p3d::set_last_error(buf);
    return code;
}
#define S_TRY(expr)                                                                                       \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return s64fail(P3D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define S_RC(expr)           \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)

inline unsigned blocks_for(size_t n, unsigned cap = 4096) { const size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); }

__device__ __forceinline__ c64 load_x(const void* x, int dtype, size_t g)
{
    switch (dtype) {
        case P3D_C128: return reinterpret_cast<const c64*>(x)[g];
        case P3D_F64: return c64{reinterpret_cast<const double*>(x)[g], 0.0};
        case P3D_C64: { const float2 v = reinterpret_cast<const float2*>(x)[g]; return c64{(double)v.x, (double)v.y}; }
        default: return c64{(double)reinterpret_cast<const float*>(x)[g], 0.0};
    }
}
__device__ __forceinline__ void store_out(void* out, int dtype, size_t g, c64 v)   // a real cube gets np.real() of the iterate (POCS.py:656)
{
    switch (dtype) {
        case P3D_C128: reinterpret_cast<c64*>(out)[g] = v; break;
        case P3D_F64: reinterpret_cast<double*>(out)[g] = v.x; break;
        case P3D_C64: reinterpret_cast<float2*>(out)[g] = float2{(float)v.x, (float)v.y}; break;
        default: reinterpret_cast<float*>(out)[g] = (float)v.x; break;
    }
}

// threshold_operator.py:9-112 on one coefficient, NumPy's semantics for a complex tau (lexicographic comparisons) -- as p3d_f64.hip's shrink64
__device__ __forceinline__ c64 shrink64(c64 X, c64 tau, int op)
{
    const double m = hypot(X.x, X.y);
    if (op == P3D_OP_HARD) {
        const bool below = m < tau.x || (m == tau.x && 0.0 < tau.y);
        return below ? c64{0.0, 0.0} : X;
    }
    if (m == 0.0) return c64{0.0, 0.0};
    double gr, gi;
    if (op == P3D_OP_SOFT) {
        gr = 1.0 - tau.x / m;
        gi = -tau.y / m;
    } else {
        const double m2 = m * m;
        gr = 1.0 - (tau.x * tau.x - tau.y * tau.y) / m2;
        gi = -(2.0 * tau.x * tau.y) / m2;
    }
    const bool keep = gr > 0.0 || (gr == 0.0 && gi >= 0.0);
    return keep ? c64{X.x * gr - X.y * gi, X.x * gi + X.y * gr} : c64{0.0, 0.0};
}

// U[b][s][i] = Psi[s][i] * F[b][i]          (grid.y = slice of the batch)
__global__ void spread64_kernel(const c64* F, const double* psi, c64* U, size_t per, int nsh, const int* done)
{
    const int b = blockIdx.y;
    if (done && done[b] != 0) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        const c64 f = F[(size_t)b * per + i];
        for (int s = 0; s < nsh; ++s) {
            const double w = psi[(size_t)s * per + i];
            U[((size_t)b * nsh + s) * per + i] = c64{f.x * w, f.y * w};
        }
    }
}

// A[b][i] = sum_s U[b][s][i] * Psi[s][i]      (np.sum over the last axis adds the shearlets in order for nsh < 8 ... pairwise beyond: the
// difference is rounding in the last place of a double)
__global__ void gather64_kernel(const c64* U, const double* psi, c64* A, size_t per, int nsh, const int* done)
{
    const int b = blockIdx.y;
    if (done && done[b] != 0) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        double ar = 0.0, ai = 0.0;
        for (int s = 0; s < nsh; ++s) {
            const double w = psi[(size_t)s * per + i];
            const c64 u = U[((size_t)b * nsh + s) * per + i];
            ar += u.x * w;
            ai += u.y * w;
        }
        A[(size_t)b * per + i] = c64{ar, ai};
    }
}

// coefficients of shearlet s of slice b: threshold with tau[b][iter][s] (grid.y = b*nsh + s); real_only drops the imaginary round-off first
// (FFST returns ST.real for real data)
__global__ void sthreshold64_kernel(c64* U, size_t per, int nsh, const c64* tau, int niter, int iter, int op, int real_only, const int* done)
{
    const int bs = blockIdx.y, b = bs / nsh, s = bs - b * nsh;
    if (done && done[b] != 0) return;
    const c64 t = tau[((size_t)b * niter + iter) * nsh + s];
    c64* p = U + (size_t)bs * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        c64 v = p[i];
        if (real_only) v.y = 0.0;
        p[i] = shrink64(v, t, op);
    }
}

// per (slice, shearlet): lexicographic (real data: signed) maximum, max |c|, min |c|, sum |c|^2 -> stats[(b*nsh + s)*5 ..]
__global__ __launch_bounds__(256) void sstats64_kernel(const c64* U, size_t per, int real_only, double* stats)
{
    __shared__ double sh[4 * 5];
    const int bs = blockIdx.x;
    const c64* p = U + (size_t)bs * per;
    double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY, sq = 0.0;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) {
        c64 v = p[i];
        if (real_only) v.y = 0.0;
        const double m = hypot(v.x, v.y);
        if (v.x > lr || (v.x == lr && v.y > li)) { lr = v.x; li = v.y; }
        mx = fmax(mx, m);
        mn = fmin(mn, m);
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
    if ((threadIdx.x & 63) == 0) {
        double* me = sh + (threadIdx.x >> 6) * 5;
        me[0] = lr; me[1] = li; me[2] = mx; me[3] = mn; me[4] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < 4; ++t) {
            const double* o = sh + t * 5;
            if (o[0] > lr || (o[0] == lr && o[1] > li)) { lr = o[0]; li = o[1]; }
            mx = fmax(mx, o[2]);
            mn = fmin(mn, o[3]);
            sq += o[4];
        }
        double* q = stats + (size_t)bs * 5;
        q[0] = lr; q[1] = li; q[2] = mx; q[3] = mn; q[4] = sq;
    }
}

// One workgroup per row, IN PLACE on F (A = ifft2(...) on entry, the next iteration's input on exit).
// mode 0: first input (F = x or its APOCS mix; rowsum = sum |x| per row)
// mode 1: re-insertion (POCS.py:616-619), rowsum = sum |x_new| per row, F = the next input (POCS.py:572-575)
__global__ __launch_bounds__(256) void supdate64_kernel(c64* F, const void* x, int dtype, const double* mask, void* out, double* rowsum, int mode, int adaptive,
                                                        int write_out, double alpha, int n1, int n2, const int* done, int zero_fill, int real_only)
{
    __shared__ double sh[256];
    const int b = blockIdx.y, r = blockIdx.x;
    const size_t g0 = ((size_t)b * n1 + r) * n2;
    const int dn = done ? done[b] : 0;
    if (dn != 0) {
        if (zero_fill && dn < 0)   // an all-zero slice is handed back untouched (POCS.py:515-521)
            for (int c = threadIdx.x; c < n2; c += blockDim.x) store_out(out, dtype, g0 + c, c64{0.0, 0.0});
        if (threadIdx.x == 0) rowsum[(size_t)b * n1 + r] = 0.0;
        return;
    }
    const double* const mrow = mask ? mask + (size_t)r * n2 : nullptr;
    double acc = 0.0;
    for (int c = threadIdx.x; c < n2; c += blockDim.x) {
        const size_t g = g0 + c;
        const c64 xo = load_x(x, dtype, g);
        const double m = mrow ? mrow[c] : 0.0;
        const double wgt = 1.0 - alpha * m;   // POCS.py:616
        c64 xn;
        if (mode == 0) {
            xn = xo;
        } else {
            c64 a = F[g];
            if (real_only) a.y = 0.0;   // inverseShearletTransformSpect of real coefficients returns the real part
            xn = c64{a.x * wgt + xo.x * alpha, a.y * wgt + xo.y * alpha};   // POCS.py:617-619
            if (write_out) store_out(out, dtype, g, xn);
        }
        acc += hypot(xn.x, xn.y);
        if (adaptive) {
            const double c1 = 1.0 - alpha;
            F[g] = c64{(xo.x * alpha + xn.x * wgt) + (xo.x - xn.x * m) * c1, (xo.y * alpha + xn.y * wgt) + (xo.y - xn.y * m) * c1};
        } else {
            F[g] = xn;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) rowsum[(size_t)b * n1 + r] = sh[0];
}

__global__ __launch_bounds__(256) void srowsum64_kernel(const double* rowsum, double* sums, int n1, const int* done)
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

__global__ void sconv64_kernel(const double* sums, int* done, int nslices, int iter, double eps)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices || done[s] != 0) return;
    const double cur = sums[(size_t)(iter + 1) * nslices + s], prev = sums[(size_t)iter * nslices + s];
    const double d = cur - prev;
    if (iter > 2 && (d * d) / (cur * cur) < eps) done[s] = iter + 1;   // POCS.py:622, 631
}

// *asym is raised when some Psi_s(-k) differs from Psi_s(k) by more than 1e-11 (the spectra are at most 1; a frame generator evaluates mirrored samples along
// different floating-point routes, the last bits may differ): real slices then have complex coefficients and the Hermitian form of the passes stays off
__global__ void psisym64_kernel(const double* psi, int* asym, int nil, int nxl, int nsh)
{
    const size_t per = (size_t)nil * nxl, total = per * nsh;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t s = i / per, r = i - s * per;
        const int k1 = (int)(r / nxl), k2 = (int)(r - (size_t)k1 * nxl);
        const int m1 = k1 ? nil - k1 : 0, m2 = k2 ? nxl - k2 : 0;
        bad = bad || (fabs(psi[i] - psi[s * per + (size_t)m1 * nxl + m2]) > 1e-11);
    }
    if (bad) atomicOr(asym, 1);
}

// F[b][k1][k2] = conj F[b][n1 - k1][(n2 - k2) mod n2] for k1 = n1/2 + 1 ... n1 - 1: the spectrum of a real slice, completed from the rows the gather pass computed
__global__ void mirror_rows64_kernel(c64* F, int n1, int n2, const int* done)
{
    if (done && done[blockIdx.y] != 0) return;
    c64* f = F + (size_t)blockIdx.y * n1 * n2;
    const size_t total = (size_t)(n1 - n1 / 2 - 1) * n2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k1 = n1 / 2 + 1 + (int)(i / n2), k2 = (int)(i % n2);
        const c64 m = f[(size_t)(n1 - k1) * n2 + (k2 ? n2 - k2 : 0)];
        f[(size_t)k1 * n2 + k2] = c64{m.x, -m.y};
    }
}

// sup[s][g] = 1 where rows g * G ... of Psi_s hold a non-zero sample (one wavefront per (g, s); sup zeroed before)
__global__ void rowsup64_kernel(const double* psi, unsigned char* sup, int nil, int nxl, int G, int ng)
{
    const int s = blockIdx.y, g = blockIdx.x;
    const int r0 = g * G, r1 = min(r0 + G, nil);
    const double* p = psi + ((size_t)s * nil + r0) * nxl;
    const size_t n = (size_t)(r1 - r0) * nxl;
    bool any = false;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) any = any || (p[i] != 0.0);
    if (__ballot(any) != 0ull && threadIdx.x == 0) sup[(size_t)s * ng + g] = 1;
}

}  // namespace

struct p3d_splan64 {
    int device = 0, nil = 0, nxl = 0, nsh = 0, max_slices = 0;
    p3d_plan64* fft = nullptr;   // unfused form: transforms + the coefficient buffer: max_slices * nsh complex128 slices
    p3d_plan64* pf = nullptr;    // fused form: the slice-sized passes (work buffer = the slices' spectra, staging buffers, mask); U is then the plan's own
    hipStream_t stream = nullptr;
    double* psi = nullptr;       // [nsh][nil][nxl]
    unsigned char* sup = nullptr;   // fused form: [nsh][sup_groups] row groups (of sup_rows rows: one workgroup of the row passes) on which Psi_s does not vanish
    int sup_groups = 0, sup_rows = 0;
    double sup_fraction = 1.0;
    bool pair = false;           // fused form, REAL cubes: Hermitian coefficient slices (rows 0 ... nil/2 only), two columns per transform (symmetric spectra, nil and nxl even;
                                 // P3D_SHEARLET64_NO_PAIR=1 switches it off)
    c64 *U = nullptr, *F = nullptr, *tau = nullptr;
    size_t tau_cap = 0, sums_cap = 0;
    double *sums = nullptr, *rowsum = nullptr, *stats = nullptr, *mask = nullptr;
    int* done = nullptr;
    void *st_x = nullptr, *st_out = nullptr;
    const void* cur_x = nullptr;
    void* cur_out = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t per() const { return (size_t)nil * nxl; }
};

namespace {

bool on_plan_device(const p3d_splan64* p, const void* ptr)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();   // ordinary host memory
        return false;
    }
    return at.type == hipMemoryTypeDevice && at.device == p->device;
}

size_t elem_bytes(int dtype) { return dtype == P3D_C128 ? 16 : (dtype == P3D_F64 || dtype == P3D_C64) ? 8 : 4; }
bool real_dtype(int dtype) { return dtype == P3D_F64 || dtype == P3D_F32; }

int s_check(p3d_splan64* p, int nslices, int dtype)
{
    if (!p) return s64fail(P3D_ERR_INVALID, "NULL plan");
    if (nslices < 1 || nslices > p->max_slices) return s64fail(P3D_ERR_INVALID, "nslices = %d outside 1..max_slices (%d)", nslices, p->max_slices);
    if (dtype != P3D_C64 && dtype != P3D_F32 && dtype != P3D_C128 && dtype != P3D_F64) return s64fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    S_TRY(hipSetDevice(p->device));
    return P3D_OK;
}

int ensure_sums(p3d_splan64* p, size_t n)
{
    if (p->sums_cap < n) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr; p->sums_cap = 0;
        S_TRY(hipMalloc((void**)&p->sums, sizeof(double) * n));
        p->sums_cap = n;
    }
    return P3D_OK;
}

// x (host or device) -> cur_x: the caller's own device buffer where it passed one, the staging buffer otherwise
int take_x(p3d_splan64* p, const void* x, int dtype, int nslices)
{
    if (on_plan_device(p, x)) {
        p->cur_x = x;
    } else {
        S_TRY(hipMemcpyAsync(p->st_x, x, elem_bytes(dtype) * p->per() * nslices, hipMemcpyDefault, p->stream));
        p->cur_x = p->st_x;
    }
    return P3D_OK;
}

// F = the input -> U = the coefficients of all shearlets (spatial domain)
int s_forward(p3d_splan64* p, int ns, const int* done)
{
    const size_t per = p->per();
    S_RC(p3d::plan64_fft2(p->fft, p->F, ns, false, done, 1));
    spread64_kernel<<<dim3(blocks_for(per, 1024), ns), 256, 0, p->stream>>>(p->F, p->psi, p->U, per, p->nsh, done);
    S_RC(p3d::plan64_fft2(p->fft, p->U, ns * p->nsh, true, done, p->nsh));
    S_TRY(hipGetLastError());
    return P3D_OK;
}

// U (coefficients) -> F = the slice
int s_inverse(p3d_splan64* p, int ns, const int* done)
{
    const size_t per = p->per();
    S_RC(p3d::plan64_fft2(p->fft, p->U, ns * p->nsh, false, done, p->nsh));
    gather64_kernel<<<dim3(blocks_for(per, 1024), ns), 256, 0, p->stream>>>(p->U, p->psi, p->F, per, p->nsh, done);
    S_RC(p3d::plan64_fft2(p->fft, p->F, ns, true, done, 1));
    S_TRY(hipGetLastError());
    return P3D_OK;
}

}  // namespace

extern "C" {

int p3d_shearlet64_plan_destroy(p3d_splan64* p)
{
    if (!p) return P3D_OK;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    void* bufs[] = {p->psi, p->sup, p->F, p->tau, p->sums, p->rowsum, p->stats, p->done};
    for (void* b : bufs) if (b) hipFree(b);
    if (p->pf) {
        if (p->U) hipFree(p->U);
    } else {
        if (p->mask) hipFree(p->mask);
        if (p->st_x) hipFree(p->st_x);
        if (p->st_out) hipFree(p->st_out);
    }
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->fft) p3d_plan64_destroy(p->fft);   // (owns the stream and U of the unfused form)
    if (p->pf) p3d_plan64_destroy(p->pf);     // (owns the stream, the mask and the staging buffers of the fused form)
    delete p;
    return P3D_OK;
}

// psi: HOST double [nsh][nil][nxl] (real spectra, FFT order)
int p3d_shearlet64_plan_create(p3d_splan64** out, int device, int nil, int nxl, int nsh, const double* psi, int max_slices)
{
    if (!out || !psi) return s64fail(P3D_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (nil < 1 || nxl < 1 || nsh < 1 || max_slices < 1) return s64fail(P3D_ERR_INVALID, "bad shape / batch size");
    if ((long long)max_slices * nsh > 65535) return s64fail(P3D_ERR_INVALID, "max_slices * nsh = %lld exceeds 65535", (long long)max_slices * nsh);
    p3d_splan64* p = new p3d_splan64;
    p->device = device; p->nil = nil; p->nxl = nxl; p->nsh = nsh; p->max_slices = max_slices;
    int rc = p3d_plan64_create(&p->pf, device, nil, nxl, max_slices);
    if (rc) { delete p; return rc; }   // message already set
    const char* env = getenv("P3D_SHEARLET64_UNFUSED");
    if (!p3d::plan64_shear_supported(p->pf) || (env && env[0] == '1')) {
        p3d_plan64_destroy(p->pf);
        p->pf = nullptr;
        rc = p3d::plan64_create_bare(&p->fft, device, nil, nxl, max_slices * nsh);
        if (rc) { delete p; return rc; }
        p->stream = p3d::plan64_stream(p->fft);
        p->U = reinterpret_cast<c64*>(p3d::plan64_work(p->fft));
    } else {
        p->stream = p3d::plan64_stream(p->pf);
    }
    auto bail = [&](const char* what, hipError_t e) {
        p3d_shearlet64_plan_destroy(p);
        return s64fail(P3D_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
#define ALLOC(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return bail(#ptr, e)
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipEventCreate(&p->ev0)) != hipSuccess) return bail("event", e);
    if ((e = hipEventCreate(&p->ev1)) != hipSuccess) return bail("event", e);
    const size_t S = (size_t)max_slices, per = p->per();
    ALLOC(p->psi, sizeof(double) * per * nsh);
    ALLOC(p->done, sizeof(int) * S);
    ALLOC(p->stats, sizeof(double) * 5 * nsh * S);
    if (p->pf) {
        ALLOC(p->U, sizeof(c64) * per * nsh * S);
        p->mask = p3d::plan64_mask(p->pf);
        p->st_x = p3d::plan64_stage_x(p->pf);
        p->st_out = p3d::plan64_stage_out(p->pf);
    } else {
        ALLOC(p->F, sizeof(c64) * per * S);
        ALLOC(p->rowsum, sizeof(double) * nil * S);
        ALLOC(p->mask, sizeof(double) * per);
        ALLOC(p->st_x, sizeof(c64) * per * S);
        ALLOC(p->st_out, sizeof(c64) * per * S);
    }
#undef ALLOC
    if ((e = hipMemcpy(p->psi, psi, sizeof(double) * per * nsh, hipMemcpyHostToDevice)) != hipSuccess) return bail("upload of Psi", e);
    if (p->pf && !getenv("P3D_SHEARLET64_NO_SUPPORT")) {
        // which row groups of every spectrum hold anything at all: the fused passes skip the others (a Parseval frame covers every frequency about
        // twice, a shearlet's spectrum vanishes on most rows) -- exact, those rows carry only zeros through the iteration
        p->sup_rows = p3d::plan64_shear_row_group(p->pf);
        p->sup_groups = (nil + p->sup_rows - 1) / p->sup_rows;
        const size_t nflag = (size_t)nsh * p->sup_groups;
        if ((e = hipMalloc((void**)&p->sup, nflag)) != hipSuccess) return bail("sup", e);
        if ((e = hipMemsetAsync(p->sup, 0, nflag, p->stream)) != hipSuccess) return bail("sup", e);
        rowsup64_kernel<<<dim3(p->sup_groups, nsh), 64, 0, p->stream>>>(p->psi, p->sup, nil, nxl, p->sup_rows, p->sup_groups);
        std::vector<unsigned char> host(nflag);
        if ((e = hipMemcpyAsync(host.data(), p->sup, nflag, hipMemcpyDeviceToHost, p->stream)) != hipSuccess) return bail("sup", e);
        if ((e = hipStreamSynchronize(p->stream)) != hipSuccess) return bail("sup", e);
        size_t on = 0;
        for (unsigned char f : host) on += f;
        p->sup_fraction = (double)on / (double)nflag;
    }
    if (p->pf && nil % 2 == 0 && nxl % 2 == 0 && !getenv("P3D_SHEARLET64_NO_PAIR")) {
        int* flag = nullptr;
        int asym = 1;
        if ((e = hipMalloc((void**)&flag, sizeof(int))) != hipSuccess) return bail("flag", e);
        e = hipMemsetAsync(flag, 0, sizeof(int), p->stream);
        if (e == hipSuccess) {
            psisym64_kernel<<<1024, 256, 0, p->stream>>>(p->psi, flag, nil, nxl, nsh);
            e = hipMemcpyAsync(&asym, flag, sizeof(int), hipMemcpyDeviceToHost, p->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        hipFree(flag);
        if (e != hipSuccess) return bail("symmetry check of Psi", e);
        p->pair = asym == 0;
    }
    *out = p;
    return P3D_OK;
}

// 1 when a plan for (nil, nxl) slices would run the fused passes (both extents on the double-precision register engine): what the host asks before it
// prefers this loop to the float32 one for a single-precision cube (functions/POCS.py)
int p3d_shearlet64_fused_shape(int nil, int nxl)
{
    const char* env = getenv("P3D_SHEARLET64_UNFUSED");
    if (env && env[0] == '1') return 0;
    return p3d::plan64_engine_shape(nil, nxl) ? 1 : 0;
}

int p3d_shearlet64_info(p3d_splan64* p, int* fused, double* row_group_fraction)
{
    if (!p || !fused) return s64fail(P3D_ERR_INVALID, "NULL argument");
    *fused = p->pf ? (p->pair ? 3 : 1) : 0;   // bit 1: real cubes take the Hermitian form (two columns per transform)
    if (row_group_fraction) *row_group_fraction = p->sup ? p->sup_fraction : 1.0;
    return P3D_OK;
}

// statistics of transform(x) for the schedule (POCS.py:257-258, 285, 318): stats HOST double [nslices][nsh][5] =
// Re, Im of the lexicographic (real cubes: signed) maximum, max |c|, min |c|, sum |c|^2 per shearlet.  x: host or device pointer,
// dtype P3D_C128 / P3D_F64 / P3D_C64 / P3D_F32.
int p3d_shearlet64_stats(p3d_splan64* p, const void* x, int dtype, int nslices, double* stats)
{
    S_RC(s_check(p, nslices, dtype));
    if (!x || !stats) return s64fail(P3D_ERR_INVALID, "NULL buffer");
    S_RC(take_x(p, x, dtype, nslices));
    if (p->pf) {
        S_RC(ensure_sums(p, (size_t)nslices));
        p3d::plan64_bind(p->pf, p->cur_x, nullptr);
        S_RC(p3d::plan64_shear_first(p->pf, dtype, p->sums, 0, 1.0, nslices, nullptr));
        const int pair = (p->pair && real_dtype(dtype)) ? 1 : 0;
        S_RC(p3d::plan64_shear_spread(p->pf, p->psi, p->U, nslices, p->nsh, nullptr, p->sup, pair ? p->nil / 2 + 1 : 0));
        S_RC(p3d::plan64_shear_cols(p->pf, p->U, nullptr, nslices, p->nsh, 0, 0, 0, 0, 1, 1.0 / ((double)p->nil * p->nxl), nullptr, p->sup, pair));
    } else {
        supdate64_kernel<<<dim3(p->nil, nslices), 256, 0, p->stream>>>(p->F, p->cur_x, dtype, nullptr, nullptr, p->rowsum, 0, 0, 0, 1.0, p->nil, p->nxl, nullptr, 0, 0);
        S_RC(s_forward(p, nslices, nullptr));
    }
    sstats64_kernel<<<nslices * p->nsh, 256, 0, p->stream>>>(p->U, p->per(), real_dtype(dtype) ? 1 : 0, p->stats);
    S_TRY(hipGetLastError());
    S_TRY(hipMemcpyAsync(stats, p->stats, sizeof(double) * 5 * (size_t)nslices * p->nsh, hipMemcpyDeviceToHost, p->stream));
    S_TRY(hipStreamSynchronize(p->stream));
    return P3D_OK;
}

// the loop (POCS.py:549-632 with the SHEARLET branches) in double precision; tau: HOST [nslices][niter][nsh][2] doubles; mask: DOUBLE
// [nil][nxl], host or device; x / out: host or device, dtype as above (complex64 / float32 cubes are widened on load, narrowed on store)
int p3d_shearlet64_run(p3d_splan64* p, const void* x, int dtype, const double* mask, const double* tau, const uint8_t* active, const p3d_pocs_params* prm,
                       void* out, int nslices, int32_t* niter_done, double* sums, double* elapsed_ms)
{
    S_RC(s_check(p, nslices, dtype));
    if (!x || !mask || !tau || !prm || !out) return s64fail(P3D_ERR_INVALID, "NULL argument");
    if (prm->niter < 1) return s64fail(P3D_ERR_INVALID, "niter must be >= 1");
    if (prm->thresh_op < P3D_OP_HARD || prm->thresh_op > P3D_OP_GARROTE)
        return s64fail(P3D_ERR_UNSUPPORTED, "thresh_op %d is not implemented for the shearlet transform", prm->thresh_op);
    const int niter = prm->niter, nsh = p->nsh;
    const bool early = prm->eps > 0.0, adaptive = prm->version == P3D_VER_ADAPTIVE, real_only = real_dtype(dtype);
    const size_t per = p->per();
    const size_t ntau = (size_t)nslices * niter * nsh, nsum = (size_t)(niter + 1) * nslices;
    if (real_only)
        for (size_t i = 0; i < ntau; ++i)
            if (tau[2 * i + 1] != 0.0) return s64fail(P3D_ERR_INVALID, "complex thresholds need a complex cube");
    if (p->tau_cap < ntau) {
        if (p->tau) hipFree(p->tau);
        p->tau = nullptr; p->tau_cap = 0;
        S_TRY(hipMalloc((void**)&p->tau, sizeof(c64) * ntau));
        p->tau_cap = ntau;
    }
    S_RC(ensure_sums(p, nsum));
    std::vector<int> done_h(nslices, 0);
    if (active) for (int s = 0; s < nslices; ++s) done_h[s] = active[s] ? 0 : -1;
    S_RC(take_x(p, x, dtype, nslices));
    const bool direct_out = on_plan_device(p, out);
    p->cur_out = direct_out ? out : p->st_out;
    S_TRY(hipMemcpyAsync(p->mask, mask, sizeof(double) * per, hipMemcpyDefault, p->stream));
    S_TRY(hipMemcpyAsync(p->tau, tau, sizeof(c64) * ntau, hipMemcpyHostToDevice, p->stream));
    S_TRY(hipMemcpyAsync(p->done, done_h.data(), sizeof(int) * nslices, hipMemcpyHostToDevice, p->stream));
    S_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nsum, p->stream));
    S_TRY(hipEventRecord(p->ev0, p->stream));
    if (p->pf) {
        const double scale = 1.0 / ((double)p->nil * p->nxl);
        const int pair = (p->pair && real_only) ? 1 : 0, rows = pair ? p->nil / 2 + 1 : 0;
        p3d::plan64_bind(p->pf, p->cur_x, p->cur_out);
        S_RC(p3d::plan64_shear_first(p->pf, dtype, p->sums, adaptive ? 1 : 0, prm->alpha, nslices, p->done));
        for (int k = 0; k < niter; ++k) {
            const bool last = k + 1 == niter;
            S_RC(p3d::plan64_shear_spread(p->pf, p->psi, p->U, nslices, nsh, p->done, p->sup, rows));
            S_RC(p3d::plan64_shear_cols(p->pf, p->U, p->tau, nslices, nsh, niter, k, prm->thresh_op, real_only ? 1 : 0, 0, scale, p->done, p->sup, pair));
            S_RC(p3d::plan64_shear_gather(p->pf, p->U, p->psi, nslices, nsh, p->done, p->sup, rows));
            if (pair) mirror_rows64_kernel<<<dim3(blocks_for((size_t)(p->nil / 2) * p->nxl, 1024), nslices), 256, 0, p->stream>>>(
                reinterpret_cast<c64*>(p3d::plan64_work(p->pf)), p->nil, p->nxl, p->done);
            S_RC(p3d::plan64_shear_back(p->pf, dtype, p->sums + (size_t)(k + 1) * nslices, last, adaptive ? 1 : 0, early ? 1 : 0, prm->alpha, nslices, p->done,
                                        last ? 1 : 0));
            if (early) sconv64_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
        }
    } else {
        const dim3 ugrid(p->nil, nslices);
        supdate64_kernel<<<ugrid, 256, 0, p->stream>>>(p->F, p->cur_x, dtype, p->mask, p->cur_out, p->rowsum, 0, adaptive ? 1 : 0, 0, prm->alpha, p->nil, p->nxl, p->done, 0,
                                                      real_only ? 1 : 0);
        srowsum64_kernel<<<nslices, 256, 0, p->stream>>>(p->rowsum, p->sums, p->nil, p->done);
        for (int k = 0; k < niter; ++k) {
            const bool last = k + 1 == niter;
            S_RC(s_forward(p, nslices, p->done));
            sthreshold64_kernel<<<dim3(blocks_for(per, 64), nslices * nsh), 256, 0, p->stream>>>(p->U, per, nsh, p->tau, niter, k, prm->thresh_op, real_only ? 1 : 0, p->done);
            S_RC(s_inverse(p, nslices, p->done));
            // (early exit: every iterate is stored, so that a slice that converges keeps its last one -- sconv64_kernel switches it off afterwards)
            supdate64_kernel<<<ugrid, 256, 0, p->stream>>>(p->F, p->cur_x, dtype, p->mask, p->cur_out, p->rowsum, 1, (adaptive && !last) ? 1 : 0, (early || last) ? 1 : 0,
                                                          prm->alpha, p->nil, p->nxl, p->done, last ? 1 : 0, real_only ? 1 : 0);
            srowsum64_kernel<<<nslices, 256, 0, p->stream>>>(p->rowsum, p->sums + (size_t)(k + 1) * nslices, p->nil, p->done);
            if (early) sconv64_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
        }
    }
    S_TRY(hipGetLastError());
    S_TRY(hipEventRecord(p->ev1, p->stream));
    S_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
    if (sums) S_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
    if (!direct_out) S_TRY(hipMemcpyAsync(out, p->st_out, elem_bytes(dtype) * per * nslices, hipMemcpyDefault, p->stream));
    S_TRY(hipStreamSynchronize(p->stream));   // (the caller may read `out` on any stream once this returns)
    if (niter_done) for (int s = 0; s < nslices; ++s) niter_done[s] = done_h[s] < 0 ? 0 : (done_h[s] > 0 ? done_h[s] : niter);
    if (elapsed_ms) {
        float ms = 0.f;
        S_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        *elapsed_ms = ms;
    }
    return P3D_OK;
}

}  // extern "C"
