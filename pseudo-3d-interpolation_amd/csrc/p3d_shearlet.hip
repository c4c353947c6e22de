// p3d_shearlet.hip -- SHEARLET variant of the POCS path (transform_kind = 'SHEARLET').
//
// The reference hands FFST.shearletTransformSpect / inverseShearletTransformSpect with precomputed spectra Psi to
// POCS_algorithm (pseudo_3D_interpolation/cube_POCS_interpolation_3D.py:269-274; used at functions/POCS.py:526-527, 589-590,
// 610-611) and thresholds every shearlet with its own tau by broadcasting over the last axis (POCS.py:598 with a (nsh,) tau).
// FFST (PyShearlets) is a third-party package that is neither part of the reference nor installed in this image; the transform
// is the frequency-domain frame of its tutorial (see oracle/shearlet_oracle.py, "parity unpinned" for that part):
//     ST_s = ifft2(Psi_s * fft2(x)),        x = ifft2(sum_s fft2(ST_s) * Psi_s),        sum_s Psi_s^2 = 1.
// Psi comes from the caller (auxiliary_data), as in the reference.
//
// Coefficients of a slice are nsh full-size arrays (125 x 16 MiB for configs[4]); they are never sent to the host.  One
// iteration of a batch of nb slices is  fft2(feed) -> spread over the shearlets (x Psi_s) -> batched ifft2 (nb*nsh slices of the
// FFT engine of p3d_fft.hpp) -> threshold -> batched fft2 -> weighted sum over the shearlets -> ifft2 -> re-insertion.
// float32 cubes keep real coefficients (the real part is taken where FFST takes it: after the inverse FFTs).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "p3d.h"
#include "p3d_fft.hpp"
#include "p3d_internal.hpp"
#include "p3d_shrink.hpp"

using p3d::c32;

namespace {

int sfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    p3d::set_last_error(buf);
    return code;
}
#define S_TRY(expr)                                                                                     \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return sfail(P3D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define S_RC(expr)                \
    do {                          \
        int rc_ = (expr);         \
        if (rc_) return rc_;      \
    } while (0)

inline unsigned blocks_for(size_t n, unsigned cap = 4096) { const size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); }

// U[b][s][i] = Psi[s][i] * F[b][i]          (grid.y = slice of the batch)
__global__ void spread_kernel(const c32* F, const float* psi, c32* U, size_t per, int nsh, const int* done)
{
    const int b = blockIdx.y;
    if (done && done[b] != 0) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        const c32 f = F[(size_t)b * per + i];
        for (int s = 0; s < nsh; ++s) {
            const float w = psi[(size_t)s * per + i];
            U[((size_t)b * nsh + s) * per + i] = c32{f.x * w, f.y * w};
        }
    }
}

// A[b][i] = sum_s U[b][s][i] * Psi[s][i]
__global__ void gather_kernel(const c32* U, const float* psi, c32* A, size_t per, int nsh, const int* done)
{
    const int b = blockIdx.y;
    if (done && done[b] != 0) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        float ar = 0.f, ai = 0.f;
        for (int s = 0; s < nsh; ++s) {
            const float w = psi[(size_t)s * per + i];
            const c32 u = U[((size_t)b * nsh + s) * per + i];
            ar += u.x * w;
            ai += u.y * w;
        }
        A[(size_t)b * per + i] = c32{ar, ai};
    }
}

// coefficients of shearlet s of slice b: threshold with tau[b][iter][s] (grid.y = b*nsh + s); real_only drops the imaginary
// round-off first (FFST returns ST.real for real data)
__global__ void sthreshold_kernel(c32* U, size_t per, int nsh, const c32* tau, int niter, int iter, int op, int real_only, const int* done)
{
    const int bs = blockIdx.y, b = bs / nsh, s = bs - b * nsh;
    if (done && done[b] != 0) return;
    const c32 t = tau[((size_t)b * niter + iter) * nsh + s];
    c32* p = U + (size_t)bs * per;
    const p3d::Shrink shr(t, op);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        const c32 v = p[i];
        p[i] = real_only ? c32{p3d::shrink(v.x, t, op), 0.f} : shr(v);
    }
}

// bit g of sup[s * words + g / 32] = rows 8 g ... 8 g + 7 of Psi_s hold a non-zero sample (one wavefront per (s, g); sup zeroed before)
__global__ void rowsup_kernel(const float* psi, unsigned* sup, int nil, int nxl, int words)
{
    const int s = blockIdx.y, g = blockIdx.x;
    const int r0 = 8 * g, r1 = min(r0 + 8, nil);
    const float* p = psi + ((size_t)s * nil + r0) * nxl;
    const size_t n = (size_t)(r1 - r0) * nxl;
    bool any = false;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) any = any || (p[i] != 0.0f);
    if (__ballot(any) != 0ull && threadIdx.x == 0) atomicOr(sup + (size_t)s * words + (g >> 5), 1u << (g & 31));
}

// *asym is raised when some Psi_s(-k) != Psi_s(k) (exact comparison): real slices then have complex coefficients, and the
// two-columns-per-transform column pass (p3d_col_shear.hpp), which relies on real ones, stays off
__global__ void psisym_kernel(const float* psi, int* asym, int nil, int nxl, int nsh)
{
    const size_t per = (size_t)nil * nxl, total = per * nsh;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t s = i / per, r = i - s * per;
        const int k1 = (int)(r / nxl), k2 = (int)(r - (size_t)k1 * nxl);
        const int m1 = k1 ? nil - k1 : 0, m2 = k2 ? nxl - k2 : 0;
        bad = bad || (psi[i] != psi[s * per + (size_t)m1 * nxl + m2]);
    }
    if (bad) atomicOr(asym, 1);
}

// F[b][k1][k2] = conj F[b][n1 - k1][(n2 - k2) mod n2] for k1 = n1/2 + 1 ... n1 - 1: the spectrum of a real slice, completed from the
// rows the gather pass computed (Hermitian work slices, ShearArgs::half)
__global__ void mirror_rows_kernel(c32* F, int n1, int n2)
{
    c32* f = F + (size_t)blockIdx.y * n1 * n2;
    const size_t total = (size_t)(n1 - n1 / 2 - 1) * n2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k1 = n1 / 2 + 1 + (int)(i / n2), k2 = (int)(i % n2);
        const c32 m = f[(size_t)(n1 - k1) * n2 + (k2 ? n2 - k2 : 0)];
        f[(size_t)k1 * n2 + k2] = c32{m.x, -m.y};
    }
}

// per (slice, shearlet): lexicographic (real data: signed) maximum, max |c|, min |c|, sum |c|^2 -> stats[(b*nsh + s)*5 ..]
__global__ void sstats_kernel(const c32* U, size_t per, int real_only, float* stats)
{
    __shared__ float sh[256 * 5];
    const int bs = blockIdx.x;
    const c32* p = U + (size_t)bs * per;
    float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY, sq = 0.f;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) {
        c32 v = p[i];
        if (real_only) v.y = 0.f;
        const float q = v.x * v.x + v.y * v.y;
        if (v.x > lr || (v.x == lr && v.y > li)) { lr = v.x; li = v.y; }
        mx = fmaxf(mx, q);
        mn = fminf(mn, q);
        sq += q;
    }
    float* me = sh + threadIdx.x * 5;
    me[0] = lr; me[1] = li; me[2] = mx; me[3] = mn; me[4] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = sq;
        for (int t = 1; t < (int)blockDim.x; ++t) {
            const float* o = sh + t * 5;
            if (o[0] > lr || (o[0] == lr && o[1] > li)) { lr = o[0]; li = o[1]; }
            mx = fmaxf(mx, o[2]);
            mn = fminf(mn, o[3]);
            tot += o[4];
        }
        float* q = stats + (size_t)bs * 5;
        q[0] = lr; q[1] = li; q[2] = sqrtf(mx); q[3] = sqrtf(mn); q[4] = (float)tot;
    }
}

// mode 0: first input (feed = x or its APOCS mix; sums += |x|)
// mode 1: re-insertion (POCS.py:616-619) of A = ifft2(...), sums += |x_new|, feed for the next iteration
__global__ void supdate_kernel(const c32* A, c32* feed, const void* x, int dtype, const float* mask, void* out, double* sums, int mode, int adaptive,
                               int write_out, float alpha, size_t per, const int* done, int zero_fill)
{
    __shared__ double sh[256];
    const int b = blockIdx.y;
    const int dn = done ? done[b] : 0;
    const bool real_only = dtype == P3D_F32;
    if (zero_fill && dn < 0)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
            if (real_only) reinterpret_cast<float*>(out)[(size_t)b * per + i] = 0.f;
            else reinterpret_cast<c32*>(out)[(size_t)b * per + i] = c32{0.f, 0.f};
        }
    double acc = 0.0;
    if (dn == 0) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
            const size_t g = (size_t)b * per + i;
            const c32 xo = real_only ? c32{reinterpret_cast<const float*>(x)[g], 0.f} : reinterpret_cast<const c32*>(x)[g];
            const float m = mask ? mask[i] : 0.f;
            const float wgt = 1.0f - alpha * m;
            c32 xn;
            if (mode == 0) {
                xn = xo;
            } else {
                c32 a = A[g];
                if (real_only) a.y = 0.f;
                xn = c32{a.x * wgt + xo.x * alpha, a.y * wgt + xo.y * alpha};
                if (write_out) {
                    if (real_only) reinterpret_cast<float*>(out)[g] = xn.x;
                    else reinterpret_cast<c32*>(out)[g] = xn;
                }
            }
            acc += (double)sqrtf(xn.x * xn.x + xn.y * xn.y);
            if (adaptive) {
                const float c1 = 1.0f - alpha;
                feed[g] = c32{(xo.x * alpha + xn.x * wgt) + (xo.x - xn.x * m) * c1, (xo.y * alpha + xn.y * wgt) + (xo.y - xn.y * m) * c1};
            } else {
                feed[g] = xn;
            }
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && dn == 0) atomicAdd(sums + b, sh[0]);
}

__global__ void sconv_kernel(const double* sums, int* done, int nslices, int iter, double eps)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices || done[s] != 0) return;
    const double cur = sums[(size_t)(iter + 1) * nslices + s], prev = sums[(size_t)iter * nslices + s];
    const double d = cur - prev;
    if (iter > 2 && (d * d) / (cur * cur) < eps) done[s] = iter + 1;
}

}  // namespace

struct p3d_splan {
    int device = 0, nil = 0, nxl = 0, nsh = 0, max_slices = 0;
    p3d_plan* fft = nullptr;  // batched 2-D FFT of up to max_slices * nsh slices
    hipStream_t stream = nullptr;
    float* psi = nullptr;      // [nsh][nil][nxl]
    unsigned* sup = nullptr;   // [nsh][sup_words] bitmap of the 8-row groups on which Psi_s does not vanish (fused passes; nullptr: dense)
    int sup_words = 0;
    double sup_fraction = 1.0; // share of the (shearlet, row group) pairs that are on
    bool pair = false;         // float32 cubes: two columns per transform in the column pass (spectra symmetric; P3D_SHEARLET_NO_PAIR unset)
    bool fused = false;        // three fused passes per iteration (power-of-two extents); P3D_SHEARLET_UNFUSED=1 disables
    c32 *U = nullptr, *F = nullptr, *feed = nullptr, *tau = nullptr;
    size_t tau_cap = 0, sums_cap = 0;
    double* sums = nullptr;
    int* done = nullptr;
    float *stats = nullptr, *mask = nullptr;
    void *st_x = nullptr, *st_out = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t per() const { return (size_t)nil * nxl; }
};

extern "C" int p3d_shearlet_plan_destroy(p3d_splan* p)
{
    if (!p) return P3D_OK;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    void* bufs[] = {p->psi, p->sup, p->U, p->F, p->feed, p->tau, p->sums, p->done, p->stats, p->mask, p->st_x, p->st_out};
    for (void* b : bufs) if (b) hipFree(b);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->fft) p3d_plan_destroy(p->fft);
    delete p;
    return P3D_OK;
}

extern "C" int p3d_shearlet_plan_create(p3d_splan** out, int device, int nil, int nxl, int nsh, const float* psi, int max_slices)
{
    if (!out || !psi) return sfail(P3D_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (nil < 1 || nxl < 1 || nsh < 1 || max_slices < 1) return sfail(P3D_ERR_INVALID, "bad shape / batch size");
    if ((long long)max_slices * nsh > 65535) return sfail(P3D_ERR_INVALID, "max_slices * nsh = %lld exceeds 65535", (long long)max_slices * nsh);
    p3d_splan* p = new p3d_splan;
    p->device = device; p->nil = nil; p->nxl = nxl; p->nsh = nsh; p->max_slices = max_slices;
    int rc = p3d_plan_create(&p->fft, device, nil, nxl, max_slices * nsh);
    if (rc) { delete p; return rc; }  // message already set
    p->stream = p3d::plan_stream(p->fft);
    auto bail = [&](const char* what, hipError_t e) {
        p3d_shearlet_plan_destroy(p);
        return sfail(P3D_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
#define ALLOC(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return bail(#ptr, e)
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipEventCreate(&p->ev0)) != hipSuccess) return bail("event", e);
    if ((e = hipEventCreate(&p->ev1)) != hipSuccess) return bail("event", e);
    const size_t S = (size_t)max_slices, per = p->per();
    ALLOC(p->psi, sizeof(float) * per * nsh);
    ALLOC(p->U, sizeof(c32) * per * nsh * S);
    ALLOC(p->F, sizeof(c32) * per * S);
    ALLOC(p->feed, sizeof(c32) * per * S);
    ALLOC(p->done, sizeof(int) * S);
    ALLOC(p->stats, sizeof(float) * 5 * nsh * S);
    ALLOC(p->mask, sizeof(float) * per);
    ALLOC(p->st_x, sizeof(c32) * per * S);
    ALLOC(p->st_out, sizeof(c32) * per * S);
#undef ALLOC
    if ((e = hipMemcpy(p->psi, psi, sizeof(float) * per * nsh, hipMemcpyHostToDevice)) != hipSuccess) return bail("upload of Psi", e);
    const char* env = getenv("P3D_SHEARLET_UNFUSED");
    p->fused = p3d::shearlet_fused_supported(p->fft) && !(env && env[0] == '1');
    if (p->fused && !getenv("P3D_SHEARLET_NO_SUPPORT")) {
        // which 8-row groups of every spectrum hold anything at all: the fused passes skip the others (ShearArgs::sup)
        const int groups = (nil + 7) / 8, words = (groups + 31) / 32;
        if ((e = hipMalloc((void**)&p->sup, sizeof(unsigned) * (size_t)nsh * words)) != hipSuccess) return bail("sup", e);
        if ((e = hipMemsetAsync(p->sup, 0, sizeof(unsigned) * (size_t)nsh * words, p->stream)) != hipSuccess) return bail("sup", e);
        rowsup_kernel<<<dim3(groups, nsh), 64, 0, p->stream>>>(p->psi, p->sup, nil, nxl, words);
        std::vector<unsigned> host((size_t)nsh * words);
        if ((e = hipMemcpyAsync(host.data(), p->sup, sizeof(unsigned) * host.size(), hipMemcpyDeviceToHost, p->stream)) != hipSuccess) return bail("sup", e);
        if ((e = hipStreamSynchronize(p->stream)) != hipSuccess) return bail("sup", e);
        size_t on = 0;
        for (unsigned w : host) on += (size_t)__builtin_popcount(w);
        p->sup_words = words;
        p->sup_fraction = (double)on / ((double)groups * nsh);
    }
    if (p->fused && !getenv("P3D_SHEARLET_NO_PAIR")) {
        int* flag = nullptr;
        int asym = 1;
        if ((e = hipMalloc((void**)&flag, sizeof(int))) != hipSuccess) return bail("flag", e);
        e = hipMemsetAsync(flag, 0, sizeof(int), p->stream);
        if (e == hipSuccess) {
            psisym_kernel<<<1024, 256, 0, p->stream>>>(p->psi, flag, nil, nxl, nsh);
            e = hipMemcpyAsync(&asym, flag, sizeof(int), hipMemcpyDeviceToHost, p->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        hipFree(flag);
        if (e != hipSuccess) return bail("symmetry check of Psi", e);
        p->pair = asym == 0 && p3d::shearlet_pair_supported(p->fft);
    }
    *out = p;
    return P3D_OK;
}

static int s_check(p3d_splan* p, int nslices, int dtype)
{
    if (!p) return sfail(P3D_ERR_INVALID, "NULL plan");
    if (nslices < 1 || nslices > p->max_slices) return sfail(P3D_ERR_INVALID, "nslices = %d outside 1..max_slices (%d)", nslices, p->max_slices);
    if (dtype != P3D_C64 && dtype != P3D_F32) return sfail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    S_TRY(hipSetDevice(p->device));
    return P3D_OK;
}

// feed -> U = coefficients of all shearlets (spatial domain)
static int s_forward(p3d_splan* p, int ns, bool real_only, const int* done)
{
    const size_t per = p->per();
    S_RC(p3d::fft2_async(p->fft, p->feed, p->F, ns, 0));
    spread_kernel<<<dim3(blocks_for(per, 1024), ns), 256, 0, p->stream>>>(p->F, p->psi, p->U, per, p->nsh, done);
    S_RC(p3d::fft2_async(p->fft, p->U, p->U, ns * p->nsh, 1));
    (void)real_only;
    S_TRY(hipGetLastError());
    return P3D_OK;
}

// U (coefficients) -> F = the slice
static int s_inverse(p3d_splan* p, int ns, const int* done)
{
    const size_t per = p->per();
    S_RC(p3d::fft2_async(p->fft, p->U, p->U, ns * p->nsh, 0));
    gather_kernel<<<dim3(blocks_for(per, 1024), ns), 256, 0, p->stream>>>(p->U, p->psi, p->F, per, p->nsh, done);
    S_RC(p3d::fft2_async(p->fft, p->F, p->F, ns, 1));
    S_TRY(hipGetLastError());
    return P3D_OK;
}

extern "C" {

int p3d_shearlet_info(p3d_splan* p, double* row_group_fraction, int* paired)
{
    if (!p || !row_group_fraction) return sfail(P3D_ERR_INVALID, "NULL argument");
    *row_group_fraction = p->sup ? p->sup_fraction : 1.0;
    if (paired) *paired = p->pair ? 1 : 0;
    return P3D_OK;
}

// test hooks: x HOST complex64 [nslices][nil][nxl] <-> st HOST complex64 [nslices][nsh][nil][nxl]
int p3d_shearlet_transform_c64(p3d_splan* p, const void* x, void* st, int nslices)
{
    S_RC(s_check(p, nslices, P3D_C64));
    if (!x || !st) return sfail(P3D_ERR_INVALID, "NULL buffer");
    S_TRY(hipMemcpy(p->feed, x, sizeof(c32) * p->per() * nslices, hipMemcpyHostToDevice));
    S_RC(s_forward(p, nslices, false, nullptr));
    S_TRY(hipStreamSynchronize(p->stream));
    S_TRY(hipMemcpy(st, p->U, sizeof(c32) * p->per() * p->nsh * nslices, hipMemcpyDeviceToHost));
    return P3D_OK;
}

int p3d_shearlet_inverse_c64(p3d_splan* p, const void* st, void* x, int nslices)
{
    S_RC(s_check(p, nslices, P3D_C64));
    if (!x || !st) return sfail(P3D_ERR_INVALID, "NULL buffer");
    S_TRY(hipMemcpy(p->U, st, sizeof(c32) * p->per() * p->nsh * nslices, hipMemcpyHostToDevice));
    S_RC(s_inverse(p, nslices, nullptr));
    S_TRY(hipStreamSynchronize(p->stream));
    S_TRY(hipMemcpy(x, p->F, sizeof(c32) * p->per() * nslices, hipMemcpyDeviceToHost));
    return P3D_OK;
}

// statistics of transform(x) for the schedule (POCS.py:257-258, 285, 318): stats HOST double [nslices][nsh][5] =
// Re, Im of the lexicographic (float32 cubes: signed) maximum, max |c|, min |c|, sum |c|^2 per shearlet
int p3d_shearlet_stats(p3d_splan* p, const void* x, int dtype, int nslices, double* stats)
{
    S_RC(s_check(p, nslices, dtype));
    if (!x || !stats) return sfail(P3D_ERR_INVALID, "NULL buffer");
    const size_t esz = dtype == P3D_C64 ? sizeof(c32) : sizeof(float);
    // x: host or device pointer.  On the plan's stream: a device-to-device hipMemcpy runs on the null stream and need not have finished when it
    // returns, and the plan's stream does not wait for the null stream -- the kernels below would read st_x early (seen as wrong statistics
    // when several processes share the GPU)
    S_TRY(hipMemcpyAsync(p->st_x, x, esz * p->per() * nslices, hipMemcpyDefault, p->stream));
    if (p->sums_cap < (size_t)nslices) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr; p->sums_cap = 0;
        S_TRY(hipMalloc((void**)&p->sums, sizeof(double) * 2 * p->max_slices));
        p->sums_cap = 2 * (size_t)p->max_slices;
    }
    S_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nslices, p->stream));
    supdate_kernel<<<dim3(blocks_for(p->per(), 256), nslices), 256, 0, p->stream>>>(nullptr, p->feed, p->st_x, dtype, nullptr, nullptr, p->sums, 0, 0, 0, 1.0f,
                                                                                  p->per(), nullptr, 0);
    std::vector<float> host((size_t)nslices * p->nsh * 5);
    if (p->pair && dtype == P3D_F32) {
        // float32 cubes on symmetric spectra: the coefficients through the passes of the loop (support rows only, Hermitian half, two
        // columns per transform) -- 3 ms instead of 26 for 8 slices of configs[4]; the statistics agree with the general route to rounding
        S_RC(p3d::fft2_async(p->fft, p->feed, p->F, nslices, 0));
        S_RC(p3d::shearlet_spread_inv(p->fft, p->F, p->psi, nslices, p->nsh, p->sup, p->sup_words, true));
        S_RC(p3d::shearlet_col_stats_pair(p->fft, nslices, p->nsh, p->sup, p->sup_words, host.data()));
    } else {
        S_RC(s_forward(p, nslices, dtype == P3D_F32, nullptr));
        sstats_kernel<<<nslices * p->nsh, 256, 0, p->stream>>>(p->U, p->per(), dtype == P3D_F32, p->stats);
        S_TRY(hipGetLastError());
        S_TRY(hipMemcpyAsync(host.data(), p->stats, sizeof(float) * host.size(), hipMemcpyDeviceToHost, p->stream));
        S_TRY(hipStreamSynchronize(p->stream));
    }
    for (size_t i = 0; i < host.size(); ++i) stats[i] = host[i];
    return P3D_OK;
}

// the loop (POCS.py:549-632 with the SHEARLET branches); tau: HOST [nslices][niter][nsh][2] doubles
int p3d_shearlet_run(p3d_splan* p, const void* x, int dtype, const float* mask, const double* tau, const uint8_t* active, const p3d_pocs_params* prm,
                     void* out, int nslices, int32_t* niter_done, double* sums, double* elapsed_ms)
{
    S_RC(s_check(p, nslices, dtype));
    if (!x || !mask || !tau || !prm || !out) return sfail(P3D_ERR_INVALID, "NULL argument");
    if (prm->niter < 1) return sfail(P3D_ERR_INVALID, "niter must be >= 1");
    if (prm->thresh_op < P3D_OP_HARD || prm->thresh_op > P3D_OP_GARROTE)
        return sfail(P3D_ERR_UNSUPPORTED, "thresh_op %d is not implemented for the shearlet transform", prm->thresh_op);
    const int niter = prm->niter, nsh = p->nsh;
    const bool early = prm->eps > 0.0, adaptive = prm->version == P3D_VER_ADAPTIVE, real_only = dtype == P3D_F32;
    const size_t per = p->per(), esz = real_only ? sizeof(float) : sizeof(c32);
    const size_t ntau = (size_t)nslices * niter * nsh, nsum = (size_t)(niter + 1) * nslices;
    if (p->tau_cap < ntau) {
        if (p->tau) hipFree(p->tau);
        p->tau = nullptr; p->tau_cap = 0;
        S_TRY(hipMalloc((void**)&p->tau, sizeof(c32) * ntau));
        p->tau_cap = ntau;
    }
    if (p->sums_cap < nsum) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr; p->sums_cap = 0;
        S_TRY(hipMalloc((void**)&p->sums, sizeof(double) * nsum));
        p->sums_cap = nsum;
    }
    std::vector<c32> tau_f(ntau);
    for (size_t i = 0; i < ntau; ++i) {
        tau_f[i] = p3d::tau_for_device(tau[2 * i], tau[2 * i + 1], prm->thresh_op == P3D_OP_HARD);
        if (real_only && tau[2 * i + 1] != 0.0) return sfail(P3D_ERR_INVALID, "complex thresholds need a complex64 cube");
    }
    std::vector<int> done_h(nslices, 0);
    if (active) for (int s = 0; s < nslices; ++s) done_h[s] = active[s] ? 0 : -1;
    S_TRY(hipMemcpyAsync(p->st_x, x, esz * per * nslices, hipMemcpyDefault, p->stream));   // x, mask, out: host or device pointers (on the plan's stream, see p3d_shearlet_stats)
    S_TRY(hipMemcpyAsync(p->mask, mask, sizeof(float) * per, hipMemcpyDefault, p->stream));
    S_TRY(hipMemcpyAsync(p->tau, tau_f.data(), sizeof(c32) * ntau, hipMemcpyHostToDevice, p->stream));
    S_TRY(hipMemcpyAsync(p->done, done_h.data(), sizeof(int) * nslices, hipMemcpyHostToDevice, p->stream));
    S_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nsum, p->stream));
    S_TRY(hipEventRecord(p->ev0, p->stream));
    const dim3 ugrid(blocks_for(per, 256), nslices);
    supdate_kernel<<<ugrid, 256, 0, p->stream>>>(nullptr, p->feed, p->st_x, dtype, p->mask, p->st_out, p->sums, 0, adaptive ? 1 : 0, 0, (float)prm->alpha, per,
                                                p->done, 0);
    for (int k = 0; k < niter; ++k) {
        const bool last = k + 1 == niter;
        if (p->fused) {
            // three passes over the coefficients instead of twelve: spectra x Psi_s folded into the inverse row pass, the
            // threshold into the column pass between its two transforms, x Psi_s and the sum over s into the forward row pass
            S_RC(p3d::fft2_async(p->fft, p->feed, p->F, nslices, 0));
            const bool pair = p->pair && real_only;
            S_RC(p3d::shearlet_spread_inv(p->fft, p->F, p->psi, nslices, nsh, p->sup, p->sup_words, pair));
            S_RC(p3d::shearlet_col_shrink(p->fft, p->tau, nslices, nsh, niter, k, prm->thresh_op, real_only ? 1 : 0, p->sup, p->sup_words, pair));
            S_RC(p3d::shearlet_gather_fwd(p->fft, p->psi, p->F, nslices, nsh, p->sup, p->sup_words, pair));
            if (pair) mirror_rows_kernel<<<dim3(blocks_for((size_t)(p->nil / 2) * p->nxl, 512), nslices), 256, 0, p->stream>>>(p->F, p->nil, p->nxl);
            S_RC(p3d::fft2_async(p->fft, p->F, p->F, nslices, 1));
        } else {
            S_RC(s_forward(p, nslices, real_only, p->done));
            sthreshold_kernel<<<dim3(blocks_for(per, 64), nslices * nsh), 256, 0, p->stream>>>(p->U, per, nsh, p->tau, niter, k, prm->thresh_op,
                                                                                          real_only ? 1 : 0, p->done);
            S_RC(s_inverse(p, nslices, p->done));
        }
        supdate_kernel<<<ugrid, 256, 0, p->stream>>>(p->F, p->feed, p->st_x, dtype, p->mask, p->st_out, p->sums + (size_t)(k + 1) * nslices, 1,
                                                    (adaptive && !last) ? 1 : 0, (early || last) ? 1 : 0, (float)prm->alpha, per, p->done, last ? 1 : 0);
        if (early) sconv_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
    }
    S_TRY(hipGetLastError());
    S_TRY(hipEventRecord(p->ev1, p->stream));
    S_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
    if (sums) S_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
    S_TRY(hipStreamSynchronize(p->stream));
    S_TRY(hipMemcpyAsync(out, p->st_out, esz * per * nslices, hipMemcpyDefault, p->stream));
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
