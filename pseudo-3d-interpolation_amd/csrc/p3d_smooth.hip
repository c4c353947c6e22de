// p3d_smooth.hip -- the slice smoothing filters of step 15 (cube_postprocessing_3D.py:88-124 wraps scipy.ndimage.gaussian_filter /
// median_filter; applied per (iline, xline) slice at :628-660).  SciPy's semantics, restated:
//   gaussian_filter(x, sigma): separable correlation along axis 0 then axis 1 with w[j] ~ exp(-j^2 / (2 sigma^2)), |j| <= r =
//                              int(truncate * sigma + 0.5), normalised to sum 1; boundary mode 'reflect' (d c b a | a b c d | d c b a)
//   median_filter(x, size=s):  median of the s x s window centred on the sample (odd s), same boundary mode
// Simple gather kernels (one thread per output sample): this is a once-per-cube post-processing step, not the hot path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "p3d.h"
#include "p3d_internal.hpp"

namespace {

int mfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    p3d::set_last_error(buf);
    return code;
}
#define M_TRY(expr)                                                                                     \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return mfail(P3D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
};

// 'reflect' of scipy.ndimage: index i of an axis of n samples, any integer
__device__ __forceinline__ int reflect(int i, int n)
{
    const int per = 2 * n;
    int m = i % per;
    if (m < 0) m += per;
    return m >= n ? per - 1 - m : m;
}

// out[s][y][x] = sum_j w[j] in[s][reflect(y + j - r)][x]   (axis 0)   or along x (axis 1)
__global__ void gauss_axis_kernel(const float* in, float* out, const float* w, int r, int ny, int nx, int axis, size_t total)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t per = (size_t)ny * nx, s = i / per, e = i - s * per;
        const int y = (int)(e / nx), x = (int)(e - (size_t)y * nx);
        const float* base = in + s * per;
        float acc = 0.f;
        if (axis == 0) for (int j = -r; j <= r; ++j) acc += w[j + r] * base[(size_t)reflect(y + j, ny) * nx + x];
        else for (int j = -r; j <= r; ++j) acc += w[j + r] * base[(size_t)y * nx + reflect(x + j, nx)];
        out[i] = acc;
    }
}

// median of the S x S neighbourhood: the window sits in registers, odd-even transposition sort with compile-time indices
template <int S>
__global__ void median_kernel(const float* in, float* out, int ny, int nx, size_t total)
{
    constexpr int N = S * S, R = S / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t per = (size_t)ny * nx, s = i / per, e = i - s * per;
        const int y = (int)(e / nx), x = (int)(e - (size_t)y * nx);
        const float* base = in + s * per;
        float v[N];
#pragma unroll
        for (int dy = 0; dy < S; ++dy) {
            const size_t row = (size_t)reflect(y + dy - R, ny) * nx;
#pragma unroll
            for (int dx = 0; dx < S; ++dx) v[dy * S + dx] = base[row + reflect(x + dx - R, nx)];
        }
#pragma unroll
        for (int round = 0; round < N; ++round) {
#pragma unroll
            for (int k = round & 1; k + 1 < N; k += 2) {
                const float a = fminf(v[k], v[k + 1]), b = fmaxf(v[k], v[k + 1]);
                v[k] = a;
                v[k + 1] = b;
            }
        }
        out[i] = v[N / 2];
    }
}

unsigned blocks_for(size_t n) { const size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b)); }

int check(int device, const float* x, float* out, size_t nslices, int ny, int nx)
{
    if (!x || !out) return mfail(P3D_ERR_INVALID, "NULL buffer");
    if (nslices < 1 || ny < 1 || nx < 1) return mfail(P3D_ERR_INVALID, "bad shape");
    int ndev = 0;
    M_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return mfail(P3D_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    M_TRY(hipSetDevice(device));
    return P3D_OK;
}

}  // namespace

extern "C" {

int p3d_smooth_gaussian(int device, const float* x, size_t nslices, int ny, int nx, double sigma, double truncate, float* out)
{
    int rc = check(device, x, out, nslices, ny, nx);
    if (rc) return rc;
    if (!(sigma > 0.0) || !(truncate > 0.0)) return mfail(P3D_ERR_INVALID, "sigma and truncate must be positive");
    const int r = (int)(truncate * sigma + 0.5);
    if (r > 4096) return mfail(P3D_ERR_UNSUPPORTED, "kernel radius %d: up to 4096 samples", r);
    std::vector<double> wd(2 * r + 1);
    double sum = 0.0;
    for (int j = -r; j <= r; ++j) sum += wd[j + r] = std::exp(-0.5 * (double)j * j / (sigma * sigma));
    std::vector<float> w(2 * r + 1);
    for (int j = 0; j <= 2 * r; ++j) w[j] = (float)(wd[j] / sum);
    const size_t total = nslices * (size_t)ny * nx;
    DevBuf a, b, dw;
    M_TRY(hipMalloc(&a.p, sizeof(float) * total));
    M_TRY(hipMalloc(&b.p, sizeof(float) * total));
    M_TRY(hipMalloc(&dw.p, sizeof(float) * w.size()));
    M_TRY(hipMemcpy(a.p, x, sizeof(float) * total, hipMemcpyHostToDevice));
    M_TRY(hipMemcpy(dw.p, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
    gauss_axis_kernel<<<blocks_for(total), 256>>>((const float*)a.p, (float*)b.p, (const float*)dw.p, r, ny, nx, 0, total);
    gauss_axis_kernel<<<blocks_for(total), 256>>>((const float*)b.p, (float*)a.p, (const float*)dw.p, r, ny, nx, 1, total);
    M_TRY(hipGetLastError());
    M_TRY(hipDeviceSynchronize());
    M_TRY(hipMemcpy(out, a.p, sizeof(float) * total, hipMemcpyDeviceToHost));
    return P3D_OK;
}

int p3d_smooth_median(int device, const float* x, size_t nslices, int ny, int nx, int size, float* out)
{
    int rc = check(device, x, out, nslices, ny, nx);
    if (rc) return rc;
    if (size != 3 && size != 5 && size != 7) return mfail(P3D_ERR_UNSUPPORTED, "median window %d: 3, 5 and 7 are implemented", size);
    const size_t total = nslices * (size_t)ny * nx;
    DevBuf a, b;
    M_TRY(hipMalloc(&a.p, sizeof(float) * total));
    M_TRY(hipMalloc(&b.p, sizeof(float) * total));
    M_TRY(hipMemcpy(a.p, x, sizeof(float) * total, hipMemcpyHostToDevice));
    if (size == 3) median_kernel<3><<<blocks_for(total), 256>>>((const float*)a.p, (float*)b.p, ny, nx, total);
    else if (size == 5) median_kernel<5><<<blocks_for(total), 256>>>((const float*)a.p, (float*)b.p, ny, nx, total);
    else median_kernel<7><<<blocks_for(total), 256>>>((const float*)a.p, (float*)b.p, ny, nx, total);
    M_TRY(hipGetLastError());
    M_TRY(hipDeviceSynchronize());
    M_TRY(hipMemcpy(out, b.p, sizeof(float) * total, hipMemcpyDeviceToHost));
    return P3D_OK;
}

}  // extern "C"
