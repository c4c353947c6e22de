// p3d_select.hip -- order statistics of a slice's spectrum for the 'data-driven' threshold model, on the device.
//
// get_threshold_decay(thresh_model='data-driven', transform_kind='FFT') (functions/POCS.py:356-362) picks the schedule from
// the forward-transformed input itself:
//     idx = (x_fwd > tau_min) & (x_fwd < tau_max);  v = np.sort(x_fwd[idx])[::-1]
//     tau[0] = v[0];  tau[i] = v[ceil(i (Nv - 1) / (niter - 1))]
// x_fwd is COMPLEX: NumPy orders complex numbers lexicographically (real part, then imaginary part), in the comparisons and in the
// sort.  (re, im) -> one 64-bit key whose unsigned order is that order; the keys of every slice are sorted once (rocPRIM segmented
// radix sort, descending), the two bounds are found by bisection, the niter picks are reads.  The host used to download the
// spectrum and sort it per slice (80 ms per 1024 x 1024 slice).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "p3d_select.hpp"

namespace p3d {

namespace {

// float -> uint32 whose unsigned order is the float order (-0.0 is made +0.0 first: NumPy compares them equal)
__device__ __forceinline__ uint32_t ord32(float v)
{
    const uint32_t u = __float_as_uint(v + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord32(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ uint64_t lex_key(float re, float im) { return ((uint64_t)ord32(re) << 32) | ord32(im); }

__global__ void lex_keys_kernel(uint64_t* inout, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const float2 v = reinterpret_cast<const float2*>(inout)[i];
        inout[i] = lex_key(v.x, v.y);
    }
}

__global__ void offsets_kernel(unsigned* off, unsigned per, int nslices)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nslices) off[i] = (unsigned)i * per;
}

__global__ void peaks_kernel(const uint64_t* sorted, size_t per, int nslices, float* peaks)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    const uint64_t k = sorted[(size_t)s * per];   // descending: the lexicographic maximum (x_fwd.max(), POCS.py:288)
    peaks[2 * s] = unord32((uint32_t)(k >> 32));
    peaks[2 * s + 1] = unord32((uint32_t)k);
}

// number of keys of the descending array d[0 .. n) that are > k (strict) or >= k
__device__ size_t count_above(const uint64_t* d, size_t n, uint64_t k, bool strict)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        const bool above = strict ? d[mid] > k : d[mid] >= k;
        if (above) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// one block per slice: bounds [s][4] = tau_min (re, im), tau_max (re, im)
__global__ void pick_kernel(const uint64_t* sorted, size_t per, int niter, const float* bounds, float* tau, long long* count)
{
    const int s = blockIdx.x;
    const uint64_t* d = sorted + (size_t)s * per;
    __shared__ size_t first, nv;
    if (threadIdx.x == 0) {
        const uint64_t klo = lex_key(bounds[4 * s], bounds[4 * s + 1]), khi = lex_key(bounds[4 * s + 2], bounds[4 * s + 3]);
        const size_t i0 = count_above(d, per, khi, false);   // x < tau_max starts here
        const size_t i1 = count_above(d, per, klo, true);    // x > tau_min ends here
        first = i0;
        nv = i1 > i0 ? i1 - i0 : 0;
        count[s] = (long long)nv;
    }
    __syncthreads();
    if (nv == 0) return;
    for (int i = threadIdx.x; i < niter; i += blockDim.x) {
        // ceil(i (Nv - 1) / (niter - 1)) in integers (POCS.py:361-362 take it in float64: the same integer)
        const size_t j = i == 0 || niter < 2 ? 0 : ((size_t)i * (nv - 1) + (size_t)(niter - 2)) / (size_t)(niter - 1);
        const uint64_t k = d[first + j];
        tau[((size_t)s * niter + i) * 2] = unord32((uint32_t)(k >> 32));
        tau[((size_t)s * niter + i) * 2 + 1] = unord32((uint32_t)k);
    }
}

}  // namespace

hipError_t lex_sort_desc(c32* spectrum, void* sorted, size_t per, int nslices, float* peaks_dev, hipStream_t st)
{
    const size_t count = per * (size_t)nslices;
    if (count > 0xffffffffull || per > 0xffffffffull) return hipErrorInvalidValue;
    uint64_t* keys = reinterpret_cast<uint64_t*>(spectrum);
    lex_keys_kernel<<<4096, 256, 0, st>>>(keys, count);
    unsigned* off = nullptr;
    hipError_t e = hipMalloc((void**)&off, sizeof(unsigned) * (nslices + 1));
    if (e != hipSuccess) return e;
    offsets_kernel<<<(nslices + 256) / 256, 256, 0, st>>>(off, (unsigned)per, nslices);
    size_t tmp_bytes = 0;
    void* tmp = nullptr;
    e = rocprim::segmented_radix_sort_keys_desc(nullptr, tmp_bytes, keys, reinterpret_cast<uint64_t*>(sorted), (unsigned)count, (unsigned)nslices,
                                                off, off + 1, 0, 64, st);
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 8);
    if (e == hipSuccess)
        e = rocprim::segmented_radix_sort_keys_desc(tmp, tmp_bytes, keys, reinterpret_cast<uint64_t*>(sorted), (unsigned)count, (unsigned)nslices,
                                                    off, off + 1, 0, 64, st);
    if (e == hipSuccess) {
        peaks_kernel<<<(nslices + 255) / 256, 256, 0, st>>>(reinterpret_cast<const uint64_t*>(sorted), per, nslices, peaks_dev);
        e = hipGetLastError();
    }
    const hipError_t es = hipStreamSynchronize(st);
    if (tmp) hipFree(tmp);
    hipFree(off);
    return e != hipSuccess ? e : es;
}

hipError_t data_driven_pick(const void* sorted, size_t per, int nslices, int niter, const float* bounds_dev, float* tau_dev, long long* count_dev,
                            hipStream_t st)
{
    pick_kernel<<<nslices, 128, 0, st>>>(reinterpret_cast<const uint64_t*>(sorted), per, niter, bounds_dev, tau_dev, count_dev);
    return hipGetLastError();
}

}  // namespace p3d
