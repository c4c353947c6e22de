// p3d_select.hip -- order statistics of a slice's spectrum for the 'data-driven' threshold model, on the device.
//
// get_threshold_decay(thresh_model='data-driven', transform_kind='FFT') (functions/POCS.py:356-362) picks the schedule from
// the forward-transformed input itself:
//     idx = (x_fwd > tau_min) & (x_fwd < tau_max);  v = np.sort(x_fwd[idx])[::-1]
//     tau[0] = v[0];  tau[i] = v[ceil(i (Nv - 1) / (niter - 1))]
// x_fwd is COMPLEX: NumPy orders complex numbers lexicographically (real part, then imaginary part), in the comparisons and in the
// sort.  (re, im) -> one 64-bit key whose unsigned order is that order; the keys of every slice are sorted once (a hand-written segmented
// least-significant-digit radix sort, descending: sixteen 4-bit passes of histogram / scan / stable scatter -- round 4; rocPRIM's
// segmented sort before), the two bounds are found by bisection, the niter picks are reads.  The host used to download the
// spectrum and sort it per slice (80 ms per 1024 x 1024 slice).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

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

// ---- segmented radix sort, descending, one segment per slice ---------------------------------------------------------------------------
// 4 bits per pass (bucket = 15 - digit: the largest digit first), tiles of 4096 keys (256 threads x 16 consecutive keys: a thread's keys,
// then the threads, then the tiles are in index order, so ranks counted that way keep the pass stable).  Off the iteration loop, once per
// job of the 'data-driven' model: clarity over speed (48 launches, ~50 ms for 512 slices of 2^20 keys).
constexpr int SORT_T = 256, SORT_K = 16, SORT_TILE = SORT_T * SORT_K;
__device__ __forceinline__ unsigned sort_bucket(uint64_t k, int shift) { return 15u - (unsigned)((k >> shift) & 15ull); }

// hist[(slice * 16 + bucket) * tiles + tile] = keys of the tile in the bucket
__global__ void sort_hist_kernel(const uint64_t* src, unsigned* hist, size_t per, int tiles, int shift)
{
    __shared__ unsigned cnt[16];
    const int tile = blockIdx.x, slice = blockIdx.y;
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t* s = src + (size_t)slice * per;
    const size_t lo = (size_t)tile * SORT_TILE;
    for (int j = 0; j < SORT_K; ++j) {
        const size_t i = lo + (size_t)j * SORT_T + threadIdx.x;   // (any order will do for counting: coalesced)
        if (i < per) atomicAdd(&cnt[sort_bucket(s[i], shift)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16) hist[((size_t)slice * 16 + threadIdx.x) * tiles + tile] = cnt[threadIdx.x];
}
// exclusive prefix of a slice's counts in (bucket, tile) order, in place: where the tile's keys of that bucket start in the sorted slice
__global__ void sort_scan_kernel(unsigned* hist, int tiles)
{
    __shared__ unsigned total[16];
    const int slice = blockIdx.x, b = threadIdx.x;   // 16 threads
    unsigned* h = hist + ((size_t)slice * 16 + b) * tiles;
    unsigned t = 0;
    for (int i = 0; i < tiles; ++i) t += h[i];
    total[b] = t;
    __syncthreads();
    unsigned run = 0;
    for (int q = 0; q < b; ++q) run += total[q];
    for (int i = 0; i < tiles; ++i) { const unsigned c = h[i]; h[i] = run; run += c; }
}
__global__ __launch_bounds__(SORT_T) void sort_scatter_kernel(const uint64_t* src, uint64_t* dst, const unsigned* offs, size_t per, int tiles, int shift)
{
    __shared__ unsigned cnt[16][SORT_T];   // [bucket][thread]: keys of the thread in the bucket, then their exclusive prefix over the threads
    const int tile = blockIdx.x, slice = blockIdx.y, t = threadIdx.x;
    const uint64_t* s = src + (size_t)slice * per;
    uint64_t* d = dst + (size_t)slice * per;
    const size_t i0 = (size_t)tile * SORT_TILE + (size_t)t * SORT_K;   // a thread's 16 CONSECUTIVE keys
#pragma unroll
    for (int b = 0; b < 16; ++b) cnt[b][t] = 0u;
    uint64_t key[SORT_K];
    unsigned rank[SORT_K];
#pragma unroll
    for (int j = 0; j < SORT_K; ++j) {
        key[j] = i0 + j < per ? s[i0 + j] : 0ull;
        if (i0 + j < per) {
            const unsigned b = sort_bucket(key[j], shift);
            rank[j] = cnt[b][t];          // keys of this thread, of this bucket, before this one (the thread owns its column: no race)
            cnt[b][t] = rank[j] + 1u;
        } else {
            rank[j] = 0u;
        }
    }
    __syncthreads();
    if (t < 16) {   // thread b: exclusive prefix of its bucket's row over the 256 threads
        unsigned run = 0;
        for (int i = 0; i < SORT_T; ++i) { const unsigned c = cnt[t][i]; cnt[t][i] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_K; ++j) {
        if (i0 + j < per) {
            const unsigned b = sort_bucket(key[j], shift);
            d[offs[((size_t)slice * 16 + b) * tiles + tile] + cnt[b][t] + rank[j]] = key[j];
        }
    }
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
    if (per > 0xffffffffull || nslices < 1) return hipErrorInvalidValue;
    const size_t count = per * (size_t)nslices;
    uint64_t* keys = reinterpret_cast<uint64_t*>(spectrum);
    uint64_t* other = reinterpret_cast<uint64_t*>(sorted);
    lex_keys_kernel<<<4096, 256, 0, st>>>(keys, count);
    const int tiles = (int)((per + SORT_TILE - 1) / SORT_TILE);
    unsigned* hist = nullptr;
    hipError_t e = hipMalloc((void**)&hist, sizeof(unsigned) * 16 * (size_t)tiles * nslices);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)tiles, (unsigned)nslices);
    for (int pass = 0; pass < 16; ++pass) {   // 16 passes: the sorted keys end where they started; a copy puts them into `sorted`
        const uint64_t* src = (pass & 1) ? other : keys;
        uint64_t* dst = (pass & 1) ? keys : other;
        sort_hist_kernel<<<grid, SORT_T, 0, st>>>(src, hist, per, tiles, 4 * pass);
        sort_scan_kernel<<<nslices, 16, 0, st>>>(hist, tiles);
        sort_scatter_kernel<<<grid, SORT_T, 0, st>>>(src, dst, hist, per, tiles, 4 * pass);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(other, keys, sizeof(uint64_t) * count, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) {
        peaks_kernel<<<(nslices + 255) / 256, 256, 0, st>>>(reinterpret_cast<const uint64_t*>(sorted), per, nslices, peaks_dev);
        e = hipGetLastError();
    }
    const hipError_t es = hipStreamSynchronize(st);
    hipFree(hist);
    return e != hipSuccess ? e : es;
}

hipError_t data_driven_pick(const void* sorted, size_t per, int nslices, int niter, const float* bounds_dev, float* tau_dev, long long* count_dev,
                            hipStream_t st)
{
    pick_kernel<<<nslices, 128, 0, st>>>(reinterpret_cast<const uint64_t*>(sorted), per, niter, bounds_dev, tau_dev, count_dev);
    return hipGetLastError();
}

}  // namespace p3d
