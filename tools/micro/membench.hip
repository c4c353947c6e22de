// membench.hip -- what HBM rate do the access patterns of the two POCS passes reach with NO compute?
// (build: hipcc -O3 --offload-arch=gfx950 membench.hip -o membench ; run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int N1 = 1024, N2 = 1024;

// linear in-place scale, V = float2 / float4
template <class V> __global__ void lin(V* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { V v = p[i]; v.x *= 1.0001f; p[i] = v; }
}
// row pattern: one wave per 8 KiB row, 16 x float2 (lane + 64 q) or 8 x float4 (lane + 64 q); read p, (optionally read x), write p
template <int W16, int EXTRA> __global__ __launch_bounds__(256) void rowpat(float2* p, const float2* x) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + wave;
    float2* r = p + row * N2; const float2* xr = x + row * N2;
    if (W16) {
        float4 v[8], w[8];
        #pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = reinterpret_cast<float4*>(r)[lane + 64 * q];
        if (EXTRA) {
            #pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = reinterpret_cast<const float4*>(xr)[lane + 64 * q];
            #pragma unroll
            for (int q = 0; q < 8; ++q) { v[q].x += w[q].x; v[q].z += w[q].w; }
        }
        #pragma unroll
        for (int q = 0; q < 8; ++q) { v[q].x *= 1.0001f; reinterpret_cast<float4*>(r)[lane + 64 * q] = v[q]; }
    } else {
        float2 v[16], w[16];
        #pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = r[lane + 64 * q];
        if (EXTRA) {
            #pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = xr[lane + 64 * q];
            #pragma unroll
            for (int q = 0; q < 16; ++q) v[q].x += w[q].y;
        }
        #pragma unroll
        for (int q = 0; q < 16; ++q) { v[q].x *= 1.0001f; r[lane + 64 * q] = v[q]; }
    }
}
// column-tile pattern, in place.  T columns x 1024 rows per workgroup.
//  MODE 0: float2 per lane, thread (c = tid % T, j = tid / T), rows j + (1024/16... ) 16 loads
//  MODE 1: float4 per lane (two adjacent columns), thread (cp = tid % (T/2), j = tid / (T/2)), 8 loads, rows j + 128 q
template <int T, int MODE, int LDSKB> __global__ __launch_bounds__(MODE ? (T / 2) * 128 : T * 64) void colpat(float2* p) {
    extern __shared__ char dummy[];   // only to limit residency like the real kernel does
    const size_t base = (size_t)blockIdx.y * N1 * N2 + (size_t)blockIdx.x * T;
    if (MODE == 0) {
        const int c = threadIdx.x % T, j = threadIdx.x / T;
        float2* q0 = p + base + c;
        float2 v[16];
        #pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = q0[(size_t)(j + 64 * q) * N2];
        #pragma unroll
        for (int q = 0; q < 16; ++q) { v[q].x *= 1.0001f; q0[(size_t)(j + 64 * q) * N2] = v[q]; }
    } else {
        const int cp = threadIdx.x % (T / 2), j = threadIdx.x / (T / 2);
        float2* q0 = p + base + 2 * cp;
        float4 v[8];
        #pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<float4*>(q0 + (size_t)(j + 128 * q) * N2);
        #pragma unroll
        for (int q = 0; q < 8; ++q) { v[q].x *= 1.0001f; *reinterpret_cast<float4*>(q0 + (size_t)(j + 128 * q) * N2) = v[q]; }
    }
    if (LDSKB < 0) dummy[threadIdx.x] = 0;
}


// (theta) T=8 on the standard layout, half-tiles of one 128-B line pair placed 8 blocks apart (same XCD under round-robin)
template <int PAIRED> __global__ __launch_bounds__(512) void colpat8(float2* p) {
    extern __shared__ char dummy[];
    int b = blockIdx.x;
    if (PAIRED) { const int pair = b >> 4, w = b & 15; b = pair * 16 + (w & 7) * 2 + (w >> 3); }
    const size_t base = (size_t)blockIdx.y * N1 * N2 + (size_t)b * 8;
    const int c = threadIdx.x & 7, j = threadIdx.x >> 3;
    float2* q0 = p + base + c;
    float2 v[16];
    #pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = q0[(size_t)(j + 64 * q) * N2];
    #pragma unroll
    for (int q = 0; q < 16; ++q) { v[q].x *= 1.0001f; q0[(size_t)(j + 64 * q) * N2] = v[q]; }
}
// (iota) column-blocked layout  W[slice][colblock = N2/8][row][8]: the column pass streams contiguous 64 KiB tiles
__global__ __launch_bounds__(512) void colblk(float2* p) {
    extern __shared__ char dummy[];
    float2* q0 = p + ((size_t)blockIdx.y * (N2 / 8) + blockIdx.x) * (size_t)N1 * 8 + threadIdx.x;   // (row j, col c) = tid/8, tid%8
    float2 v[16];
    #pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = q0[(size_t)q * 64 * 8];
    #pragma unroll
    for (int q = 0; q < 16; ++q) { v[q].x *= 1.0001f; q0[(size_t)q * 64 * 8] = v[q]; }
}
// ... and the row pass on that layout: wave = row, element e = lane + 64 q lives at colblock e/8, (row, e%8); 4 adjacent rows per WG
template <int EXTRA> __global__ __launch_bounds__(256) void rowblk(float2* p, const float2* x) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t rowg = (size_t)blockIdx.x * 4 + wave;          // global row index over all slices
    const size_t slice = rowg / N1, row = rowg % N1;
    float2* sb = p + slice * N1 * N2;
    const float2* xr = x + rowg * N2;
    float2 v[16], w[16];
    #pragma unroll
    for (int q = 0; q < 16; ++q) { const int e = lane + 64 * q; v[q] = sb[((size_t)(e >> 3) * N1 + row) * 8 + (e & 7)]; }
    if (EXTRA) {
        #pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = xr[lane + 64 * q];
        #pragma unroll
        for (int q = 0; q < 16; ++q) v[q].x += w[q].y;
    }
    #pragma unroll
    for (int q = 0; q < 16; ++q) { const int e = lane + 64 * q; v[q].x *= 1.0001f; sb[((size_t)(e >> 3) * N1 + row) * 8 + (e & 7)] = v[q]; }
}
// read-only / write-only streams, each WG a contiguous 64 KiB chunk (what the column pass of a sparse spectrum / the row pass's store side look like to the memory)
__global__ void read4(const float4* a, float* sink, size_t n) {
    const size_t base = (size_t)blockIdx.x * 4096;
    float4 v[16];
    #pragma unroll
    for (int k = 0; k < 16; ++k) { const size_t i = base + k * 256 + threadIdx.x; v[k] = i < n ? a[i] : float4{0.f, 0.f, 0.f, 0.f}; }
    float s = 0.f;
    #pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k].x + v[k].w;
    if (s == 123.456f) sink[0] = s;   // (never: keeps the loads)
}
__global__ void write4(float4* b, size_t n, float val) {
    const size_t base = (size_t)blockIdx.x * 4096;
    #pragma unroll
    for (int k = 0; k < 16; ++k) { const size_t i = base + k * 256 + threadIdx.x; if (i < n) b[i] = float4{val, val, val, val}; }
}
__global__ void copy4(const float4* a, float4* b, size_t n) {   // each WG a contiguous 64 KiB chunk
    const size_t base = (size_t)blockIdx.x * 4096;
    #pragma unroll
    for (int k = 0; k < 16; ++k) { const size_t i = base + k * 256 + threadIdx.x; if (i < n) b[i] = a[i]; }
}


// row pattern on the blocked layout with limited residency (dynamic LDS as ballast) and R rows per wave in flight
template <int R> __global__ __launch_bounds__(256) void rowblk_occ(float2* p, const float2* x) {
    extern __shared__ char dummy[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float2 v[R][16], w[R][16];
    #pragma unroll
    for (int r = 0; r < R; ++r) {
        const size_t rowg = ((size_t)blockIdx.x * R + r) * 4 + wave;
        const size_t slice = rowg / N1, row = rowg % N1;
        float2* sb = p + slice * N1 * N2; const float2* xr = x + rowg * N2;
        #pragma unroll
        for (int q = 0; q < 16; ++q) { const int e = lane + 64 * q; v[r][q] = sb[((size_t)(e >> 3) * N1 + row) * 8 + (e & 7)]; }
        #pragma unroll
        for (int q = 0; q < 16; ++q) w[r][q] = xr[lane + 64 * q];
    }
    #pragma unroll
    for (int r = 0; r < R; ++r) {
        const size_t rowg = ((size_t)blockIdx.x * R + r) * 4 + wave;
        const size_t slice = rowg / N1, row = rowg % N1;
        float2* sb = p + slice * N1 * N2;
        #pragma unroll
        for (int q = 0; q < 16; ++q) { const int e = lane + 64 * q; v[r][q].x = v[r][q].x * 1.0001f + w[r][q].y; sb[((size_t)(e >> 3) * N1 + row) * 8 + (e & 7)] = v[r][q]; }
    }
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 512;
    const size_t n = (size_t)S * N1 * N2;
    float2 *p, *x;
    CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&x, n * 8));
    CK(hipMemset(p, 0, n * 8)); CK(hipMemset(x, 0, n * 8));
    const double gb = n * 8 / 1e9;
    auto rep = [&](const char* name, float ms, double bytes_gb) { printf("%-58s %8.3f ms  %7.1f GB/s\n", name, ms, bytes_gb / (ms * 1e-3)); };
    rep("linear float4 r+w (grid 2048x256)", timeit([&] { lin<float4><<<2048, 256>>>((float4*)p, n / 2); }, 5), 2 * gb);
    rep("linear float2 r+w (grid 2048x256)", timeit([&] { lin<float2><<<2048, 256>>>(p, n); }, 5), 2 * gb);
    rep("linear float4 r+w (grid 16384x256)", timeit([&] { lin<float4><<<16384, 256>>>((float4*)p, n / 2); }, 5), 2 * gb);
    const int rowblocks = S * N1 / 4;
    rep("row 8B/lane   r+w", timeit([&] { rowpat<0, 0><<<rowblocks, 256>>>(p, x); }, 5), 2 * gb);
    rep("row 16B/lane  r+w", timeit([&] { rowpat<1, 0><<<rowblocks, 256>>>(p, x); }, 5), 2 * gb);
    rep("row 8B/lane   r+r+w", timeit([&] { rowpat<0, 1><<<rowblocks, 256>>>(p, x); }, 5), 3 * gb);
    rep("row 16B/lane  r+r+w", timeit([&] { rowpat<1, 1><<<rowblocks, 256>>>(p, x); }, 5), 3 * gb);
    auto col = [&](auto kern, int T, int threads, int ldskb, const char* name) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldskb * 1024));
        rep(name, timeit([&] { kern<<<dim3(N2 / T, S), threads, ldskb * 1024>>>(p); }, 5), 2 * gb);
    };
    col(colpat<16, 0, 1>, 16, 1024, 144, "col T=16 8B/lane  1 WG/CU (144 KiB LDS)");
    col(colpat<16, 1, 1>, 16, 1024, 144, "col T=16 16B/lane 1 WG/CU (144 KiB LDS)");
    col(colpat<16, 0, 1>, 16, 1024, 72, "col T=16 8B/lane  2 WG/CU (72 KiB LDS)");
    col(colpat<16, 1, 1>, 16, 1024, 72, "col T=16 16B/lane 2 WG/CU (72 KiB LDS)");
    col(colpat<8, 0, 1>, 8, 512, 72, "col T=8  8B/lane  2 WG/CU (72 KiB LDS)  64-B segments");
    col(colpat<8, 1, 1>, 8, 512, 72, "col T=8  16B/lane 2 WG/CU (72 KiB LDS)  64-B segments");
    col(colpat<8, 0, 1>, 8, 512, 36, "col T=8  8B/lane  4 WG/CU (36 KiB LDS)  64-B segments");
    col(colpat<16, 0, 1>, 16, 1024, 1, "col T=16 8B/lane  no LDS limit");
    col(colpat<16, 1, 1>, 16, 1024, 1, "col T=16 16B/lane no LDS limit");
    rep("read only float4, WG-contiguous 64 KiB", timeit([&] { read4<<<(unsigned)((n / 2 + 4095) / 4096), 256>>>((const float4*)x, (float*)p, n / 2); }, 5), gb);
    rep("write only float4, WG-contiguous 64 KiB", timeit([&] { write4<<<(unsigned)((n / 2 + 4095) / 4096), 256>>>((float4*)p, n / 2, 1.5f); }, 5), gb);
    rep("copy float4 a->b, WG-contiguous 64 KiB (r+w)", timeit([&] { copy4<<<(unsigned)((n / 2 + 4095) / 4096), 256>>>((const float4*)x, (float4*)p, n / 2); }, 5), 2 * gb);
    auto col8 = [&](auto kern, int ldskb, const char* name) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldskb * 1024));
        rep(name, timeit([&] { kern<<<dim3(N2 / 8, S), 512, ldskb * 1024>>>(p); }, 5), 2 * gb);
    };
    col8(colpat8<0>, 72, "col T=8 std layout, unpaired, 2 WG/CU");
    col8(colpat8<1>, 72, "col T=8 std layout, XCD-paired halves, 2 WG/CU");
    col8(colblk, 72, "col T=8 BLOCKED layout (contiguous tile), 2 WG/CU");
    col8(colblk, 36, "col T=8 BLOCKED layout (contiguous tile), 4 WG/CU");
    rep("row on BLOCKED layout 8B/lane r+w", timeit([&] { rowblk<0><<<rowblocks, 256>>>(p, x); }, 5), 2 * gb);
    rep("row on BLOCKED layout 8B/lane r+r+w", timeit([&] { rowblk<1><<<rowblocks, 256>>>(p, x); }, 5), 3 * gb);
    auto rocc = [&](auto kern, int R, int ldskb, const char* name) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldskb * 1024));
        rep(name, timeit([&] { kern<<<rowblocks / R, 256, ldskb * 1024>>>(p, x); }, 5), 3 * gb);
    };
    rocc(rowblk_occ<1>, 1, 1, "rowblk r+r+w 1 row/wave, no LDS limit");
    rocc(rowblk_occ<1>, 1, 20, "rowblk r+r+w 1 row/wave, 8 WG/CU (32 waves)");
    rocc(rowblk_occ<1>, 1, 32, "rowblk r+r+w 1 row/wave, 5 WG/CU (20 waves)");
    rocc(rowblk_occ<1>, 1, 40, "rowblk r+r+w 1 row/wave, 4 WG/CU (16 waves)");
    rocc(rowblk_occ<1>, 1, 52, "rowblk r+r+w 1 row/wave, 3 WG/CU (12 waves)");
    rocc(rowblk_occ<1>, 1, 80, "rowblk r+r+w 1 row/wave, 2 WG/CU (8 waves)");
    rocc(rowblk_occ<1>, 1, 150, "rowblk r+r+w 1 row/wave, 1 WG/CU (4 waves)");
    rocc(rowblk_occ<2>, 2, 52, "rowblk r+r+w 2 rows/wave, 3 WG/CU (12 waves)");
    rocc(rowblk_occ<2>, 2, 80, "rowblk r+r+w 2 rows/wave, 2 WG/CU (8 waves)");
    return 0;
}
