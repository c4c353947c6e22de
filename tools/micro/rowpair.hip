// rowpair.hip -- go / no-go measurement for VERDICT r02 item 5: what does a 1024-point inverse + forward transform pair of one wavefront cost
//   (A) as the row pass runs it today: register-resident Stockham passes 4-16-16 / 16-16-4 of p3d_fft.hpp, four exchanges through LDS per row,
//   (B) with NO LDS: one in-register radix-16 pass per transform and the lane dimension (64 = 2^6) as six stages of cross-lane butterflies
//       (the "no autosort, permuted row-frequency order" variant: the column pass would not mind the order).
// (B) is a COST model, not a transform: every stage does what a real one would have to -- two lane-crossing moves per complex register (DPP
// moves 4 bytes per lane; packed VOP3P has no DPP), the butterfly against a per-lane sign, one per-lane complex twiddle -- with dummy lane
// patterns and twiddles, so its results mean nothing and only its time does.  Both run in the configuration of row_pipe64_kernel<1024>:
// 16 wavefronts (rows) per workgroup, one workgroup per CU, no global memory traffic inside the loop.
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I../../pseudo-3d-interpolation_amd/csrc rowpair.hip -o rowpair
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "p3d_fft.hpp"
using namespace p3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int N = 1024, ROWS = 16, LSTR = LdsRow::stride(N);

template <int CTRL>
__device__ __forceinline__ float xlane(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }

// one cross-lane radix-2 stage on 16 complex registers: partner through DPP, x' = x * s + partner (s = +-1 per lane), then a per-lane twiddle
template <int CTRL>
__device__ __forceinline__ void lane_stage(c32 (&v)[16], float sgn, c32 w)
{
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float pr = xlane<CTRL>(v[q].x), pi = xlane<CTRL>(v[q].y);
        c32 b{__builtin_fmaf(v[q].x, sgn, pr), __builtin_fmaf(v[q].y, sgn, pi)};
        v[q] = b * w;
    }
}

template <int VARIANT>
__global__ __launch_bounds__(1024, 4) void pair_kernel(const c32* tw_g, c32* out, int rows_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* twl = reinterpret_cast<c32*>(smem);
    c32* data = twl + PassTables<N>::slots();
    const int tid = threadIdx.x, line = tid >> 6, tl = tid & 63;
    for (int i = tid; i < PassTables<N>::slots(); i += 1024) twl[i] = tw_g[i];
    __syncthreads();
    const TwOrdered tw{twl};
    const LdsRow lds{data + line * LSTR};
    c32 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = c32{(float)(tl + 64 * q) * 1e-3f, (float)(line + q) * 1e-3f};
    const float sgn = (tl & 1) ? -1.0f : 1.0f;
    const c32 w = twl[1 + (tl & 15)];
    for (int r = 0; r < rows_per_wave; ++r) {
        if (VARIANT == 0) {
            line_fft<N, INV, true>(v, lds, tw, tl);
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = v[q] * (1.0f / 1024.0f);
            line_fft<N, FWD, true>(v, lds, tw, tl);
        } else {
            // inverse: six lane stages, then the in-register radix-16; forward: radix-16, then six lane stages
            lane_stage<0xB1>(v, sgn, w);    // quad_perm [1,0,3,2]  (lane ^ 1)
            lane_stage<0x4E>(v, sgn, w);    // quad_perm [2,3,0,1]  (lane ^ 2)
            lane_stage<0x141>(v, sgn, w);   // row_half_mirror      (stands for lane ^ 4)
            lane_stage<0x140>(v, sgn, w);   // row_mirror           (stands for lane ^ 8)
            lane_stage<0x128>(v, sgn, w);   // row_ror:8            (stands for the 16-lane step; permlane16_swap on gfx950)
            lane_stage<0x124>(v, sgn, w);   // row_ror:4            (stands for the 32-lane step; permlane32_swap on gfx950)
            pass_compute<N, FWD, 0>(v, tw, tl);   // in-register radix 16 of the inverse transform (the forward butterfly stands in: same cost)
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = v[q] * (1.0f / 1024.0f);
            pass_compute<N, FWD, 0>(v, tw, tl);   // in-register radix 16 of the forward transform
            lane_stage<0xB1>(v, sgn, w);
            lane_stage<0x4E>(v, sgn, w);
            lane_stage<0x141>(v, sgn, w);
            lane_stage<0x140>(v, sgn, w);
            lane_stage<0x128>(v, sgn, w);
            lane_stage<0x124>(v, sgn, w);
        }
    }
    c32 acc{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = acc + v[q];
    out[(size_t)blockIdx.x * 1024 + tid] = acc;
}

// (C) 1024 = 32 x 32: 32 points per thread, a row in HALF a wavefront (two rows per wavefront, 8 wavefronts per CU for the same 16 rows in flight), ONE
// exchange through LDS per transform instead of two.  A real transform (Stockham, pass 1 without twiddles, pass 2 with the ordered table w^(t j)), results in
// the canonical layout (register k of lane j = element j + 32 k).  The exchange: lane j stores its 32 outputs contiguously (16 x 128 bits, two padding slots per
// 32 elements keep the stores 16-byte aligned and spread the lanes over the banks), then reads element j + 32 t with lane stride 1.
template <int DIR>
__device__ __forceinline__ void dft32(c32 (&x)[32])
{
    constexpr float C[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                             0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
                             -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f, 0.83146961230254523708f,
                             0.92387953251128675613f, 0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
    c32 e[16], o[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { e[m] = x[2 * m]; o[m] = x[2 * m + 1]; }
    Dft<16, DIR>::run(e);
    Dft<16, DIR>::run(o);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const c32 ek = e[digit_rev<16>(k)];
        c32 ok = o[digit_rev<16>(k)];
        if (k == 8) ok = mul_i<DIR>(ok);
        else if (k != 0) ok = ok * c32{C[k], DIR * S[k]};
        x[k] = ek + ok;
        x[k + 16] = ek - ok;
    }
}
constexpr int LSTR32 = 1024 + 2 * 32;   // two padding slots per 32 elements
template <int DIR>
__device__ __forceinline__ void fft1024_32x32(c32 (&v)[32], c32* row, const c32* tw2, int j)
{
    dft32<DIR>(v);                                  // pass 1 (no twiddles): X[k] of butterfly j -> position 32 j + k
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        float4* p = reinterpret_cast<float4*>(row + 34 * j);
#pragma unroll
        for (int m = 0; m < 16; ++m) p[m] = float4{v[2 * m].x, v[2 * m].y, v[2 * m + 1].x, v[2 * m + 1].y};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v[0] = row[j];
#pragma unroll
    for (int t = 1; t < 32; ++t) v[t] = row[j + 34 * t] * tw2[(t - 1) * 32 + j];   // pass 2: in[j + 32 t] * w^(t j)
    dft32<DIR>(v);                                  // X[k] = element j + 32 k: canonical again
}
__global__ __launch_bounds__(512, 2) void pair32_kernel(const c32* tw_g, c32* out, int trips)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* twl = reinterpret_cast<c32*>(smem);            // [2][31][32]: forward, inverse
    c32* data = twl + 2 * 31 * 32;
    const int tid = threadIdx.x, rowi = tid >> 5, j = tid & 31;
    for (int i = tid; i < 2 * 31 * 32; i += 512) twl[i] = tw_g[i];
    __syncthreads();
    c32* row = data + rowi * LSTR32;
    c32 v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = c32{(float)(j + 32 * q) * 1e-3f, (float)(rowi + q) * 1e-3f};
    for (int r = 0; r < trips; ++r) {
        fft1024_32x32<INV>(v, row, twl + 31 * 32, j);
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = v[q] * (1.0f / 1024.0f);
        fft1024_32x32<FWD>(v, row, twl, j);
    }
    c32 acc{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 32; ++q) acc = acc + v[q];
    out[(size_t)blockIdx.x * 1024 + tid] = acc;
    if (trips == 1 && blockIdx.x == 0 && rowi == 0) {   // self-check: one round trip reproduces the input (stored behind the sums)
#pragma unroll
        for (int q = 0; q < 32; ++q) out[(size_t)gridDim.x * 1024 + j + 32 * q] = v[q];
    }
}

int main()
{
    std::vector<c32> host(PassTables<N>::slots());
    PassTables<N>::build(host.data());
    c32 *tw, *out;
    int cus = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    CK(hipMalloc(&tw, sizeof(c32) * host.size()));
    CK(hipMemcpy(tw, host.data(), sizeof(c32) * host.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&out, sizeof(c32) * 1024 * (size_t)cus));
    const size_t lds = sizeof(c32) * (PassTables<N>::slots() + (size_t)ROWS * LSTR);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int rows_per_wave = 128;   // the row pass of the headline cube: 2048 rows per CU = 128 per wavefront
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (variant == 0) pair_kernel<0><<<cus, 1024, lds>>>(tw, out, rows_per_wave);
            else pair_kernel<1><<<cus, 1024, lds>>>(tw, out, rows_per_wave);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s: %.3f ms for %d rows per CU (%d per wavefront, 16 wavefronts per CU) = %.3f us per row and CU\n",
               variant == 0 ? "A  LDS Stockham passes (today)            " : "B  cross-lane stages, no LDS (cost model)  ", best, rows_per_wave * ROWS, rows_per_wave,
               best * 1e3f / (rows_per_wave * ROWS));
    }
    {   // (C)
        std::vector<c32> t2(2 * 31 * 32);
        for (int d = 0; d < 2; ++d)
            for (int t = 1; t < 32; ++t)
                for (int j = 0; j < 32; ++j) {
                    const double ang = (d == 0 ? -1.0 : 1.0) * 6.283185307179586476925286766559 * t * j / 1024.0;
                    t2[(size_t)d * 31 * 32 + (t - 1) * 32 + j] = c32{(float)cos(ang), (float)sin(ang)};
                }
        c32 *tw2, *out2;
        CK(hipMalloc(&tw2, sizeof(c32) * t2.size()));
        CK(hipMemcpy(tw2, t2.data(), sizeof(c32) * t2.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&out2, sizeof(c32) * 1024 * ((size_t)cus + 1)));
        const size_t lds2 = sizeof(c32) * (2 * 31 * 32 + (size_t)ROWS * LSTR32);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        pair32_kernel<<<cus, 512, lds2>>>(tw2, out2, 1);
        std::vector<c32> back(1024);
        CK(hipMemcpy(back.data(), out2 + (size_t)cus * 1024, sizeof(c32) * 1024, hipMemcpyDeviceToHost));
        double err = 0.0;
        for (int e = 0; e < 1024; ++e) {   // row 0: input element e = (e * 1e-3, (e / 32) * 1e-3), one inverse + scale + forward round trip
            const double dx = back[e].x - e * 1e-3, dy = back[e].y - (e / 32) * 1e-3;
            err = fmax(err, sqrt(dx * dx + dy * dy));
        }
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            pair32_kernel<<<cus, 512, lds2>>>(tw2, out2, rows_per_wave);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("C  32 x 32, one exchange, 32 points per thread : %.3f ms for %d rows per CU (two rows per wavefront, 8 wavefronts per CU) = %.3f us per row and CU; round-trip error %.1e\n",
               best, rows_per_wave * ROWS, best * 1e3f / (rows_per_wave * ROWS), err);
    }
    return 0;
}
