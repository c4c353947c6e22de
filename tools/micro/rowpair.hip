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
    return 0;
}
