// p3d_resident.hip -- the whole POCS job of a SMALL slice in one kernel: one workgroup per slice, the slice in registers, every
// transform through LDS, all K iterations without touching HBM.
//
// The reference runs POCS_algorithm once per (iline, xline) slice (pseudo_3D_interpolation/cube_POCS_interpolation_3D.py:314-340,
// functions/POCS.py:549-632); for slices of 32 x 32 ... 128 x 128 points (BASELINE configs[0]: 64 x 64 x 128) the two-pass kernels
// of p3d_kernels.hpp are launch bound (two launches + bookkeeping per iteration for 32 KiB of data per slice).  A slice of
// N1 x N2 <= 16384 points is 16 points per thread of an N1 N2 / 16-thread workgroup, so here
//
//     x_obs, mask bits -> registers (once)                                   POCS.py:549
//     K x { row FFT -> [LDS] -> column FFT -> threshold -> inverse column FFT -> [LDS] -> inverse row FFT
//           -> x = x (1 - alpha mask) + alpha x_obs -> sum |x| -> cost -> early exit }          POCS.py:560-632
//     result -> HBM (once)
//
// HBM traffic per slice: (8 + 8) B/point per JOB instead of per iteration.  Same templates (line_fft, Shrink), same twiddle tables,
// same order of operations and the same reduction order of the cost sums as the two-pass path: results, sums and iteration counts
// are bit-identical to it (tests/test_gpu_parity.py::test_resident_small_slices_equal_the_two_pass_path).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "p3d_fft.hpp"
#include "p3d_resident.hpp"
#include "p3d_shrink.hpp"

namespace p3d {

// |x| for the cost sums, as in p3d_kernels.hpp
__device__ __forceinline__ float res_abs(c32 v) { return __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y); }

// column view of the slice buffer: rows are LdsRow-padded lines of pitch P; `base` points at (row 0, this thread's column)
template <int P>
struct LdsColPitch {
    c32* base;
    P3D_HD c32& at(int pos) const { return base[pos * P]; }
    P3D_HD c32* ptr(int pos) const { return base + pos * P; }
    static constexpr int rel(int c) { return c * P; }
};

template <int N1, int N2>
constexpr size_t resident_lds_bytes()
{
    return sizeof(c32) * (PassTables<N2>::slots() + ColTables<N1>::slots() + (size_t)N1 * LdsRow::stride(N2)) + sizeof(double) * (N1 + 8);
}

template <int N1, int N2, int OP>
__global__ __launch_bounds__(N1* N2 / 16) void resident_kernel(const ResidentArgs a)
{
    constexpr int T = N1 * N2 / 16, TPL2 = N2 / 16, TPL1 = N1 / 16, PPT = 16;
    constexpr int P = LdsRow::stride(N2);
    static_assert(Plan<N1>::PPT == 16 && Plan<N2>::PPT == 16 && T <= 1024 && T >= 64, "16 points per thread, one workgroup per slice");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twr = reinterpret_cast<c32*>(smem_raw);
    c32* twc = twr + PassTables<N2>::slots();
    c32* buf = twc + ColTables<N1>::slots();
    double* rows = reinterpret_cast<double*>(buf + (size_t)N1 * P);   // per-row sums of |x|, then [N1] the slice's sum, [N1 + 1] the cost

    const int tid = threadIdx.x;
    const int slice = blockIdx.x;
    const int r = tid / TPL2, tl = tid % TPL2;      // row phase: thread tl of row r holds columns tl + TPL2 q
    const int c = tid % N2, tl1 = tid / N2;         // column phase: thread tl1 of column c holds rows tl1 + TPL1 q
    for (int i = tid; i < PassTables<N2>::slots(); i += T) twr[i] = a.tw_row[i];
    for (int i = tid; i < ColTables<N1>::slots(); i += T) twc[i] = a.tw_col[i];
    const TwOrdered tw_r{twr};
    const TwCol tw_c{twc};
    const LdsColPitch<P> lcol{buf + c + (c >> 4)};

    const int state = a.done[slice];   // < 0: all-zero slice, handed back untouched (POCS.py:515-521)
    const size_t base = ((size_t)slice * N1 + r) * N2 + tl;
    if (state < 0) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[base + TPL2 * q] = c32{0.f, 0.f};
            else reinterpret_cast<float*>(a.out)[base + TPL2 * q] = 0.f;
        }
        return;
    }
    __syncthreads();

    // slice sum of |x| in the order of the two-pass path: 16 terms per thread in float, the threads of a row by a shuffle tree in
    // double (row_kernel), the rows by the pairwise tree of reduce_rows_kernel; the total ends up in rows[N1]
    auto slice_sum = [&](float acc) {
        double ws = (double)acc;
#pragma unroll
        for (int o = TPL2 / 2; o > 0; o >>= 1) ws += __shfl_down(ws, o, TPL2);
        if (tl == 0) rows[r] = ws;
        __syncthreads();
        if (tid < 64) {
            double t = 0.0;
            if constexpr (N1 > 64) t = rows[tid] + rows[tid + 64];
            else if (tid < N1) t = rows[tid];
            constexpr int W0 = N1 >= 64 ? 32 : N1 / 2;
#pragma unroll
            for (int o = W0; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
            if (tid == 0) rows[N1] = t;
        }
        __syncthreads();
        return rows[N1];
    };

    // observed samples and mask bits stay in registers for the whole job -- except in the 1024-thread workgroup of a 128 x 128
    // slice, whose threads have 128 registers each: there the samples are read again every iteration (the slice's 128 KiB stay
    // in L2)
    constexpr bool KEEP = T < 1024;
    auto load_obs = [&](int q) -> c32 {
        if (a.dtype == 0) return reinterpret_cast<const c32*>(a.x)[base + TPL2 * q];
        return c32{reinterpret_cast<const float*>(a.x)[base + TPL2 * q], 0.f};
    };
    c32 xo[KEEP ? PPT : 1], v[PPT];
    const unsigned mbits = a.bits[(size_t)r * TPL2 + tl];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const c32 x = load_obs(q);
        if (KEEP) xo[q] = x;
        acc += res_abs(x);
        v[q] = x;
    }
    double prev = slice_sum(acc);
    if (tid == 0) a.sums[slice] = prev;

    int done_at = 0;
    for (int k = 0; k < a.niter; ++k) {
        // (the transforms' LDS and twiddle addresses are functions of the thread index alone; hoisted out of this loop the four
        // sets cost ~40 registers -- and the 1024-thread workgroup of a 128 x 128 slice has 128 per thread.  Recomputed instead.)
        int tl_k = tl, tl1_k = tl1;
        asm volatile("" : "+v"(tl_k), "+v"(tl1_k));
        const LdsRow lrow{buf + (tid / TPL2) * P};
        // ---- fft2 ----
        line_fft<N2, FWD, true>(v, lrow, tw_r, tl_k);
#pragma unroll
        for (int q = 0; q < PPT; ++q) lrow.at(tl_k + TPL2 * q) = v[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = lcol.at(tl1_k + TPL1 * q);
        line_fft<N1, FWD, false>(v, lcol, tw_c, tl1_k);
        // ---- threshold (threshold_operator.py:9-112) ----
        {
            const Shrink shr(a.tau[(size_t)slice * a.niter + k], OP);
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = shr(v[q]);
        }
        // ---- ifft2 ----
        line_fft<N1, INV, false>(v, lcol, tw_c, tl1_k);
        __syncthreads();   // every thread is done gathering from the buffer
#pragma unroll
        for (int q = 0; q < PPT; ++q) lcol.at(tl1_k + TPL1 * q) = v[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = lrow.at(tl_k + TPL2 * q);
        line_fft<N2, INV, true>(v, lrow, tw_r, tl_k);
        // ---- re-insertion (POCS.py:616-619) and cost (POCS.py:622) ----
        acc = 0.f;
        __builtin_amdgcn_sched_barrier(0);   // (keeps the re-read samples of the 1024-thread variant out of the transform's registers)
#pragma unroll
        for (int g = 0; g < PPT; g += 4) {
            c32 xg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xg[i] = KEEP ? xo[g + i] : load_obs(g + i);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = g + i;
                c32 xn = v[q] * a.scale;
                const float m = (float)((mbits >> q) & 1u);
                const float w = 1.0f - a.alpha * m;
                xn = axpby(xn, w, xg[i], a.alpha);
                acc += res_abs(xn);
                v[q] = xn;
            }
            if (!KEEP) __builtin_amdgcn_sched_barrier(0);
        }
        const double cur = slice_sum(acc);
        if (tid == 0) a.sums[(size_t)(k + 1) * a.nslices + slice] = cur;
        const double d = cur - prev;
        const double cost = (d * d) / (cur * cur);
        prev = cur;
        if (a.eps > 0.0 && k > 2 && cost < a.eps) {   // uniform over the workgroup (POCS.py:631)
            done_at = k + 1;
            break;
        }
    }
    if (tid == 0) a.done[slice] = done_at;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[base + TPL2 * q] = v[q];
        else reinterpret_cast<float*>(a.out)[base + TPL2 * q] = v[q].x;   // np.real(), POCS.py:656
    }
}

template <int N1, int N2>
static hipError_t launch_shape(const ResidentArgs& a, hipStream_t st)
{
    constexpr size_t lds = resident_lds_bytes<N1, N2>();
    constexpr int T = N1 * N2 / 16;
#define P3D_RES(OP)                                                                                                         \
    do {                                                                                                                    \
        if (lds > 64 * 1024) {                                                                                              \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(resident_kernel<N1, N2, OP>),            \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e != hipSuccess) return e;                                                                                  \
        }                                                                                                                   \
        resident_kernel<N1, N2, OP><<<a.nslices, T, lds, st>>>(a);                                                          \
    } while (0)
    switch (a.op) {
        case 0: P3D_RES(0); break;
        case 1: P3D_RES(1); break;
        case 2: P3D_RES(2); break;
        default: return hipErrorInvalidValue;
    }
#undef P3D_RES
    return hipGetLastError();
}

bool resident_supported(int nil, int nxl)
{
    auto ok = [](int n) { return n == 32 || n == 64 || n == 128; };
    return ok(nil) && ok(nxl);
}

hipError_t resident_launch(int nil, int nxl, const ResidentArgs& a, hipStream_t st)
{
#define P3D_SHAPE(A, B) if (nil == A && nxl == B) return launch_shape<A, B>(a, st)
    P3D_SHAPE(32, 32); P3D_SHAPE(32, 64); P3D_SHAPE(32, 128);
    P3D_SHAPE(64, 32); P3D_SHAPE(64, 64); P3D_SHAPE(64, 128);
    P3D_SHAPE(128, 32); P3D_SHAPE(128, 64); P3D_SHAPE(128, 128);
#undef P3D_SHAPE
    return hipErrorNotSupported;
}

}  // namespace p3d
