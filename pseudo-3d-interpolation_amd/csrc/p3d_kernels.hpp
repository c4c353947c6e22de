// p3d_kernels.hpp -- the two fused passes of one POCS iteration on a batch of slices.
//
// One iteration of the reference loop (pseudo_3D_interpolation/functions/POCS.py:560-632)
//     X = fft2(x_old); X = threshold(X, tau_k); x = ifft2(X); x *= 1 - alpha*mask; x += alpha*x_obs
// is regrouped so that every slice is read and written exactly twice per iteration:
//
//   spectrum (column) pass  col_kernel<N1,T,COL_ITER>:
//        load T columns of the row-transformed slice -> forward column FFT
//        -> threshold (threshold_operator.py:9-112)  -> inverse column FFT -> store
//   space (row) pass        row_kernel<N2,ROW_MID>:
//        load rows -> inverse row FFT, 1/(N1*N2) -> re-insertion of the observed traces
//        (POCS.py:616-619) -> sum|x| for the cost (POCS.py:622) -> [APOCS input mix, POCS.py:574-575]
//        -> forward row FFT of the NEXT iteration -> store
//
// ROW_FIRST starts the chain (x_obs -> forward row FFT), ROW_LAST ends it (stores x instead of
// transforming again).
//
// Work-buffer layout ("column blocked"): the intermediate between the two passes is private to this
// library, so it is stored as  W[slice][cb = col/8][row][col%8]  (complex64).  One column block
// (8 columns = 64 bytes per row) of a slice is then ONE contiguous N1*64-byte run: the column pass
// streams it with perfectly linear addresses and needs only N1*64 B of LDS per tile (2 workgroups
// per CU at N1 = 1024, so loads of one tile overlap the transforms of another), while the row pass,
// whose workgroups own adjacent rows, still touches 64-byte pieces that are neighbours in memory.
// Measured on MI355X (tools/micro/membench.hip): 4.9-5.2 TB/s for the blocked column tiles vs 4.7
// (16-column tiles, 1 WG/CU) and 3.0 TB/s (8-column tiles) on the row-major layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "p3d_fft.hpp"
#include "p3d_shrink.hpp"

// build-time knobs for experiments (tools/build_variant.sh)
#ifndef P3D_ROW_THREADS
#define P3D_ROW_THREADS 256
#endif
#ifndef P3D_WAVES_PER_EU
#define P3D_WAVES_PER_EU 4  // register budget of the column pass: 512/4 = 128 VGPRs -> 16 waves per CU
#endif
#ifndef P3D_XO_GROUP
#define P3D_XO_GROUP 4   // observed samples fetched per step of the re-insertion loop (late mode)
#endif
#ifndef P3D_XO_EARLY
#define P3D_XO_EARLY 0   // 1: prefetch all observed samples ahead of the inverse transform (32 VGPRs)
#endif
#ifndef P3D_XCD_PAIR
#define P3D_XCD_PAIR 1   // narrow column tiles of one 64-byte block on the same XCD, back to back
#endif
// persistent row pass, waves per SIMD: with the full-cube observed samples v[], bx[], by[] are live across both
// transforms (~220 VGPRs -> 2); with compact samples, fetched after the inverse transform, 157 VGPRs -> 3
#ifndef P3D_PIPE_WAVES_PER_EU
#define P3D_PIPE_WAVES_PER_EU 2
#endif
#ifndef P3D_PIPE_WAVES_PER_EU_COMPACT
#define P3D_PIPE_WAVES_PER_EU_COMPACT 3
#endif
#ifndef P3D_COMPACT_LATE
#define P3D_COMPACT_LATE 1  // 1: fetch the compact observed samples after the inverse transform (frees 32 VGPRs across it)
#endif
#ifndef P3D_PIPE_LOCKSTEP
#define P3D_PIPE_LOCKSTEP 1
#endif
#ifndef P3D_PIPE_PREFETCH
#define P3D_PIPE_PREFETCH 1  // 0 (experiment): no row-ahead prefetch of the work buffer (32 VGPRs less)
#endif
#ifndef P3D_ROW_WAVES_PER_EU
#define P3D_ROW_WAVES_PER_EU (P3D_ROW_THREADS >= 512 ? 4 : 3)
#endif

namespace p3d {

// ROW_SPREAD_INV / COL_SHRINK / ROW_GATHER_FWD: the three passes of one SHEARLET iteration (p3d_shearlet.hip).  The sum over
// the shearlets sits in the row pass because that one has the registers for 16 more accumulators (80 vs 106 VGPRs).
enum RowMode { ROW_FIRST = 0, ROW_MID = 1, ROW_LAST = 2, ROW_SPREAD_INV = 3, ROW_GATHER_FWD = 4 };
enum ColMode { COL_ITER = 0, COL_STATS = 1, COL_FWD = 2, COL_INV = 3, COL_ITER_SOFT = 4, COL_ITER_GARROTE = 5,  // COL_ITER = hard
               COL_SHRINK = 6 };

// Shearlet frame: batch entry b*nsh + s of the work buffer holds shearlet s of slice b.  psi: real spectra [nsh][n1][N]
// (row-major, FFT order); tau: [nb][niter][nsh].
struct ShearArgs {
    const float* psi;
    const c32* tau;
    int nsh, niter, iter, op, real_only;
};

#ifndef P3D_ROW1024_MAXMODE
#define P3D_ROW1024_MAXMODE 2
#endif
constexpr int ROW_THREADS = P3D_ROW_THREADS;
// The one-launch-per-iteration row pass copies its twiddle tables (2N entries) into LDS once per workgroup: 16 KiB per four
// 8-KiB rows at N = 1024.  Sixteen rows per workgroup (one 155-KiB workgroup per CU) take 1.8 ms off the first + last pass of a
// job on the headline cube.  The SHEARLET modes keep their measured configuration.
template <int N, int MODE>
constexpr int row_threads() { return (N == 1024 && MODE <= P3D_ROW1024_MAXMODE && P3D_ROW_THREADS < 1024) ? 1024 : P3D_ROW_THREADS; }
constexpr int STATS_PARTIAL = 8;  // floats per (slice, tile) written by COL_STATS

// ---- column-blocked work layout ------------------------------------------------------------------
__host__ __device__ inline size_t wk_slice_stride(int n1, int n2) { return (size_t)((n2 + 7) / 8) * 8 * n1; }
__host__ __device__ inline size_t wk_off(int row, int col, int n1) { return ((size_t)(col >> 3) * n1 + row) * 8 + (col & 7); }


// Work-buffer element (row, e = tl + TPL*q) of a slice whose base is `ws`: written as a wave-uniform pointer
// (ws + q*qstride, scalar registers) plus ONE per-lane 32-bit offset shared by all q, so that 16 accesses do
// not pin 16 offsets (or 16 64-bit addresses) in vector registers.
template <int TPL>
__device__ __forceinline__ unsigned wk_lane_off(int tl, int row, unsigned wblk)
{
    if constexpr (TPL % 8 == 0) return (unsigned)(tl >> 3) * wblk + (unsigned)row * 8 + (tl & 7);
    else return (unsigned)row * 8;  // short lines: the q-dependent part carries everything
}
template <int TPL, class P>
__device__ __forceinline__ P wk_q_ptr(P ws, int q, int tl, unsigned wblk)
{
    if constexpr (TPL % 8 == 0) return ws + (size_t)q * (TPL / 8) * wblk;
    else { const int e = tl + TPL * q; return ws + (size_t)(e >> 3) * wblk + (e & 7); }
}

// Observed samples in compact form.  For a line (TPL lanes of one wave) and register q, the lanes whose mask
// bit q is set hold consecutive observed positions (element e = tl + TPL*q grows with tl), so their samples
// are consecutive in the compact array: index = rowbase + (observed positions with smaller q) + (rank of
// the lane among the set lanes of its line).  Everything comes from wave ballots of the mask word.
template <int TPL>
struct CompactIndex {
    unsigned long long line_mask;  // lanes of this thread's line
    unsigned long long below;      // lanes of the line below this lane
    unsigned running;              // observed positions of the line in registers < q
    __device__ __forceinline__ CompactIndex(int lane, unsigned base)
    {
        const int first = lane & ~(TPL - 1);
        line_mask = TPL == 64 ? ~0ull : (((1ull << TPL) - 1ull) << first);
        below = line_mask & ((1ull << lane) - 1ull);
        running = base;
    }
    // index of this lane's sample for register q (valid when `set`), then advance to q+1
    __device__ __forceinline__ unsigned next(bool set)
    {
        const unsigned long long b = __ballot(set);
        const unsigned idx = running + (unsigned)__popcll(b & below);
        running += (unsigned)__popcll(b & line_mask);
        return idx;
    }
};

// lane-mask tables of the wave-uniform persistent row pass (row_pipe64_kernel): one 64-bit word per (row or slice, wavefront of
// the row, register q)
__host__ __device__ constexpr size_t pipe64_word(size_t row_or_slice, int wpl, int wsub, int q) { return (row_or_slice * wpl + wsub) * 16 + q; }

// experiment switches a plan reads from the environment when it is created and hands to the launchers (RowArgs / ColArgs::host_sw)
enum { P3D_SW_FLEX_NO_PERSIST = 1, P3D_SW_FLEX_NO_INPLACE = 2 };

struct RowArgs {
    const void* x;         // observed cube (c64 or f32), [nslices][n1][N]
    const float* mask;     // [n1][N] float weights (generic path) or nullptr
    const uint16_t* bits;  // [n1][TPL] packed binary mask: bit q of entry (row, tl) = mask[row][tl + TPL*q]
    void* xc;              // compact observed samples [nslices][nobs] (type of x), row-major order of the observed
                           // positions; written by ROW_FIRST, read by the persistent row pass        (or nullptr)
    const unsigned* rowbase;  // [n1+1] number of observed positions before each row
    unsigned nobs;         // observed positions per slice = rowbase[n1]
    int* violation;        // raised by ROW_FIRST when x != 0 at a position the mask calls missing
    c32* work;             // column-blocked work buffer
    void* out;             // result cube (c64 or f32), row-major         (MID if write_out, LAST)
    const c32* tw;         // per-pass ordered twiddle tables of length N, both directions (device)
    double* sums;          // [nslices][n1] per-row sums of |x| (plain stores; reduce_rows_kernel adds them up in a
                           // fixed order -- same-address atomics from 1024 rows serialise at the memory side and,
                           // sitting in the in-order vmcnt queue, delay every later load of the wave), or nullptr
    const int* done;       // per slice: 0 running, >0 finished at that iteration, <0 all-zero slice; or nullptr
    int n1;
    int nslices;
    int sum_row;
    int dtype;             // 0 = c64, 1 = f32 (of x and out)
    int adaptive;          // APOCS input mix
    int write_out;         // MID: also store the iterate to `out` (needed only when eps > 0)
    int plain;             // LAST: plain inverse transform (no re-insertion): fft2 hook
    const uint16_t* nzm;   // MID / LAST / pipe: per (slice, tl/8) one bit per register q, clear = the column block of element
                           // tl + TPL*q was zeroed entirely by the threshold and NOT stored by the column pass (nullptr: dense)
    unsigned zero_off;     // element index (from `work`) of a zero the loads of such blocks are pointed at
    const uint8_t* nzflag; // the flexible row pass (p3d_flex.hip) reads the column pass's tile flags directly: [nslices][nz_tiles],
    int nz_tiles;          //   a tile spans nz_col_t columns
    int nz_col_t;
    int only_done;         // LAST, > 0: "finalize" launch of the early exit -- only slices whose done == only_done; their work
                           // rows hold the forward row transform of the converged iterate, which is handed to `out`
    const unsigned long long* bits64;  // rows of whole wavefronts: the mask as lane masks, word pipe64_word(row, TPL/64, wsub, q) bit l =
                                       // mask[row][64*wsub + l + TPL*q]
    const unsigned long long* nzl;     // the same for nzm, per slice (pipe64_word); nullptr: dense
    const unsigned* cbase;             // observed traces of the slice before the first column of each word (pipe64_word)
    float alpha;
    float scale;           // 1/(n1*N)
    int len;               // N, the row length (the tuned kernels know it at compile time; p3d_flex.hip reads it here)
    int real_2048;         // host side only: the row-pair path for rows of 2048 samples is switched on (experiment switch P3D_REAL_2048)
    int tstore;            // host side only: rows of one wavefront hand their transforms round through LDS and store 1-KiB runs (P3D_NO_TSTORE unset)
    int host_sw;           // host side only: P3D_SW_* experiment switches of the plan (read from the environment once per plan)
    ShearArgs sh;          // ROW_SPREAD_INV, ROW_GATHER_FWD
};

struct ColArgs {
    const c32* in;
    c32* out;           // may alias `in`
    const c32* tw;      // the column pass's twiddle tables (ColTables<N>, device)
    const c32* tau;     // [nslices][niter] (COL_ITER, optional for COL_FWD)
    const int* done;
    float* partials;    // [nslices][tiles][STATS_PARTIAL] (COL_STATS)
    int n2;
    int nslices;
    int niter;
    int iter;
    int op;
    int in_std;         // `in` is row-major [nslices][N][n2] instead of column-blocked
    int out_std;        // same for `out`
    ShearArgs sh;       // COL_SHRINK
    int len;            // N, the column length (see RowArgs::len)
    uint8_t* nzflag;    // COL_ITER*: [nslices][tiles] 1 = the tile kept at least one coefficient; tiles that kept none are
                        // neither transformed back nor stored (nullptr: always store)
    int cus;            // host side only: compute units of the plan's device (0: ask the current device)
    int host_sw;        // host side only: P3D_SW_* experiment switches of the plan
    int flex_over;      // host side only: runs per CU of the persistent flexible-length column pass (P3D_FLEX_COL_OVER, default 8)
};

// |x| for the cost sums: the hardware square root (1 ulp) without the IEEE fix-up sequence the library call expands to (8 more
// instructions per sample in a VALU-bound pass); the sums only feed the convergence test (POCS.py:622)
__device__ __forceinline__ float abs_c32(c32 v) { return __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y); }

// per-thread partial sums are float (16 terms); across the wave they are combined in double so that the
// cost, a difference of two nearly equal sums (POCS.py:622), keeps its leading digits
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// =================================================================================================
// space (row) pass
// =================================================================================================
#ifndef P3D_SHEAR_XCD
#define P3D_SHEAR_XCD 1
#endif
// BITS: the trace mask is binary and comes as one packed 16-bit word per thread and row.
template <int N, int MODE, bool BITS>
__global__ __launch_bounds__((row_threads<N, MODE>()), (row_threads<N, MODE>() >= 512 ? 4 : 3)) void row_kernel(const RowArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    constexpr int ROW_THREADS = row_threads<N, MODE>();   // shadows the global default
    constexpr int LB = ROW_THREADS / TPL;  // lines per workgroup
    constexpr int LSTR = LdsRow::stride(N);
    constexpr bool WAVE = TPL <= 64;       // a line never leaves its wavefront
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + PassTables<N>::slots();
    const TwOrdered tw{twl};

    const int tid = threadIdx.x;
    const int line = tid / TPL;
    const int tl = tid - line * TPL;
    int slice = blockIdx.y, rgroup = blockIdx.x;
    if constexpr (MODE == ROW_SPREAD_INV || MODE == ROW_GATHER_FWD) {
        // The shearlet spectra Psi_s (4 B per point and shearlet, 1 GiB at 2048 x 1024 x 125) are the same for every slice of the
        // batch: the workgroups that hold the SAME rows of different slices are made neighbours on one XCD (ids g, g + 8, ... of the
        // linear grid), so that they walk through the shearlets together and all but one of them find Psi in that XCD's L2.
        const unsigned gx = gridDim.x, nb = gridDim.y;
        if (P3D_SHEAR_XCD && nb > 1 && gx % 8 == 0) {
            const unsigned id = blockIdx.y * gx + blockIdx.x, xcd = id & 7u, j = id >> 3;
            slice = (int)(j % nb);
            rgroup = (int)((j / nb) * 8 + xcd);
        }
    }
    const int row = rgroup * LB + line;
    const bool valid = row < a.n1;

    const int dn = a.done ? a.done[slice] : 0;
    if (MODE == ROW_LAST && a.only_done) {
        if (dn != a.only_done) return;
    } else if (MODE == ROW_LAST) {
        if (dn > 0) return;  // converged earlier: `out` already holds that iterate
        if (dn < 0) {        // all-zero slice is handed back untouched (POCS.py:515-521)
            if (valid) {
                const size_t off = ((size_t)slice * a.n1 + row) * N;
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    const int e = tl + TPL * q;
                    if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[off + e] = c32{0.f, 0.f};
                    else reinterpret_cast<float*>(a.out)[off + e] = 0.f;
                }
            }
            return;
        }
    } else if (dn != 0) {
        return;
    }

    for (int i = tid; i < PassTables<N>::slots(); i += ROW_THREADS) twl[i] = a.tw[i];
    __syncthreads();

    const LdsRow lds{data + line * LSTR};
    const int vrow = valid ? row : 0;
    // Addressing: wave-uniform 64-bit bases (scalar registers) + 32-bit per-lane element offsets, so the 16
    // loads and 16 stores of a thread do not each pin a 64-bit address in VGPRs (one slice is < 2^31 elements).
    const size_t sbase = (size_t)slice * a.n1 * N;                                  // row-major cubes (x, out)
    const unsigned off = (unsigned)vrow * N + tl;                                   // + TPL*q
    c32* const wslice = a.work + (size_t)slice * wk_slice_stride(a.n1, N);          // column-blocked work buffer
    const unsigned wblk = (unsigned)a.n1 * 8;                                       // elements per column block
    const unsigned wlane = wk_lane_off<TPL>(tl, vrow, wblk);
    c32 v[PPT];

    // observed data (every mode except a plain inverse transform) and the mask word of this thread
    const bool need_obs = MODE < ROW_SPREAD_INV && ((MODE == ROW_FIRST) || !a.plain);
    unsigned mbits = 0;
    if (BITS && need_obs) mbits = valid ? a.bits[(size_t)vrow * TPL + tl] : 0u;
    auto obs_at = [&](int q) -> c32 {
        if (!valid) return c32{0.f, 0.f};
        if (a.dtype == 0) return (reinterpret_cast<const c32*>(a.x) + sbase)[off + TPL * q];
        return c32{(reinterpret_cast<const float*>(a.x) + sbase)[off + TPL * q], 0.f};
    };
    auto mask_at = [&](int q) -> float {
        if (BITS) return (float)((mbits >> q) & 1u);
        return valid ? a.mask[off + TPL * q] : 0.f;
    };

    c32 xe[P3D_XO_EARLY ? PPT : 1];
    if (P3D_XO_EARLY && MODE != ROW_FIRST && !a.plain) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) xe[q] = obs_at(q);
    }

    float acc = 0.f;
    if (MODE == ROW_FIRST) {
        // ranks come from wave ballots: a line inside one wave counts as it goes (CompactIndex); a line of several waves takes
        // the number of observed traces before each of its 64-column words from the table of the persistent pass (RowArgs::cbase)
        // The order of the compact array is a convention between this kernel and the persistent passes.  Where the table of the
        // wave-uniform pass exists (RowArgs::cbase: rows of 128 ... 4096 samples) a wavefront's samples of one register q are
        // consecutive: index = cbase[word] + rank of the lane among the set lanes of the WAVE (for rows shorter than a wavefront
        // the word spans the 64 / TPL adjacent rows the wave holds).  Otherwise: row-major, counted line by line.
        constexpr bool WORDS = BITS && PPT == 16 && TPL >= 8 && TPL <= 256;
        constexpr bool CAN_COMPACT = BITS && (TPL <= 64 || WORDS);
        const bool by_words = WORDS && a.cbase != nullptr;
        const bool compact = CAN_COMPACT && a.xc != nullptr && (TPL <= 64 || by_words);
        CompactIndex<(TPL <= 64 ? TPL : 64)> ci(tid & 63, (compact && !by_words) ? a.rowbase[vrow] : 0u);
        constexpr int RPW_ = TPL >= 64 ? 1 : 64 / TPL, WPL_ = TPL >= 64 ? TPL / 64 : 1;
        const size_t word0 = pipe64_word((size_t)(vrow / RPW_), WPL_, (tid >> 6) % WPL_, 0);
        bool bad = false;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const c32 x = obs_at(q);
            if (CAN_COMPACT) {
                if (compact) {  // uniform
                    const bool set = ((mbits >> q) & 1u) != 0;
                    unsigned idx;
                    if (by_words) {   // uniform
                        const unsigned long long b = __ballot(set && valid);
                        idx = a.cbase[word0 + q] + (unsigned)__popcll(b & ((1ull << (tid & 63)) - 1ull));
                    } else {
                        idx = ci.next(set);
                    }
                    if (set && valid) {
                        if (a.dtype == 0) reinterpret_cast<c32*>(a.xc)[(size_t)slice * a.nobs + idx] = x;
                        else reinterpret_cast<float*>(a.xc)[(size_t)slice * a.nobs + idx] = x.x;
                    }
                    bad = bad || (!set && (x.x != 0.f || x.y != 0.f));
                }
            }
            acc += abs_c32(x);
            if (a.adaptive) {
                // x_old = x at the first iteration (POCS.py:549, 574-575)
                const float m = mask_at(q);
                const float w = 1.0f - a.alpha * m;
                const c32 blend = x * a.alpha + x * w;
                v[q] = blend + (x - x * m) * (1.0f - a.alpha);
            } else {
                v[q] = x;
            }
        }
        if (CAN_COMPACT) {
            if (compact && bad && valid) atomicOr(a.violation, 1);
        }
    } else if (MODE == ROW_SPREAD_INV) {
        // work[b*nsh + s] = inverse row FFT of Psi_s * F[b] for every s   (F = a.x: spectra of slice b, row-major; grid.y = b).
        // The row of F is read once and kept in registers across the shearlets.
        const c32* const f = reinterpret_cast<const c32*>(a.x) + sbase;
        c32 fr[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) fr[q] = valid ? f[off + TPL * q] : c32{0.f, 0.f};
        for (int s = 0; s < a.sh.nsh; ++s) {
            const float* const w = a.sh.psi + (size_t)s * a.n1 * N;
            c32* const ws = a.work + ((size_t)slice * a.sh.nsh + s) * wk_slice_stride(a.n1, N);
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = fr[q] * (valid ? w[off + TPL * q] : 0.f);
            line_fft<N, INV, WAVE>(v, lds, tw, tl);
            __syncthreads();   // adjacent rows share 128-byte lines of the work buffer: store them together
            if (valid) {
#pragma unroll
                for (int q = 0; q < PPT; ++q) wk_q_ptr<TPL>(ws, q, tl, wblk)[wlane] = v[q];
            }
        }
        return;
    } else if (MODE == ROW_GATHER_FWD) {
        // out[b] = sum_s Psi_s * forward row FFT of work[b*nsh + s]   (out row-major spectra; grid.y = b)
        c32 acc[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) acc[q] = c32{0.f, 0.f};
        for (int s = 0; s < a.sh.nsh; ++s) {
            const c32* const ws = a.work + ((size_t)slice * a.sh.nsh + s) * wk_slice_stride(a.n1, N);
            const float* const w = a.sh.psi + (size_t)s * a.n1 * N;
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = valid ? wk_q_ptr<TPL>(ws, q, tl, wblk)[wlane] : c32{0.f, 0.f};
            line_fft<N, FWD, WAVE>(v, lds, tw, tl);
            // the weights are fetched after the transform, four at a time, to keep the register count of the transform
#pragma unroll
            for (int g = 0; g < PPT; g += 4) {
                float wq[4];
#pragma unroll
                for (int i = 0; i < 4 && g + i < PPT; ++i) wq[i] = valid ? w[off + TPL * (g + i)] : 0.f;
#pragma unroll
                for (int i = 0; i < 4 && g + i < PPT; ++i) acc[g + i] = acc[g + i] + v[g + i] * wq[i];
            }
        }
        if (valid) {
            c32* const o = reinterpret_cast<c32*>(a.out) + sbase;
#pragma unroll
            for (int q = 0; q < PPT; ++q) o[off + TPL * q] = acc[q];
        }
        return;
    } else {
        if constexpr (TPL % 8 == 0) {
            if (a.nzm != nullptr && !a.only_done) {   // blocks the column pass did not store read a zero instead
                const unsigned nz = a.nzm[(size_t)slice * (TPL / 8) + (tl >> 3)];
                const unsigned zbase = a.zero_off - (unsigned)slice * (unsigned)wk_slice_stride(a.n1, N);
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    const unsigned o = ((nz >> q) & 1u) ? wlane : zbase - (unsigned)q * (TPL / 8) * wblk;
                    v[q] = valid ? wk_q_ptr<TPL>(wslice, q, tl, wblk)[o] : c32{0.f, 0.f};
                }
            } else {
#pragma unroll
                for (int q = 0; q < PPT; ++q) v[q] = valid ? wk_q_ptr<TPL>(wslice, q, tl, wblk)[wlane] : c32{0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = valid ? wk_q_ptr<TPL>(wslice, q, tl, wblk)[wlane] : c32{0.f, 0.f};
        }
        line_fft<N, INV, WAVE>(v, lds, tw, tl);
        // The observed samples are fetched here, a few at a time, instead of being prefetched ahead of
        // the inverse transform: holding 16 of them across the transform costs 32 VGPRs and the 16
        // waves per CU this kernel is budgeted for (128 VGPRs) cover the latency instead.
        constexpr int G = PPT < P3D_XO_GROUP ? PPT : P3D_XO_GROUP;
        asm volatile("" : "+v"(mbits));  // keep the 16 mask weights from being expanded ahead of the transform
#pragma unroll
        for (int g = 0; g < PPT; g += G) {
            c32 xo[G];
            if (!a.plain) {
#pragma unroll
                for (int i = 0; i < G; ++i) xo[i] = P3D_XO_EARLY ? xe[g + i] : obs_at(g + i);
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int q = g + i;
                c32 xn = v[q] * a.scale;
                float m = 0.f;
                if (MODE == ROW_LAST && a.only_done) {
                    // xn = the converged iterate up to the round-off of one row-transform round trip; where a trace was
                    // observed and alpha = 1 the iterate IS the observed sample (POCS.py:616-619): hand that back exactly
                    if (a.alpha == 1.0f && mask_at(q) == 1.0f) xn = xo[i];
                } else if (!a.plain) {
                    m = mask_at(q);
                    const float w = 1.0f - a.alpha * m;       // POCS.py:616
                    xn = axpby(xn, w, xo[i], a.alpha);        // POCS.py:619
                }
                acc += abs_c32(xn);
                if (MODE == ROW_LAST || a.write_out) {
                    if (valid) {
                        if (a.dtype == 0) (reinterpret_cast<c32*>(a.out) + sbase)[off + TPL * q] = xn;
                        else (reinterpret_cast<float*>(a.out) + sbase)[off + TPL * q] = xn.x;  // np.real(), POCS.py:656
                    }
                }
                if (MODE == ROW_MID) {
                    if (a.adaptive) {  // x_input of the next iteration (POCS.py:574-575)
                        const float w = 1.0f - a.alpha * m;
                        const c32 blend = xo[i] * a.alpha + xn * w;
                        v[q] = blend + (xo[i] - xn * m) * (1.0f - a.alpha);
                    } else {
                        v[q] = xn;
                    }
                }
            }
            if (!P3D_XO_EARLY) __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (a.sums != nullptr) {  // one line = TPL consecutive lanes (TPL > 64: several waves, combined through LDS below)
        double ws = valid ? (double)acc : 0.0;
        if constexpr (TPL <= 64) {
#pragma unroll
            for (int o = TPL / 2; o > 0; o >>= 1) ws += __shfl_down(ws, o, TPL);
            if (tl == 0 && valid) a.sums[(size_t)slice * a.n1 + row] = ws;
        } else {
            ws = wave_sum(ws);
            double* red = reinterpret_cast<double*>(data + LB * LSTR);  // scratch behind the line buffers
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = ws;
            __syncthreads();
            if (tl == 0 && valid) {
                double t = 0.0;
                for (int w = 0; w < TPL / 64; ++w) t += red[line * (TPL / 64) + w];
                a.sums[(size_t)slice * a.n1 + row] = t;
            }
            __syncthreads();
        }
    }

    if (MODE != ROW_LAST) {
        line_fft<N, FWD, WAVE>(v, lds, tw, tl);
        __syncthreads();   // the rows of a workgroup are adjacent and share 128-byte lines of the work buffer: store together
        if (valid) {
            for (int q = 0; q < PPT; ++q) wk_q_ptr<TPL>(wslice, q, tl, wblk)[wlane] = v[q];
        }
    }
}

// =================================================================================================
// space (row) pass, steady state: persistent, software pipelined across rows
// =================================================================================================
// Same arithmetic as row_kernel<N, ROW_MID, BITS> (bit for bit), different schedule.  Both passes are bound
// by the bytes a CU keeps in flight, and that is capped by registers: a row in progress occupies ~155
// VGPRs per thread, i.e. 12 waves per CU.  Here every wave walks over many rows and keeps one row's worth
// of loads in flight WHILE it computes: the observed samples of row r arrive during the inverse transform
// of row r, and the work-buffer loads of the NEXT row arrive during the forward transform of row r, in the
// registers the observed samples just vacated (no extra VGPRs).
// Requires TPL <= 64 (a line never leaves its wavefront: no workgroup barrier inside the loop).
// EXTRA: the rarely used options (APOCS input mix, per-iteration output for early exit) are compiled in.
// COMPACT: the observed samples are read from the compact array (BITS only).
template <int N, bool BITS, int DT, bool EXTRA, bool COMPACT>
__global__ __launch_bounds__(ROW_THREADS, (COMPACT && P3D_COMPACT_LATE) ? P3D_PIPE_WAVES_PER_EU_COMPACT : P3D_PIPE_WAVES_PER_EU) void
row_pipe_kernel(const RowArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    static_assert(TPL <= 64, "line must fit a wavefront");
    constexpr int LB = ROW_THREADS / TPL;
    constexpr int LSTR = LdsRow::stride(N);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + PassTables<N>::slots();
    const TwOrdered tw{twl};

    const int tid = threadIdx.x;
    const int line = tid / TPL;
    const int tl = tid - line * TPL;
    for (int i = tid; i < PassTables<N>::slots(); i += ROW_THREADS) twl[i] = a.tw[i];
    __syncthreads();
    const LdsRow lds{data + line * LSTR};

    // Addressing: every cube pointer stays wave-uniform (a.work + q*qstride etc., scalar registers); the
    // position of a line is ONE 32-bit element offset per lane (the launcher guarantees the batch has
    // fewer than 2^32 elements), because lines sharing a wave (N < 1024) sit in different rows or slices.
    const unsigned total = (unsigned)a.nslices * a.n1;   // lines of the whole batch
    const unsigned step = gridDim.x * LB;                // lines per sweep of the grid
    const unsigned wblk = (unsigned)a.n1 * 8;
    const unsigned wstride = (unsigned)wk_slice_stride(a.n1, N);

    struct Where { unsigned slice, row; bool on; };
    // NOTE on ordering: s_waitcnt vmcnt counts vector-memory operations IN ISSUE ORDER, so a small load issued
    // after a bulk prefetch cannot be consumed without draining the prefetch as well.  Every small per-row load
    // (mask word, `done` flag of the slice) is therefore issued one row early and AHEAD of the bulk loads of
    // that iteration; the fast path (EXTRA = false) has no `done` lookup at all.
    auto locate = [&](unsigned g) -> Where {
        Where w;
        w.on = g < total;
        const unsigned gg = w.on ? g : 0u;
        w.slice = gg / (unsigned)a.n1;
        w.row = gg - w.slice * (unsigned)a.n1;
        if (EXTRA) {
            if (w.on && a.done && a.done[w.slice] != 0) w.on = false;   // finished / empty slice: leave it alone
        }
        return w;
    };
    auto wlane = [&](const Where& w) -> unsigned { return w.slice * wstride + wk_lane_off<TPL>(tl, (int)w.row, wblk); };

    unsigned g = blockIdx.x * LB + line;
    Where cur = locate(g);
    Where nxt = locate(g + step);
    // Software pipeline, one full row deep.  While row r is being transformed, two sets of loads are in
    // flight per wave: by[] <- work buffer of row r+1 (issued at the top of row r, consumed at the top of row
    // r+1) and bx[] <- observed samples of row r+1 (issued right after the re-insertion of row r freed bx[],
    // consumed by the re-insertion of row r+1).  Loads are never predicated: a line that is switched off
    // (beyond the end, finished or empty slice) reads line 0 instead (locate() clamps) and its results are
    // simply not stored or summed.
    c32 v[PPT], bx[PPT];
#if P3D_PIPE_PREFETCH
    c32 by[PPT];
#endif
    // nz: bit q clear = the column block of register q was emptied by the threshold and not stored (see RowArgs::nzm)
    constexpr bool CAN_SPARSE = TPL % 8 == 0;
    const bool sparse = CAN_SPARSE && a.nzm != nullptr;
    auto load_work = [&](c32 (&dst)[PPT], const Where& w, unsigned nz) {
        const unsigned wl = wlane(w);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            unsigned o = wl;
            if (CAN_SPARSE) {
                if (sparse) o = ((nz >> q) & 1u) ? wl : a.zero_off - (unsigned)q * (TPL / 8) * wblk;
            }
            dst[q] = wk_q_ptr<TPL>((const c32*)a.work, q, tl, wblk)[o];
        }
    };
    auto nz_of = [&](const Where& w) -> unsigned { return sparse ? (unsigned)a.nzm[w.slice * (TPL / 8) + (tl >> 3)] : 0xffffu; };
    // (wbits, wbase): mask word / compact row base of the row being loaded (fetched a row earlier, see NOTE)
    auto load_obs = [&](c32 (&dst)[PPT], const Where& w, unsigned wbits, unsigned wbase) {
        if constexpr (COMPACT) {
            CompactIndex<TPL> ci(tid & 63, w.slice * a.nobs + wbase);
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const bool set = ((wbits >> q) & 1u) != 0;
                const unsigned idx = ci.next(set);
                c32 val{0.f, 0.f};
                if (set) {
                    if (DT == 0) val = reinterpret_cast<const c32*>(a.xc)[idx];
                    else val.x = reinterpret_cast<const float*>(a.xc)[idx];
                }
                dst[q] = val;
            }
        } else {
            const unsigned off = (w.slice * (unsigned)a.n1 + w.row) * N + tl;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                if (DT == 0) dst[q] = (reinterpret_cast<const c32*>(a.x) + TPL * q)[off];
                else dst[q] = c32{(reinterpret_cast<const float*>(a.x) + TPL * q)[off], 0.f};
            }
        }
    };
    unsigned mbits = 0, rbase = 0;
    if (BITS) mbits = a.bits[cur.row * TPL + tl];
    if (COMPACT) rbase = a.rowbase[cur.row];
    unsigned nz_nxt = nz_of(nxt);   // consumed by the prefetch of the next row: fetched a row early like the mask word
#if P3D_PIPE_PREFETCH
    load_work(by, cur, nz_of(cur));
#else
    unsigned nz_cur = nz_of(cur);
#endif
    constexpr bool LATE = COMPACT && P3D_COMPACT_LATE;
    if (!LATE) load_obs(bx, cur, mbits, rbase);

    // every line of the workgroup runs the same number of sweeps (uniform loop, predicated work)
    for (unsigned g0 = blockIdx.x * LB; g0 < total; g0 += step) {
#if P3D_PIPE_LOCKSTEP
        // The lines of a workgroup are ADJACENT rows, and in the column-blocked work buffer adjacent rows share
        // 128-byte lines (64 bytes each).  Keeping the waves in step makes the two halves of a line arrive at
        // L2 together; waves that drift apart turn every line into two partial-line transactions.
        __syncthreads();
#endif
        // small loads of the rows ahead first (see NOTE), then the bulk prefetch of row r+1
        const Where nxt2 = locate(g + 2 * step);
        unsigned mbits_nxt = 0, rbase_nxt = 0;
        if (BITS) mbits_nxt = a.bits[nxt.row * TPL + tl];
        if (COMPACT) rbase_nxt = a.rowbase[nxt.row];
        const unsigned nz_nxt2 = nz_of(nxt2);
#if P3D_PIPE_PREFETCH
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = by[q];
        load_work(by, nxt, nz_nxt);
#else
        load_work(v, cur, nz_cur);
#endif
        const unsigned off = (cur.slice * (unsigned)a.n1 + cur.row) * N + tl;

        __builtin_amdgcn_sched_barrier(0);
        line_fft<N, INV, true>(v, lds, tw, tl);
        __builtin_amdgcn_sched_barrier(0);
        if (LATE) {
            load_obs(bx, cur, mbits, rbase);
            __builtin_amdgcn_sched_barrier(0);
        }

        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            c32 xn = v[q] * a.scale;
            float m;
            if (BITS) m = (float)((mbits >> q) & 1u);
            else m = (a.mask + TPL * q)[cur.row * N + tl];
            const float w = 1.0f - a.alpha * m;       // POCS.py:616
            xn = axpby(xn, w, bx[q], a.alpha);        // POCS.py:619
            acc += abs_c32(xn);
            if (EXTRA && a.write_out && cur.on) {
                if (DT == 0) (reinterpret_cast<c32*>(a.out) + TPL * q)[off] = xn;
                else (reinterpret_cast<float*>(a.out) + TPL * q)[off] = xn.x;
            }
            if (EXTRA && a.adaptive) {  // x_input of the next iteration (POCS.py:574-575)
                const c32 blend = bx[q] * a.alpha + xn * w;
                v[q] = blend + (bx[q] - xn * m) * (1.0f - a.alpha);
            } else {
                v[q] = xn;
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // bx[] is free only now: keep the next loads below this point
        if (!LATE) load_obs(bx, nxt, mbits_nxt, rbase_nxt);

        if (a.sums != nullptr) {  // one line = TPL consecutive lanes: segmented reduction
            double ws = cur.on ? (double)acc : 0.0;
#pragma unroll
            for (int o = TPL / 2; o > 0; o >>= 1) ws += __shfl_down(ws, o, TPL);
            if (tl == 0 && cur.on) a.sums[(size_t)cur.slice * a.n1 + cur.row] = ws;
        }

        __builtin_amdgcn_sched_barrier(0);
        line_fft<N, FWD, true>(v, lds, tw, tl);
        __builtin_amdgcn_sched_barrier(0);

        if (cur.on) {
            const unsigned wl = wlane(cur);
#pragma unroll
            for (int q = 0; q < PPT; ++q) wk_q_ptr<TPL>(a.work, q, tl, wblk)[wl] = v[q];
        }
        g += step;
        cur = nxt;
        nxt = nxt2;
        mbits = mbits_nxt;
        rbase = rbase_nxt;
#if !P3D_PIPE_PREFETCH
        nz_cur = nz_nxt;
#endif
        nz_nxt = nz_nxt2;
    }
}

// =================================================================================================
// persistent row pass in units of wavefronts (rows of 128 ... 4096 samples), binary mask, compact observed samples
// =================================================================================================
// Same arithmetic as row_kernel<N, ROW_MID, true> / row_pipe_kernel (bit for bit); what changes is WHERE the bookkeeping runs.
// A wavefront works on 64 consecutive columns of one row per register q (rows of 1024 / 2048 / 4096 samples = 1 / 2 / 4
// wavefronts) or on the same columns of 2 / 4 / 8 adjacent rows (512 / 256 / 128 samples), so slice, row, every base address,
// the trace mask of those samples and the emptied-block flags of the slice are wave-uniform: they live in scalar registers (s_load / SALU), predicates
// are 64-bit lane masks applied as EXEC or as the selector of v_cndmask, the rank of a lane among the observed traces is
// v_mbcnt, and every access is "scalar base + one 32-bit lane offset".  The generic kernel spends ~40 % of its vector
// instructions on exactly that bookkeeping.  Rows of 2048 / 4096 samples (2 / 4 wavefronts, workgroup barriers inside the
// transforms) had no persistent pass at all: each 2-row workgroup of row_kernel re-reads 32 / 64 KiB of twiddle tables.
// Measured on the headline cube (profiles/r01_rowpass_wave_uniform.txt): sixteen rows per workgroup (1024 threads, one workgroup
// of 154 KiB LDS per CU, 4 waves per SIMD inside the 128-VGPR budget) beats three 4-row workgroups; a row-ahead prefetch of the
// work buffer (tried: 32 more VGPRs) buys nothing once most of its blocks are skipped, the early request of the observed samples a
// little.
#ifndef P3D_PIPE64_LOCKSTEP
#define P3D_PIPE64_LOCKSTEP 1
#endif
#ifndef P3D_PIPE64_XCD
#define P3D_PIPE64_XCD 1
#endif
#ifndef P3D_PIPE64_MAXROWS
#define P3D_PIPE64_MAXROWS 2
#endif

#ifndef P3D_ABL_NOSTORE   // ablations of the persistent row pass (timing experiments, results wrong): tools/rowpass_ablation.sh
#define P3D_ABL_NOSTORE 0
#endif
#ifndef P3D_ABL_NOFFT
#define P3D_ABL_NOFFT 0
#endif
#ifndef P3D_ABL_NOSUMS
#define P3D_ABL_NOSUMS 0
#endif
#ifndef P3D_ABL_NOOBS
#define P3D_ABL_NOOBS 0
#endif
#ifndef P3D_ABL_NOWORK
#define P3D_ABL_NOWORK 0
#endif
// rows per workgroup: as many as 160 KiB of LDS hold next to the twiddle tables, at most 1024 threads
template <int N>
constexpr int pipe64_rows()
{
    constexpr int TPL = Plan<N>::TPL;
    // rows of several wavefronts synchronise the whole workgroup at every exchange of a transform: two rows per workgroup (the
    // pair that shares 128-byte lines), several workgroups per CU (2048 samples: 2.19 ms against 2.83 with 7 rows, 5.87 before)
    int rows = TPL > 64 ? P3D_PIPE64_MAXROWS : ((P3D_EXP_HALFWG && TPL == 64) ? 8 : 1024 / TPL);
    while (rows > 1 && sizeof(c32) * (PassTables<N>::slots() + (size_t)rows * (LdsRow::stride(N) + (TPL == 64 ? 4 : 0))) + 16 * sizeof(double) > 160 * 1024) --rows;
    return rows;
}
// Rows of ONE wavefront (N = 1024) can hand their forward transforms to each other through LDS before storing (TS, see
// row_pipe64_kernel): the row buffers are then read ACROSS rows, and a row pitch of 8704 bytes = 0 mod 256 would put all sixteen
// rows on the same banks; four more slots per row (32 bytes = 8 banks) spread them.
template <int N>
constexpr bool pipe64_can_tstore() { return Plan<N>::TPL == 64; }
template <int N>
constexpr int pipe64_lstr() { return LdsRow::stride(N) + (pipe64_can_tstore<N>() ? 4 : 0); }
template <int N>
constexpr size_t pipe64_lds_bytes() { return sizeof(c32) * (PassTables<N>::slots() + (size_t)pipe64_rows<N>() * pipe64_lstr<N>()) + 16 * sizeof(double); }
template <int N>
constexpr int pipe64_threads() { return pipe64_rows<N>() * Plan<N>::TPL; }

// ---- buffer addressing (wave-uniform descriptor + per-lane 32-bit byte offset) --------------------------------------------------
// A predicated access is written as an UNCONDITIONAL buffer instruction whose switched-off lanes carry an offset beyond the
// descriptor's range: the hardware range check returns zero for such a load lane / drops such a store lane without touching
// memory.  That matters beyond the saved branch: `s_waitcnt vmcnt` counts in issue order, and hipcc can only count exactly through
// straight-line code -- with one `s_cbranch_execz` per predicated global_load (what `if (lane_pred) x = *p;` compiles to) every
// wait of the loop became vmcnt(0), i.e. each row waited for the write acknowledgements of the row before it
// (profiles/r02_rowpass_*.txt).
typedef unsigned p3d_u2 __attribute__((ext_vector_type(2)));
constexpr unsigned BUF_OOB = 0x80000000u;   // every descriptor below spans less than 2 GiB
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_srd(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// (loads whose results are carried around a loop travel as raw 64-bit integers: a loop-carried pair of floats invites the
// vectoriser to keep it shuffled, and the copies that undo the shuffle sit -- with their wait -- in front of the back edge)
typedef unsigned long long raw64;
__device__ __forceinline__ raw64 buf_load_raw64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const p3d_u2 t = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return (raw64)t.x | ((raw64)t.y << 32);
}
__device__ __forceinline__ c32 raw_c32(raw64 u) { return c32{__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))}; }
__device__ __forceinline__ c32 buf_load_c32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return raw_c32(buf_load_raw64(r, voff, soff)); }
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store_c32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, c32 v)
{
    __builtin_amdgcn_raw_buffer_store_b64(p3d_u2{__float_as_uint(v.x), __float_as_uint(v.y)}, r, (int)voff, (int)soff, 0);
}
typedef unsigned p3d_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_store_2c32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, c32 a, c32 b)
{
    __builtin_amdgcn_raw_buffer_store_b128(p3d_u4{__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.y)}, r,
                                           (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    __builtin_amdgcn_raw_buffer_store_b64(p3d_u2{(unsigned)u, (unsigned)(u >> 32)}, r, (int)voff, (int)soff, 0);
}

// In-kernel stamps (diagnostic build -DP3D_STAMPS=1 only, tools/rowpass_stamps.sh): cycles each wave of row_pipe64_kernel spends
// between fixed points of a row, summed over its rows, in a buffer of their own that nothing else reads.
#ifndef P3D_STAMPS
#define P3D_STAMPS 0
#endif
#if P3D_STAMPS
constexpr int STAMP_PHASES = 10;
static __device__ unsigned p3d_stamp_buf[1024 * 16 * STAMP_PHASES];   // (one copy per translation unit; the reader sits next to the kernels)
#define P3D_STAMP(i)                                                     \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();      \
        st_acc[i] += (unsigned)(t_ - st_prev);                           \
        st_prev = t_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)
#else
#define P3D_STAMP(i) do { } while (0)
#endif

// (Tried and dropped in round 1: a wavefront that owns TWO adjacent rows and stores them together, so that the 64-byte halves of a
// line pair up inside the wave and the lock-step barrier can go -- 12 waves per CU at 168 VGPRs: 2.03 ms against 1.92, niter = 10.)
//
// Schedule of one row (round 2).  Vector-memory operations retire in issue order, so a wave that waits for a load also waits for
// every store it issued before that load.  The loop therefore never issues a load behind the stores it does not want to wait for:
//
//     top of row r:   v <- by            (work-buffer elements of row r, requested before the forward transform of row r-1)
//                     inverse transform
//                     re-insertion with bx (observed samples of row r, requested behind the forward transform of row r-1), sum |x|
//                     scalar tables of row r+1; by <- work-buffer elements of row r+1     <- in flight during the forward transform
//                     forward transform
//                     bx <- observed samples of row r+1
//                     lock-step barrier, stores of row r          <- a whole row of arithmetic passes before anything behind them
//                                                                    is waited for
// Every access is an unconditional buffer instruction (see above) except the work-buffer loads of emptied blocks, which are OLDER
// than everything a later wait has to leave outstanding; the prologue issues the same number of (out-of-range) stores as the loop
// body, so the compiler's wait counts at the loop header are exact: `vmcnt(32)` for the work-buffer elements (16 observed-sample
// loads and 16 stores stay in flight), `vmcnt(31 ... 16)` for the samples, where round 1 had `vmcnt(0)` throughout.
// What this bought, and what it did not: profiles/r02_rowpass_schedule.txt.
// PM: which pass of a job.  PIPE_MID: the steady state described above.  PIPE_FIRST: observed cube -> compact copy of the observed
// samples, sum |x_obs|, forward row transform -> work buffer (what row_kernel<ROW_FIRST> does, at 2.8 TB/s; without the lane-mask
// tables -- the statistics pass has no mask yet -- only the transform).  PIPE_LAST: work buffer -> inverse row transform ->
// re-insertion -> result cube (row_kernel<ROW_LAST> reads the FULL observed cube for that, zeros included: 8.6 GB where 5.2 do).
enum PipeMode { PIPE_MID = 0, PIPE_FIRST = 1, PIPE_LAST = 2 };

// TS (PIPE_MID, rows of one wavefront, n1 a multiple of the 16 rows of a workgroup): the forward transforms are handed round through
// LDS before they are stored, so that ONE dwordx4 instruction writes the 1-KiB run [16 rows][8 columns] of a column block -- whole
// 128-byte lines, half the line accesses of sixteen rows' 64-byte pieces and half the store instructions (8 instead of 16).
template <int N, int DT, bool SPARSE, int PM, bool TS = false>
__global__ __launch_bounds__((pipe64_threads<N>()), (P3D_EXP_HALFWG ? 4 : (pipe64_threads<N>() / 64 + 3) / 4)) void row_pipe64_kernel(const RowArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    static_assert((TPL % 64 == 0 || 64 % TPL == 0) && TPL >= 8 && PPT == 16, "whole wavefronts per row or whole rows per wavefront");
    constexpr int WPL = TPL >= 64 ? TPL / 64 : 1; // wavefronts per row
    constexpr int RPW = TPL >= 64 ? 1 : 64 / TPL; // rows per wavefront: a "unit" = RPW adjacent rows of one slice (n1 % RPW == 0)
    constexpr int THREADS = pipe64_threads<N>();
    constexpr int LB = THREADS / TPL;             // rows per workgroup
    constexpr int UPB = LB / RPW;                 // units per workgroup
    constexpr int LSTR = pipe64_lstr<N>();
    constexpr bool WAVE = WPL == 1;
    static_assert(!TS || (PM == PIPE_MID && pipe64_can_tstore<N>() && LB == 16), "transposed stores: sixteen one-wavefront rows per workgroup");
    constexpr unsigned ES = DT == 0 ? 8u : 4u;    // bytes per observed sample
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + PassTables<N>::slots();
    const TwOrdered tw{twl};

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int uline = wave / WPL, wsub = wave % WPL;
    const int lane = tid & 63;
    const int sub = TPL >= 64 ? 0 : lane / TPL;                        // row of this lane inside its unit
    const int tl = TPL >= 64 ? wsub * 64 + lane : lane % TPL;
    const int line = uline * RPW + sub;
    for (int i = tid; i < PassTables<N>::slots(); i += THREADS) twl[i] = a.tw[i];
    __syncthreads();
    const LdsRow lds{data + line * LSTR};
    double* red = reinterpret_cast<double*>(data + LB * LSTR);   // per-wave partial sums of rows that span waves

    const unsigned upslice = (unsigned)a.n1 / RPW;                      // units per slice
    const unsigned total = (unsigned)a.nslices * upslice;
    const unsigned wblk = (unsigned)a.n1 * 8;
    const size_t wstride = wk_slice_stride(a.n1, N);
    const unsigned slice_bytes = (unsigned)(wstride * 8);               // one slice of the work buffer: at most 128 MiB
    // element tl + TPL*q = column 64*(wsub + WPL*q) + lane (rows of whole wavefronts) or TPL*q + tl of row `sub` of the unit:
    // min(TPL, 64) / 8 column blocks per wavefront and register, adjacent rows 64 bytes apart
    const unsigned colpart = TPL >= 64 ? (unsigned)lane : (unsigned)tl;
    const unsigned lane_w = ((colpart >> 3) * wblk + (colpart & 7) + (unsigned)sub * 8u) * 8u;   // byte offset of the lane, every q
    const unsigned qs64 = 8u * wblk * 8u;                                                        // bytes per 64 columns
    const unsigned qs = TPL >= 64 ? qs64 * WPL : qs64 / RPW;                                     // bytes from register q to q + 1

    // The small tables (lane masks, compact bases) are never written while this kernel runs: reading them through the constant
    // address space lets the compiler use scalar loads although the loop also stores to the work buffer.
    typedef const unsigned long long __attribute__((address_space(4))) * kmask_t;
    typedef const unsigned __attribute__((address_space(4))) * kuint_t;
    typedef const int __attribute__((address_space(4))) * kint_t;
    const kmask_t k_bits = (kmask_t)a.bits64, k_nzl = (kmask_t)a.nzl;
    const kuint_t k_cbase = (kuint_t)a.cbase;
    const kint_t k_done = (kint_t)a.done;   // early exit (eps > 0): set between launches, constant during one

    // row: the unit's index inside its slice (= the row itself when RPW == 1); on: the unit is computed and stored; zero (PIPE_LAST):
    // the unit belongs to an all-zero slice, which is handed back untouched (POCS.py:515-521)
    struct Where { unsigned slice, row; bool on, zero; };
    auto locate = [&](unsigned g) -> Where {
        Where w;
        w.on = g < total;
        w.zero = false;
        const unsigned gg = w.on ? g : 0u;
        w.slice = gg / upslice;
        w.row = gg - w.slice * upslice;
        if (k_done != nullptr && w.on) {
            const int dn = k_done[w.slice];
            if (PM == PIPE_LAST) {   // converged earlier (dn > 0): `out` already holds that iterate
                w.zero = dn < 0;
                w.on = dn <= 0;
            } else if (dn != 0) {
                w.on = false;        // finished / empty slice: leave it alone
            }
        }
        return w;
    };
    auto work_srd = [&](const Where& w) { return buf_srd(reinterpret_cast<const char*>(a.work) + w.slice * wstride * 8, slice_bytes); };
    auto work_soff = [&](const Where& w) -> unsigned { return w.row * (unsigned)(RPW * 64) + (unsigned)wsub * qs64; };
    // by[] <- the unit's elements of the work buffer; emptied column blocks (SPARSE) read as zero without a memory access
    auto issue_work = [&](raw64 (&dst)[PPT], const Where& w) {
        const __amdgpu_buffer_rsrc_t srd = work_srd(w);
        unsigned so = work_soff(w);
        const kmask_t nz = k_nzl + pipe64_word(w.slice, WPL, wsub, 0);
        unsigned long long nzw[PPT];
        if (SPARSE) {
#pragma unroll
            for (int q = 0; q < PPT; ++q) nzw[q] = nz[q];   // all sixteen words in one go (s_load_dwordx16 twice)
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (SPARSE) {
                // A register whose 64 columns were all emptied (most of them, late in a schedule) skips the instruction: an
                // all-out-of-range load moves no data but still occupies the address unit (16 of them per row: +0.24 ms on
                // the headline cube).  The branch is wave-uniform; it does not disturb the wait counts, because every
                // operation whose count varies this way is OLDER than all those a later wait has to leave outstanding.
                raw64 r = 0;
                if (nzw[q] != 0 && !P3D_ABL_NOWORK) r = buf_load_raw64(srd, __builtin_amdgcn_inverse_ballot_w64(nzw[q]) ? lane_w : BUF_OOB, so);
                dst[q] = r;
            } else {
                dst[q] = buf_load_raw64(srd, P3D_ABL_NOWORK ? BUF_OOB : lane_w, so);
            }
            so += qs;
        }
    };
    // (the mask words are loaded again for the re-insertion instead of being kept across the transform: together with the compact
    // bases and the emptied-block words they do not fit the scalar registers, and a spilled word costs a v_readlane per use)
    auto words_of = [&](const Where& w) -> kmask_t { kmask_t m = k_bits + pipe64_word(w.row, WPL, wsub, 0); asm volatile("" : "+s"(m)); return m; };
    // bx[] <- the unit's observed samples from the compact array (zero where the trace is missing)
    auto obs_tables = [&](unsigned long long (&mwords)[PPT], unsigned (&cbs)[PPT], const Where& w) {
        const kmask_t mrow = words_of(w);
        const kuint_t cb = k_cbase + pipe64_word(w.row, WPL, wsub, 0);   // observed traces before this word, from the start of the slice
#pragma unroll
        for (int q = 0; q < PPT; ++q) { mwords[q] = mrow[q]; cbs[q] = cb[q]; }
    };
    auto issue_obs_with = [&](raw64 (&dst)[PPT], const Where& w, const unsigned long long (&mwords)[PPT], const unsigned (&cbs)[PPT]) {
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.xc) + (size_t)w.slice * a.nobs * ES, a.nobs * ES);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const unsigned long long mw = mwords[q];
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mw >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mw, 0u));
            const unsigned vo = (__builtin_amdgcn_inverse_ballot_w64(mw) && !P3D_ABL_NOOBS) ? rank * ES : BUF_OOB;
            if (DT == 0) dst[q] = buf_load_raw64(srd, vo, cbs[q] * ES);
            else dst[q] = (raw64)__builtin_amdgcn_raw_buffer_load_b32(srd, (int)vo, (int)(cbs[q] * ES), 0);   // (imaginary part: zero bits)
        }
    };
    auto store_work = [&](const c32 (&src)[PPT], const Where& w, bool really) {
        const __amdgpu_buffer_rsrc_t srd = work_srd(w);
        unsigned so = work_soff(w);
        const unsigned vo = (really && !P3D_ABL_NOSTORE) ? lane_w : BUF_OOB;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            buf_store_c32(srd, vo, so, src[q]);
            so += qs;
        }
    };
    // row-major cubes (observed cube `x`, result cube `out`; complex64 or float32): element tl + TPL*q of the unit's rows
    const unsigned cube_slice_bytes = (unsigned)a.n1 * (unsigned)N * ES;                          // at most 128 MiB
    const unsigned lane_c = ((unsigned)sub * (unsigned)N + colpart) * ES;
    const unsigned qc = (TPL >= 64 ? 64u * WPL : (unsigned)TPL) * ES;
    auto cube_soff = [&](const Where& w) -> unsigned { return (w.row * (unsigned)(RPW * N) + (unsigned)wsub * 64u) * ES; };
    auto issue_cube = [&](raw64 (&dst)[PPT], const Where& w) {   // PIPE_FIRST: the unit's samples of the observed cube
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.x) + (size_t)w.slice * cube_slice_bytes, cube_slice_bytes);
        unsigned so = cube_soff(w);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (DT == 0) dst[q] = buf_load_raw64(srd, lane_c, so);
            else dst[q] = (raw64)__builtin_amdgcn_raw_buffer_load_b32(srd, (int)lane_c, (int)so, 0);
            so += qc;
        }
    };
    auto store_cube = [&](const c32 (&src)[PPT], const Where& w) {   // PIPE_LAST: the unit's samples of the result (np.real for float32 cubes, POCS.py:656)
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.out) + (size_t)w.slice * cube_slice_bytes, cube_slice_bytes);
        unsigned so = cube_soff(w);
        const unsigned vo = w.on ? lane_c : BUF_OOB;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const c32 val = w.zero ? c32{0.f, 0.f} : src[q];
            if (DT == 0) buf_store_c32(srd, vo, so, val);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val.x), srd, (int)vo, (int)so, 0);
            so += qc;
        }
    };
    // per-row sums of |x|: [nslices][n1] doubles (< 2 GiB: nslices <= 65535, n1 <= 4096); a null table swallows the stores
    const __amdgpu_buffer_rsrc_t sums_srd = buf_srd(a.sums, a.sums != nullptr ? (unsigned)a.nslices * (unsigned)a.n1 * 8u : 0u);
    const float w_obs = 1.0f - a.alpha * 1.0f;   // POCS.py:616 at an observed trace

    const unsigned step = gridDim.x * UPB;
    // Workgroups b, b + 8, b + 16 ... share an XCD (round-robin dispatch: speed only, never correctness).  Give the workgroups of one
    // XCD ADJACENT row groups, so that what they store to a column block at about the same time is one contiguous run in one L2.
    unsigned wg = blockIdx.x;
#if P3D_PIPE64_XCD
    if (gridDim.x % 8 == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    unsigned g = wg * UPB + uline;
    Where cur = locate(g);
    c32 v[PPT];
    raw64 bx[PPT], by[PPT];
    unsigned long long mw_cur[PPT];   // the row's mask words stay in scalar registers from the request of its observed samples to its re-insertion
    unsigned cbs_cur[PPT];            // PIPE_FIRST: where the row's observed samples go in the compact array
    // per-row sum of |x| -> sums[slice][row] (one row = SEG consecutive lanes; the wavefronts of a long row in row_kernel's order)
    auto store_row_sum = [&](float acc, const Where& w) {
        double ws = (double)acc;
        constexpr int SEG = TPL >= 64 ? 64 : TPL;
#pragma unroll
        for (int o = SEG / 2; o > 0; o >>= 1) ws += __shfl_down(ws, o, SEG);
        const unsigned so = (w.slice * (unsigned)a.n1 + w.row * RPW) * 8u;
        const bool wr = w.on && !w.zero;
        if constexpr (WAVE) {
            buf_store_f64(sums_srd, ((lane & (SEG - 1)) == 0 && wr) ? (unsigned)sub * 8u : BUF_OOB, so, ws);
        } else {
            __syncthreads();
            if (lane == 0) red[wave] = ws;
            __syncthreads();
            double t = 0.0;
            for (int i = 0; i < WPL; ++i) t += red[uline * WPL + i];
            buf_store_f64(sums_srd, (wsub == 0 && lane == 0 && wr) ? 0u : BUF_OOB, so, t);
        }
    };
    // the LDS / twiddle addresses of the transforms are functions of tl alone; hoisted out of the loop they pin a dozen vector
    // registers across it, which is what pushes the kernel over the 128 a 16-wave workgroup may use (and ONE spilled register is
    // a scratch load, i.e. a vmcnt(0) in the middle of the transform).  Recomputed per row instead.
    auto fresh_tl = [&]() -> int { int t = tl; asm volatile("" : "+v"(t)); return t; };

    if constexpr (PM == PIPE_FIRST) {
        // ---- first pass of a job: observed cube -> (compact samples, sum |x_obs|) and forward row transform -> work buffer ----
        const bool tables = k_bits != nullptr && k_cbase != nullptr && a.xc != nullptr;   // uniform for the launch
        const __amdgpu_buffer_rsrc_t none = buf_srd(nullptr, 0u);
        issue_cube(bx, cur);
        if (tables) obs_tables(mw_cur, cbs_cur, cur);
        {   // as many (dropped) stores as one trip of the loop issues: exact wait counts at the loop header
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = c32{0.f, 0.f};
            store_work(v, cur, false);
#pragma unroll
            for (int q = 0; q < PPT; ++q) buf_store_c32(none, BUF_OOB, 0u, v[q]);
            buf_store_f64(sums_srd, BUF_OOB, 0u, 0.0);
        }
        for (unsigned g0 = wg * UPB; g0 < total; g0 += step) {
            const Where nxt = locate(g + step);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = DT == 0 ? raw_c32(bx[q]) : c32{__uint_as_float((unsigned)bx[q]), 0.f};
            __builtin_amdgcn_sched_barrier(0);
            float acc = 0.f;
            {   // compact copy of the observed samples (the order is a convention with the later passes: RowArgs::cbase + the rank
                // of the lane among the set lanes of its word); a non-zero sample at a trace the mask calls missing raises `violation`
                const __amdgpu_buffer_rsrc_t xsrd = tables ? buf_srd(reinterpret_cast<const char*>(a.xc) + (size_t)cur.slice * a.nobs * ES, a.nobs * ES) : none;
                bool bad = false;
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    const unsigned long long mw = tables ? mw_cur[q] : 0ull;
                    const bool set = __builtin_amdgcn_inverse_ballot_w64(mw);
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mw >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mw, 0u));
                    const unsigned vo = (set && cur.on) ? rank * ES : BUF_OOB;
                    const unsigned so = tables ? cbs_cur[q] * ES : 0u;
                    if (DT == 0) buf_store_c32(xsrd, vo, so, v[q]);
                    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].x), xsrd, (int)vo, (int)so, 0);
                    bad = bad || (!set && (v[q].x != 0.f || v[q].y != 0.f));
                    acc += abs_c32(v[q]);
                }
                if (tables && cur.on && __any(bad)) {   // (rare; an atomic older than every load a later wait covers)
                    if (lane == 0) atomicOr(a.violation, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            store_row_sum(acc, cur);
            __builtin_amdgcn_sched_barrier(0);
            issue_cube(bx, nxt);
            unsigned long long mw_nxt[PPT];
            unsigned cbs_nxt[PPT];
            if (tables) obs_tables(mw_nxt, cbs_nxt, nxt);
            __builtin_amdgcn_sched_barrier(0);
            line_fft<N, FWD, WAVE>(v, lds, tw, fresh_tl());
            __builtin_amdgcn_sched_barrier(0);
#if P3D_PIPE64_LOCKSTEP
            if (WAVE && RPW == 1) __builtin_amdgcn_s_barrier();
#endif
            store_work(v, cur, cur.on);
            __builtin_amdgcn_sched_barrier(0);
            if (tables) {
#pragma unroll
                for (int q = 0; q < PPT; ++q) { mw_cur[q] = mw_nxt[q]; cbs_cur[q] = cbs_nxt[q]; }
            }
            g += step;
            cur = nxt;
        }
        return;
    }

    issue_work(by, cur);
    {
        unsigned cbs0[PPT];
        obs_tables(mw_cur, cbs0, cur);
        issue_obs_with(bx, cur, mw_cur, cbs0);
    }
    // TS: after the lock-step barrier lane l of wave w reads, for j = 0 ... 7, the columns 8 (8 w + j) + 2 (l & 3), + 1 of row l >> 2
    // from that row's buffer and stores them as bytes 16 l ... 16 l + 15 of the 1-KiB run of column block 8 w + j
    const unsigned ts_row = (unsigned)lane >> 2;
    const c32* const ts_src = data + ts_row * LSTR + (wave * 64 + 2 * (lane & 3)) + ((wave * 64) >> 4);   // + 8 j + (j >> 1): below
    auto store_transposed = [&](const c32 (&src)[PPT], const Where& w, bool really) {
        {   // own row -> its buffer, canonical positions
            c32* const rowp = lds.ptr(lane);
#pragma unroll
            for (int q = 0; q < PPT; ++q) rowp[LdsRow::rel(64 * q)] = src[q];
        }
        __syncthreads();   // (all sixteen rows are in LDS)
        c32 ta[8], tb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // columns 8 j + 2 (l & 3) of the wave's 64: one padding slot per 16 columns
            const c32* const sp = ts_src + 8 * j + (j >> 1);
            ta[j] = sp[0];
            tb[j] = sp[1];
        }
        __syncthreads();   // (everybody has what it needs: the buffers are free for the next row's transforms)
        // the sixteen rows of a workgroup are g0 ... g0 + 15 of ONE slice (n1 % 16 == 0): row block and validity are workgroup-uniform
        const __amdgpu_buffer_rsrc_t srd = work_srd(w);
        const unsigned row0 = w.row - (unsigned)uline;
        unsigned so = row0 * 64u + (unsigned)(wave * 8) * (wblk * 8u);
        const unsigned vo = (really && !P3D_ABL_NOSTORE) ? (unsigned)lane * 16u : BUF_OOB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            buf_store_2c32(srd, vo, so, ta[j], tb[j]);
            so += wblk * 8u;
        }
    };
    {   // as many stores as one trip of the loop issues, all out of range: the wait counts at the loop header are then the same
        // along both edges into it (see the note above the kernel)
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = c32{0.f, 0.f};
        if constexpr (TS) {
            const __amdgpu_buffer_rsrc_t none = buf_srd(nullptr, 0u);
#pragma unroll
            for (int j = 0; j < 8; ++j) buf_store_2c32(none, BUF_OOB, 0u, v[0], v[1]);
        } else {
            store_work(v, cur, false);
        }
        buf_store_f64(sums_srd, BUF_OOB, 0u, 0.0);
    }
#if P3D_STAMPS
    unsigned st_acc[STAMP_PHASES] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif
    for (unsigned g0 = wg * UPB; g0 < total; g0 += step) {
        const Where nxt = locate(g + step);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = raw_c32(by[q]);
#if P3D_STAMPS
#pragma unroll
        for (int q = 0; q < PPT; ++q) asm volatile("" : "+v"(v[q].x), "+v"(v[q].y));
#endif
        P3D_STAMP(0);   // wait for the row's elements of the work buffer
        __builtin_amdgcn_sched_barrier(0);
        const int tl_r = fresh_tl();
        if (!P3D_ABL_NOFFT) line_fft<N, INV, WAVE>(v, lds, tw, tl_r);
        P3D_STAMP(1);   // inverse transform
        __builtin_amdgcn_sched_barrier(0);
        // the samples are first touched HERE: without this the compiler starts on bx * alpha in the middle of the transform and
        // waits for the loads there
#pragma unroll
        for (int q = 0; q < PPT; ++q) asm volatile("" : "+v"(bx[q]));
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            c32 xn = v[q] * a.scale;
            const float w = __builtin_amdgcn_inverse_ballot_w64(mw_cur[q]) ? w_obs : 1.0f;
            xn = axpby(xn, w, raw_c32(bx[q]), a.alpha);        // POCS.py:616-619
            acc += abs_c32(xn);
            v[q] = xn;
        }
#if P3D_STAMPS
        asm volatile("" : "+v"(acc));
#endif
        P3D_STAMP(2);   // wait for the observed samples, re-insertion
        __builtin_amdgcn_sched_barrier(0);
        if (!P3D_ABL_NOSUMS) store_row_sum(acc, cur);
        P3D_STAMP(3);   // sum of |x|
        __builtin_amdgcn_sched_barrier(0);
        // The next row's elements of the work buffer, and the scalar tables its observed samples are found with, are requested
        // BEFORE the forward transform: all waves of a workgroup run in step, so a latency nobody computes behind is a latency the
        // whole CU waits for.
        issue_work(by, nxt);
        unsigned long long mw_nxt[PPT];
        unsigned cbs_n[PPT];
        obs_tables(mw_nxt, cbs_n, nxt);
        P3D_STAMP(4);   // requests for the next row's work-buffer elements (scalar tables first)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PM == PIPE_MID) {
            if (!P3D_ABL_NOFFT) line_fft<N, FWD, WAVE>(v, lds, tw, tl_r);
        }
        P3D_STAMP(5);   // forward transform
        __builtin_amdgcn_sched_barrier(0);
        issue_obs_with(bx, nxt, mw_nxt, cbs_n);
        P3D_STAMP(6);   // requests for the next row's observed samples
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PM == PIPE_MID) {
            // Adjacent rows share the 128-byte lines of the work buffer (64 bytes each), and sixteen adjacent rows make one contiguous
            // KiB per column block: the waves of a workgroup store TOGETHER.  Measured on the headline cube (profiles/r02_rowpass_
            // schedule.txt): barrier every row 1.60 ms, every 2nd / 4th / 8th / 32nd row 1.72 / 1.92 / 2.04 / 2.11 ms, never 2.36 ms;
            // lock-step kept by groups of 2 / 4 / 8 waves only (counters in LDS) 1.74 / 1.80 / 1.71 ms.  A wavefront that holds two or
            // more adjacent rows (RPW > 1) pairs their halves up by itself.
            if constexpr (TS) {
                P3D_STAMP(7);
                store_transposed(v, cur, cur.on);   // (its two barriers keep the rows in step)
            } else {
#if P3D_PIPE64_LOCKSTEP
                if (WAVE && RPW == 1) __builtin_amdgcn_s_barrier();
#endif
                P3D_STAMP(7);   // lock-step barrier
                store_work(v, cur, cur.on);
            }
            P3D_STAMP(8);   // issue of the stores
        } else {
            store_cube(v, cur);   // last pass of a job: whole rows of the result cube, no neighbour to wait for
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < PPT; ++q) mw_cur[q] = mw_nxt[q];
        g += step;
        cur = nxt;
    }
#if P3D_STAMPS
    if (PM == PIPE_MID && lane == 0 && blockIdx.x < 1024 && wave < 16) {   // (the steady state only: the last pass runs after it)
#pragma unroll
        for (int i = 0; i < STAMP_PHASES; ++i) p3d_stamp_buf[((size_t)blockIdx.x * 16 + wave) * STAMP_PHASES + i] = st_acc[i];
    }
#endif
}

// =================================================================================================
// real cubes (float32, time domain): two rows per complex transform, half-spectrum work buffer
// =================================================================================================
// x real => fft2(x) is Hermitian, and the hard threshold (a function of |X| alone) keeps it so: columns 0 ... N/2 of the row
// transforms carry everything.  Rows 2p and 2p + 1 go through ONE complex transform, z = r_a + i r_b:
//     R_a[k] = (Z[k] + conj Z[N-k]) / 2,   R_b[k] = (Z[k] - conj Z[N-k]) / (2 i),   k = 0 ... N/2,
// and back: Z[k] = R_a[k] + i R_b[k], Z[N-k] = conj R_a[k] + i conj R_b[k].  The work buffer holds N/2 + 1 columns (the same
// column-blocked layout, 65 blocks at N = 1024), the column pass is the complex one on half the columns, and a wavefront of this
// pass owns a row pair: half the transforms, half the bytes of the complex path per row.  Z[N-k] sits in lane 64 - tl, register
// 15 - q: one cross-lane read per stored element.  The element-wise work (scale, re-insertion, sums) is the arithmetic of the
// complex path on the real parts; the imaginary part the reference carries along for a real cube is rounding noise (POCS.py:656
// returns the real part) and is dropped here every iteration instead of once at the end.
enum RealMode { REAL_FIRST = 0, REAL_MID = 1, REAL_LAST = 2 };

template <int N, int MODE, bool SPARSE>
__global__ __launch_bounds__((pipe64_threads<N>()), 4) void row_real_kernel(const RowArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    static_assert((64 % TPL == 0 || TPL % 64 == 0) && TPL >= 8 && TPL <= 256 && PPT == 16, "whole row pairs per wavefront, or whole wavefronts per pair");
    constexpr int WPL = TPL >= 64 ? TPL / 64 : 1; // wavefronts per row pair (rows of 2048 / 4096 samples: 2 / 4)
    constexpr bool WAVE = WPL == 1;
    constexpr int PW = TPL >= 64 ? 1 : 64 / TPL;  // row pairs per wavefront: lanes [sub * TPL, (sub + 1) * TPL) hold rows a = PW * 2u + sub
                                                  // and b = a + PW (so that the a-rows and the b-rows of a wave are each one unit of the
                                                  // lane-mask tables of the complex pass)
    constexpr int THREADS = pipe64_threads<N>();
    constexpr int UPB = THREADS / 64 / WPL;       // units (2 * PW rows) per workgroup
    constexpr int LSTR = LdsRow::stride(N);
    constexpr int HQ = PPT / 2;                   // registers 0 ... HQ-1 hold columns < N/2; register HQ of lane 0 holds column N/2
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + PassTables<N>::slots();
    const TwOrdered tw{twl};

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int uline = wave / WPL, wsub = wave % WPL;
    const int lane = tid & 63, tl = TPL >= 64 ? wsub * 64 + lane : lane % TPL, sub = TPL >= 64 ? 0 : lane / TPL;
    for (int i = tid; i < PassTables<N>::slots(); i += THREADS) twl[i] = a.tw[i];
    __syncthreads();
    const LdsRow lds{data + (uline * PW + sub) * LSTR};
    double* red = reinterpret_cast<double*>(data + UPB * PW * LSTR);   // per-wave partial sums of pairs that span waves

    const unsigned pps = (unsigned)a.n1 / (2 * PW);            // units per slice
    const unsigned total = (unsigned)a.nslices * pps;
    const unsigned wblk = (unsigned)a.n1 * 8;
    const size_t wstride = wk_slice_stride(a.n1, N / 2 + 1);
    const unsigned mlane = (unsigned)(TPL - tl);               // column N - e sits (TPL - tl) columns into register 15 - q's run
    const unsigned lane_w = ((unsigned)(tl >> 3) * wblk + (unsigned)(tl & 7) + (unsigned)sub * 8u) * 8u;
    const unsigned lane_wm = ((mlane >> 3) * wblk + (mlane & 7) + (unsigned)sub * 8u) * 8u;
    const size_t qs64 = (size_t)(TPL / 8) * wblk * 8u;         // bytes from register q to q + 1
    constexpr unsigned BROW = PW * 64u;                        // row b = row a + PW: bytes inside a column block

    typedef const unsigned long long __attribute__((address_space(4))) * kmask_t;
    typedef const unsigned __attribute__((address_space(4))) * kuint_t;
    typedef const int __attribute__((address_space(4))) * kint_t;
    const kmask_t k_bits = (kmask_t)a.bits64, k_nzl = (kmask_t)a.nzl;
    const kuint_t k_cbase = (kuint_t)a.cbase;
    const kint_t k_done = (kint_t)a.done;
    auto opaque = [](unsigned o) -> unsigned { asm volatile("" : "+v"(o)); return o; };
    auto qstep = [&]() -> size_t { size_t qs = qs64; asm volatile("" : "+s"(qs)); return qs; };
    const float w_obs = 1.0f - a.alpha * 1.0f;

    const unsigned step = gridDim.x * UPB;
    for (unsigned u = blockIdx.x * UPB + uline, u0 = blockIdx.x * UPB; u0 < total; u += step, u0 += step) {
        if (MODE != REAL_FIRST) __syncthreads();   // lock-step: adjacent row pairs complete the 128-byte lines of a column block
        const bool in_range = u < total;
        const unsigned uu = in_range ? u : 0u;
        const unsigned slice = uu / pps, pr = uu - slice * pps, ua = 2 * pr;   // ua, ua + 1: the table units of the a- and b-rows
        const unsigned ra = ua * PW + (unsigned)sub;                           // this lane's row a
        int dn = 0;
        if (k_done != nullptr) dn = k_done[slice];
        bool on = in_range;
        if (MODE == REAL_MID) on = on && dn == 0;
        if (MODE == REAL_LAST) on = on && (a.only_done ? dn == a.only_done : dn <= 0);
        if (MODE == REAL_FIRST) on = on && dn == 0;
        char* const wb = reinterpret_cast<char*>(a.work) + (slice * wstride + (size_t)ua * PW * 8) * 8;   // the unit's first row
        const size_t xrow = ((size_t)slice * a.n1 + ra) * N;                                               // row-major cubes, row a
        // Mask words and compact bases of the a-rows (unit ua) and the b-rows (unit ua + 1): tables of the complex pass.  They are
        // (re)loaded where they are used, one unit at a time -- 16 x (64 + 32) bits per unit; all four sets at once do not fit the
        // scalar registers and every use would then be a v_readlane from a spill lane.
        auto words_of = [&](unsigned unit) -> kmask_t { kmask_t m = k_bits + pipe64_word(unit, WPL, wsub, 0); asm volatile("" : "+s"(m)); return m; };
        auto bases_of = [&](unsigned unit) -> kuint_t { kuint_t c = k_cbase + pipe64_word(unit, WPL, wsub, 0); asm volatile("" : "+s"(c)); return c; };
        const char* const xcb = reinterpret_cast<const char*>(a.xc) + (size_t)slice * a.nobs * 4u;
        c32 v[PPT];
        float oa[PPT], ob[PPT];

        if (MODE == REAL_FIRST) {
            // ---- the observed rows themselves; their compact copy for the later passes ----
            const float* const x = reinterpret_cast<const float*>(a.x) + xrow;
            bool bad = false;
            float sa = 0.f, sb = 0.f;
            float* const xc = reinterpret_cast<float*>(a.xc) + (size_t)slice * a.nobs;
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // the a-rows, then the b-rows
                const kmask_t mw = words_of(ua + h);
                const kuint_t cw = bases_of(ua + h);
                const float* const xr = x + (size_t)h * PW * N;
                float sh = 0.f;
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    const unsigned long long m = mw[q];
                    const float xv = xr[tl + TPL * q];
                    const bool set = __builtin_amdgcn_inverse_ballot_w64(m);
                    const unsigned rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (on && a.xc != nullptr && set) xc[cw[q] + rk] = xv;
                    bad = bad || (!set && xv != 0.f);
                    sh += fabsf(xv);
                    if (h == 0) v[q].x = xv; else v[q].y = xv;
                }
                if (h == 0) sa = sh; else sb = sh;
            }
            if (on && bad && a.violation != nullptr) atomicOr(a.violation, 1);
            if (a.sums != nullptr) {
                double da = (double)sa, db = (double)sb;
                constexpr int SEG = TPL >= 64 ? 64 : TPL;
#pragma unroll
                for (int o = SEG / 2; o > 0; o >>= 1) { da += __shfl_down(da, o, SEG); db += __shfl_down(db, o, SEG); }
                if constexpr (WAVE) {
                    if (tl == 0 && on) { a.sums[(size_t)slice * a.n1 + ra] = da; a.sums[(size_t)slice * a.n1 + ra + PW] = db; }
                } else {   // the wavefronts of a pair, in order
                    __syncthreads();
                    if (lane == 0) { red[2 * wave] = da; red[2 * wave + 1] = db; }
                    __syncthreads();
                    if (tl == 0 && on) {
                        double ta = 0.0, tb = 0.0;
                        for (int w = 0; w < WPL; ++w) { ta += red[2 * (uline * WPL + w)]; tb += red[2 * (uline * WPL + w) + 1]; }
                        a.sums[(size_t)slice * a.n1 + ra] = ta;
                        a.sums[(size_t)slice * a.n1 + ra + 1] = tb;
                    }
                }
            }
        } else {
            // ---- half spectra of the two rows -> Z = R_a + i R_b on all N columns ----
            const kmask_t nz = k_nzl + pipe64_word(slice, WPL, wsub, 0);
            unsigned long long nzw[PPT];
            if (SPARSE) {
#pragma unroll
                for (int q = 0; q < PPT; ++q) nzw[q] = nz[q];
            }
            const size_t qs = qstep();
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const bool mirror = q >= HQ;
                const char* const b = wb + (size_t)(mirror ? PPT - 1 - q : q) * qs;
                c32 r0{0.f, 0.f}, r1{0.f, 0.f};
                if (!SPARSE || __builtin_amdgcn_inverse_ballot_w64(nzw[q])) {
                    const unsigned o = opaque(mirror ? lane_wm : lane_w);
                    r0 = *reinterpret_cast<const c32*>(b + o);
                    r1 = *reinterpret_cast<const c32*>(b + BROW + o);
                }
                if (mirror) { r0.y = -r0.y; r1.y = -r1.y; }
                if ((q == 0 || q == HQ) && tl == 0) { r0.y = 0.f; r1.y = 0.f; }   // columns 0 and N/2 of a real row are real
                v[q] = c32{r0.x - r1.y, r0.y + r1.x};
            }
            // observed samples of both rows (compact, float), requested before the transform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const kmask_t mw = words_of(ua + h);
                const kuint_t cw = bases_of(ua + h);
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    const unsigned long long m = mw[q];
                    const unsigned rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    float f = 0.f;
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) f = *reinterpret_cast<const float*>(xcb + (size_t)cw[q] * 4u + opaque(rk * 4u));
                    if (h == 0) oa[q] = f; else ob[q] = f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            line_fft<N, INV, WAVE>(v, lds, tw, tl);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PPT; ++q) asm volatile("" : "+v"(oa[q]), "+v"(ob[q]));
            float sa = 0.f, sb = 0.f;
            const bool handback = MODE == REAL_LAST && a.only_done != 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const kmask_t mw = words_of(ua + h);
                float sh = 0.f;
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    const bool set = __builtin_amdgcn_inverse_ballot_w64(mw[q]);
                    const float xo = h == 0 ? oa[q] : ob[q];
                    float xv = (h == 0 ? v[q].x : v[q].y) * a.scale;
                    if (handback) {   // the converged iterate up to one row-transform round trip; an observed trace with alpha = 1 IS the observation
                        if (a.alpha == 1.0f && set) xv = xo;
                    } else {
                        xv = __builtin_fmaf(xv, set ? w_obs : 1.0f, xo * a.alpha);   // POCS.py:616-619
                    }
                    sh += fabsf(xv);
                    if (h == 0) v[q].x = xv; else v[q].y = xv;
                }
                if (h == 0) sa = sh; else sb = sh;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (a.sums != nullptr) {
                double da = (double)sa, db = (double)sb;
                constexpr int SEG = TPL >= 64 ? 64 : TPL;
#pragma unroll
                for (int o = SEG / 2; o > 0; o >>= 1) { da += __shfl_down(da, o, SEG); db += __shfl_down(db, o, SEG); }
                if constexpr (WAVE) {
                    if (tl == 0 && on) { a.sums[(size_t)slice * a.n1 + ra] = da; a.sums[(size_t)slice * a.n1 + ra + PW] = db; }
                } else {   // the wavefronts of a pair, in order
                    __syncthreads();
                    if (lane == 0) { red[2 * wave] = da; red[2 * wave + 1] = db; }
                    __syncthreads();
                    if (tl == 0 && on) {
                        double ta = 0.0, tb = 0.0;
                        for (int w = 0; w < WPL; ++w) { ta += red[2 * (uline * WPL + w)]; tb += red[2 * (uline * WPL + w) + 1]; }
                        a.sums[(size_t)slice * a.n1 + ra] = ta;
                        a.sums[(size_t)slice * a.n1 + ra + 1] = tb;
                    }
                }
            }
            if (MODE == REAL_LAST) {
                if (on) {
                    float* const o = reinterpret_cast<float*>(a.out) + xrow;
                    if (dn < 0) {   // all-zero slice is handed back untouched (POCS.py:515-521)
#pragma unroll
                        for (int q = 0; q < PPT; ++q) { o[tl + TPL * q] = 0.f; o[(size_t)PW * N + tl + TPL * q] = 0.f; }
                    } else {
#pragma unroll
                        for (int q = 0; q < PPT; ++q) { o[tl + TPL * q] = v[q].x; o[(size_t)PW * N + tl + TPL * q] = v[q].y; }
                    }
                }
                continue;
            }
        }

        // ---- forward transform of z = r_a + i r_b, split into the two half spectra, store ----
        __builtin_amdgcn_sched_barrier(0);
        line_fft<N, FWD, WAVE>(v, lds, tw, tl);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == REAL_FIRST) __syncthreads();   // (first pass: keep the stores of adjacent pairs together as well)
        {
            const size_t qs = qstep();
            const int src = sub * TPL + ((TPL - tl) % TPL);
            if constexpr (!WAVE) {   // the partner may sit in another wavefront: the upper half of Z goes through the pair's LDS row
                __syncthreads();
#pragma unroll
                for (int q = HQ; q < PPT; ++q) lds.at(tl + TPL * q) = v[q];
                __syncthreads();
            }
            char* b = wb;
#pragma unroll
            for (int q = 0; q <= HQ; ++q) {
                const c32 z = v[q];
                const c32 far = v[q < HQ ? PPT - 1 - q : HQ - 1];      // lanes > 0: Z[N - e] is register 15 - q of lane 64 - tl
                c32 pz;
                if constexpr (WAVE) pz = c32{__shfl(far.x, src, 64), __shfl(far.y, src, 64)};
                else pz = lds.at(tl == 0 ? N / 2 : N - (tl + TPL * q));   // (tl = 0 is overwritten below)
                if (tl == 0) pz = q == 0 ? v[0] : v[PPT - q];          // tl = 0: Z[N - TPL q] is its own register 16 - q (q = 0: Z[0])
                const c32 Ra{0.5f * (z.x + pz.x), 0.5f * (z.y - pz.y)};
                const c32 Rb{0.5f * (z.y + pz.y), -0.5f * (z.x - pz.x)};
                if (on && (q < HQ || tl == 0)) {
                    const unsigned o = opaque(lane_w);
                    *reinterpret_cast<c32*>(b + o) = Ra;
                    *reinterpret_cast<c32*>(b + BROW + o) = Rb;
                }
                b += qs;
            }
        }
    }
}

// =================================================================================================
// spectrum (column) pass
// =================================================================================================
__device__ __forceinline__ bool lex_greater(float ar, float ai, float br, float bi)
{
    return (ar > br) || (ar == br && ai > bi);
}

// T columns per workgroup; CW = min(T, 8) of them share a 64-byte column block.
template <int N, int T, int MODE>
__global__ __launch_bounds__(T* Plan<N>::TPL, (T * Plan<N>::TPL >= 1024 ? 4 : P3D_WAVES_PER_EU)) void col_kernel(const ColArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    constexpr int THREADS = T * TPL;
    constexpr int CW = T < 8 ? T : 8;
    using LDS = LdsColW<CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + ColTables<N>::slots();
    const TwCol tw{twl};

    const int tid = threadIdx.x;
    const int c_lo = tid % CW;
    const int tl = (tid / CW) % TPL;
    const int cbl = tid / (CW * TPL);  // column block of this thread inside the tile
    const int slice = blockIdx.y;
    // Tiles narrower than a 64-byte column block (long lines): the 8/T tiles of one block are given to workgroups g, g+8, ...,
    // which the dispatcher places on the same XCD one after the other, so that the block's cache lines are fetched from HBM
    // once and the other pieces hit that XCD's L2 (workgroup g of a 2-D grid runs on XCD g % 8 when gridDim.x % 8 == 0).
    int tile = blockIdx.x;
    if constexpr (T < 8) {
        constexpr int G = 8 / T;
        if (P3D_XCD_PAIR && (gridDim.x % (8 * G)) == 0) {
            const int xcd = tile & 7, j = tile >> 3;
            tile = ((j / G) * 8 + xcd) * G + (j % G);
        }
    }
    const int col = tile * T + cbl * CW + c_lo;
    const bool valid = col < a.n2;

    if (a.done && a.done[slice] != 0) return;

    for (int i = tid; i < ColTables<N>::slots(); i += THREADS) twl[i] = a.tw[i];
    __syncthreads();

    const LDS lds{data + cbl * LDS::stride(N) + c_lo};
    constexpr bool ITER = MODE == COL_ITER || MODE == COL_ITER_SOFT || MODE == COL_ITER_GARROTE;
    const int vcol = valid ? col : 0;
    // wave-uniform slice bases + 32-bit element offsets (see row_kernel)
    const c32* const inb = a.in + (size_t)slice * (a.in_std ? (size_t)N * a.n2 : wk_slice_stride(N, a.n2));
    c32* const outb = a.out + (size_t)slice * (a.out_std ? (size_t)N * a.n2 : wk_slice_stride(N, a.n2));
    const unsigned blk0 = ((unsigned)(vcol >> 3) * N) * 8 + (vcol & 7);  // column-blocked: + row*8
    // element offset of (row r, this thread's column) = origin + r * pitch, both picked ONCE per layout (a select per element
    // costs the sixteen loads and stores of a thread 60 vector instructions)
    // (the iteration itself always works on the column-blocked buffer: compile-time pitch, the q-dependent part of an address
    // becomes an instruction immediate or one add)
    const bool in_std = !ITER && a.in_std, out_std = !ITER && a.out_std;
    const unsigned in_org = in_std ? (unsigned)vcol : blk0, in_pitch = in_std ? (unsigned)a.n2 : 8u;
    const unsigned out_org = out_std ? (unsigned)vcol : blk0, out_pitch = out_std ? (unsigned)a.n2 : 8u;
    auto eoff = [&](int std_layout, int r) -> unsigned {
        return std_layout ? (unsigned)r * a.n2 + vcol : blk0 + (unsigned)r * 8;
    };
    c32 v[PPT];
    if (MODE == COL_SHRINK) {
        // coefficients of shearlet s of slice b (grid.y = b*nsh + s): back to the space domain, threshold (POCS.py:598 with a
        // per-shearlet tau), forward again; in place on the work buffer
        const int b = slice / a.sh.nsh, s = slice - b * a.sh.nsh;
        const c32 tau = a.sh.tau[((size_t)b * a.sh.niter + a.sh.iter) * a.sh.nsh + s];
        const float scale = 1.0f / ((float)N * (float)a.n2);
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = inb[eoff(0, tl + TPL * q)];
        line_fft<N, INV, false>(v, lds, tw, tl);
        const Shrink shr(tau, a.sh.op);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            c32 c = v[q] * scale;
            if (a.sh.real_only) c.y = 0.f;   // FFST returns the real part for real data
            v[q] = shr(c);
        }
        line_fft<N, FWD, false>(v, lds, tw, tl);
        if (valid) {
#pragma unroll
            for (int q = 0; q < PPT; ++q) outb[eoff(0, tl + TPL * q)] = v[q];
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < PPT; ++q)   // scalar base + 32-bit byte offset (a slice is far below 4 GiB); columns past the edge re-read column 0
        v[q] = *reinterpret_cast<const c32*>(reinterpret_cast<const char*>(inb) + (in_org + (unsigned)(tl + TPL * q) * in_pitch) * 8u);

    if (MODE != COL_INV) line_fft<N, FWD, false>(v, lds, tw, tl);

    if (ITER || (MODE == COL_FWD && a.tau != nullptr)) {
        const c32 tau = a.tau[(size_t)slice * a.niter + a.iter];
        const int op = MODE == COL_ITER ? 0 : (MODE == COL_ITER_SOFT ? 1 : (MODE == COL_ITER_GARROTE ? 2 : a.op));
        const Shrink shr(tau, op);
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = shr(v[q]);
        if (ITER && a.nzflag != nullptr) {
            // Sparse spectra (the premise of the method): a tile the threshold emptied is all zeros after the inverse
            // transform too.  Say so instead of transforming and storing it; the row pass reads zeros for it.
            // any bit set in any coefficient (a kept -0.0 counts as kept: harmless, the tile is then simply processed)
            unsigned bits = 0;
#pragma unroll
            for (int q = 0; q < PPT; ++q) bits |= __float_as_uint(v[q].x) | __float_as_uint(v[q].y);
            const int kept = __syncthreads_or(bits != 0u ? 1 : 0);
            if (tid == 0) a.nzflag[(size_t)slice * gridDim.x + tile] = kept ? 1 : 0;
            if (!kept) {
                // The row pass skips whole 8-column BLOCKS.  A tile narrower than a block may be empty next to a sibling that
                // is not, and the row pass then reads this tile's columns too: they must hold the zeros, not last iteration's
                // values (the transform is still skipped).
                if constexpr (T < 8) {
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < PPT; ++q) {
                            unsigned o = (out_org + (unsigned)(tl + TPL * q) * out_pitch) * 8u;
                            asm volatile("" : "+v"(o));
                            *reinterpret_cast<c32*>(reinterpret_cast<char*>(outb) + o) = c32{0.f, 0.f};
                        }
                    }
                }
                return;
            }
        }
    }

    if (MODE == COL_STATS) {
        // lexicographic complex max, max|X|, min|X|, sum|X|^2 of this tile (POCS.py:261-262, 288, 299)
        float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY, sq = 0.f;
        if (valid) {
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const float p = v[q].x * v[q].x + v[q].y * v[q].y;
                if (lex_greater(v[q].x, v[q].y, lr, li)) { lr = v[q].x; li = v[q].y; }
                mx = fmaxf(mx, p);
                mn = fminf(mn, p);
                sq += p;
            }
        }
        // workgroups of short lines have fewer than 64 threads: never combine with an inactive lane
        const int lane = tid & 63;
        const int nact = (THREADS - (tid & ~63)) < 64 ? (THREADS - (tid & ~63)) : 64;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
            const float omx = __shfl_down(mx, o, 64), omn = __shfl_down(mn, o, 64), osq = __shfl_down(sq, o, 64);
            if (lane + o < nact) {
                if (lex_greater(orr, oi, lr, li)) { lr = orr; li = oi; }
                mx = fmaxf(mx, omx);
                mn = fminf(mn, omn);
                sq += osq;
            }
        }
        __syncthreads();  // LDS data region is free again
        float* red = reinterpret_cast<float*>(data);
        const int wave = tid >> 6, nw = (THREADS + 63) >> 6;
        if ((tid & 63) == 0) {
            red[wave * 5 + 0] = lr; red[wave * 5 + 1] = li; red[wave * 5 + 2] = mx;
            red[wave * 5 + 3] = mn; red[wave * 5 + 4] = sq;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nw; ++w) {
                if (lex_greater(red[w * 5], red[w * 5 + 1], lr, li)) { lr = red[w * 5]; li = red[w * 5 + 1]; }
                mx = fmaxf(mx, red[w * 5 + 2]);
                mn = fminf(mn, red[w * 5 + 3]);
                sq += red[w * 5 + 4];
            }
            float* p = a.partials + ((size_t)slice * gridDim.x + blockIdx.x) * STATS_PARTIAL;
            p[0] = lr; p[1] = li; p[2] = sqrtf(mx); p[3] = sqrtf(mn); p[4] = sq;
        }
        return;
    }

    if (ITER || MODE == COL_INV) line_fft<N, INV, false>(v, lds, tw, tl);

    if (valid) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            unsigned o = (out_org + (unsigned)(tl + TPL * q) * out_pitch) * 8u;
            asm volatile("" : "+v"(o));   // keeps the zero-extension inside this block ("scalar base + 32-bit offset" is matched per block)
            *reinterpret_cast<c32*>(reinterpret_cast<char*>(outb) + o) = v[q];
        }
    }
}

// =================================================================================================
// spectrum (column) pass, steady state: persistent, the next tile's loads in flight during the transforms
// =================================================================================================
// Same arithmetic as col_kernel<N, T, COL_ITER*> (bit for bit).  The one-launch form starts a workgroup per tile: every tile pays a
// workgroup launch, a copy of the twiddle tables into LDS (10 KiB at N = 1024) and a full load latency before its first butterfly,
// and 95 % of the tiles of a sparse spectrum end right after the threshold.  Here a workgroup stays on its CU (two per CU as
// before), copies the tables once, and requests tile t + 1 BEFORE it transforms tile t (16 more registers pairs per thread; the
// loads are issued ahead of the tile's stores, so waiting for them never waits for a store that is younger).
#ifndef P3D_COLPIPE_WAVES_PER_EU
#define P3D_COLPIPE_WAVES_PER_EU 4
#endif
// SHEAR: the column pass of a SHEARLET iteration instead (COL_SHRINK of col_kernel: inverse transform, 1/N, real part, threshold
// with the shearlet's own tau, forward transform; every tile is stored) -- `slice` then counts (slice, shearlet) pairs.
template <int N, int T, int OP, bool SHEAR = false>
__global__ __launch_bounds__(T* Plan<N>::TPL, P3D_COLPIPE_WAVES_PER_EU) void col_pipe_kernel(const ColArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT;
    constexpr int THREADS = T * TPL;
    static_assert(T % 8 == 0 && PPT == 16, "whole 64-byte column blocks per tile");
    constexpr int CW = 8;
    using LDS = LdsColW<CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + ColTables<N>::slots();
    const TwCol tw{twl};

    const int tid = threadIdx.x;
    const int c_lo = tid % CW;
    const int tl = (tid / CW) % TPL;
    const int cbl = tid / (CW * TPL);  // column block of this thread inside the tile
    for (int i = tid; i < ColTables<N>::slots(); i += THREADS) twl[i] = a.tw[i];
    __syncthreads();
    const LDS lds{data + cbl * LDS::stride(N) + c_lo};

    const unsigned tiles = (unsigned)(a.n2 + T - 1) / T;             // per slice
    const unsigned total = (unsigned)a.nslices * tiles;
    const size_t sstride = wk_slice_stride(N, a.n2);
    const unsigned slice_bytes = (unsigned)(sstride * 8);
    typedef const int __attribute__((address_space(4))) * kint_t;
    const kint_t k_done = (kint_t)a.done;
    typedef const unsigned long long __attribute__((address_space(4))) * ktau_t;
    const ktau_t k_tau = (ktau_t)(SHEAR ? a.sh.tau : a.tau);   // [nslices][niter] (SHEAR: [slice][niter][nsh]) float2, constant during the launch
    struct Tile { unsigned slice, tile; bool on; };
    auto locate = [&](unsigned g) -> Tile {
        Tile t;
        t.on = g < total;
        const unsigned gg = t.on ? g : 0u;
        t.slice = gg / tiles;
        t.tile = gg - t.slice * tiles;
        if (k_done != nullptr && t.on && k_done[t.slice] != 0) t.on = false;
        return t;
    };
    // element (row tl + TPL q, this thread's column) of the tile: byte offset inside the slice
    auto lane_off = [&](const Tile& t, bool& valid) -> unsigned {
        const int col = (int)t.tile * T + cbl * CW + c_lo;
        valid = col < a.n2;
        const int vcol = valid ? col : 0;
        return (((unsigned)(vcol >> 3) * N) * 8 + (vcol & 7) + (unsigned)tl * 8) * 8u;
    };
    auto issue = [&](raw64 (&dst)[PPT], const Tile& t) {
        bool valid;
        const unsigned vo = lane_off(t, valid);
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.in) + (size_t)t.slice * slice_bytes, slice_bytes);
#pragma unroll
        for (int q = 0; q < PPT; ++q) dst[q] = buf_load_raw64(srd, vo, (unsigned)(TPL * q) * 64u);
    };

    // Each workgroup takes a CONTIGUOUS run of tiles (on the headline cube: one whole slice, 8 MiB of consecutive addresses).  Handing
    // the tiles out with a stride of gridDim.x instead -- tile b, b + 512, ... -- costs 40 % (1.37 against 0.94 ms): with 128 tiles per
    // slice every workgroup then stays on ONE column block of every fourth slice, and the 512 concurrent streams sit 64 KiB apart.
    const unsigned per = (total + gridDim.x - 1) / gridDim.x;
    unsigned g = blockIdx.x * per;
    const unsigned g_end = g + per < total ? g + per : total;
    Tile cur = locate(g);
    raw64 nx[PPT];
    c32 v[PPT];
    issue(nx, cur);
#pragma unroll
    for (int q = 0; q < PPT; ++q) { v[q] = raw_c32(nx[q]); asm volatile("; first tile" : "+v"(v[q].x), "+v"(v[q].y)); }   // (nothing pending at the loop header)
    for (unsigned i = 0; i < per; ++i) {   // (the same trip count for every workgroup: the loop holds workgroup barriers)
        Tile nxt = locate(g + 1);
        if (g + 1 >= g_end) nxt.on = false;
        if (g >= g_end) cur.on = false;
        __builtin_amdgcn_sched_barrier(0);
        issue(nx, nxt);   // in flight during the transforms of this tile
        __builtin_amdgcn_sched_barrier(0);
        int tl_r = tl;
        asm volatile("" : "+v"(tl_r));   // (the transforms' LDS / twiddle addresses are recomputed per tile instead of living in registers across the loop)
        // (the threshold through the scalar path: a vector load here would sit BEHIND the sixteen loads of the next tile in the
        // in-order vmcnt queue, and waiting for it would wait for them)
        unsigned long long tau_bits;
        if constexpr (SHEAR) {
            const unsigned b = cur.slice / (unsigned)a.sh.nsh, sh = cur.slice - b * (unsigned)a.sh.nsh;
            tau_bits = k_tau[((size_t)b * a.sh.niter + a.sh.iter) * a.sh.nsh + sh];
        } else {
            tau_bits = k_tau[(size_t)cur.slice * a.niter + a.iter];
        }
        if constexpr (SHEAR) {
            line_fft<N, INV, false>(v, lds, tw, tl_r);
            const Shrink shr(c32{__uint_as_float((unsigned)tau_bits), __uint_as_float((unsigned)(tau_bits >> 32))}, a.sh.op);
            const float scale = 1.0f / ((float)N * (float)a.n2);
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                c32 c = v[q] * scale;
                if (a.sh.real_only) c.y = 0.f;   // FFST returns the real part for real data
                v[q] = shr(c);
            }
        } else {
            line_fft<N, FWD, false>(v, lds, tw, tl_r);
            const Shrink shr(c32{__uint_as_float((unsigned)tau_bits), __uint_as_float((unsigned)(tau_bits >> 32))}, OP);
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = shr(v[q]);
        }
        bool kept = true;
        if (!SHEAR && a.nzflag != nullptr) {   // a tile the threshold emptied is zeros after the inverse transform too: say so instead (see col_kernel)
            unsigned bits = 0;
#pragma unroll
            for (int q = 0; q < PPT; ++q) bits |= __float_as_uint(v[q].x) | __float_as_uint(v[q].y);
            kept = __syncthreads_or(bits != 0u ? 1 : 0) != 0;
            if (tid == 0 && cur.on) a.nzflag[(size_t)cur.slice * tiles + cur.tile] = kept ? 1 : 0;
        }
        // The hand-over of the next tile (v <- nx) is written out in BOTH arms, so that the compiler counts each arm by itself: behind
        // the sixteen stores of a kept tile the wait for the loads is vmcnt(16) -- they were issued first --, not the vmcnt(0) a join
        // of "stores or no stores" would force (dense spectra: 1.71 -> see profiles/r02_colpass_persistent.txt).
        if (kept) {   // workgroup-uniform
            if constexpr (SHEAR) line_fft<N, FWD, false>(v, lds, tw, tl_r);
            else line_fft<N, INV, false>(v, lds, tw, tl_r);
            bool valid;
            const unsigned vo = lane_off(cur, valid);
            const __amdgpu_buffer_rsrc_t osrd = buf_srd(reinterpret_cast<char*>(a.out) + (size_t)cur.slice * slice_bytes, slice_bytes);
            const unsigned so_v = (valid && cur.on) ? vo : BUF_OOB;
#pragma unroll
            for (int q = 0; q < PPT; ++q) buf_store_c32(osrd, so_v, (unsigned)(TPL * q) * 64u, v[q]);
            // (the empty asm statements "use" the values HERE: without them the copies are renamed away and the wait moves to the
            // first butterfly of the next trip -- behind the next issue of loads, where it covers the stores again)
#pragma unroll
            for (int q = 0; q < PPT; ++q) { v[q] = raw_c32(nx[q]); asm volatile("; kept tile" : "+v"(v[q].x), "+v"(v[q].y)); }
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q) { v[q] = raw_c32(nx[q]); asm volatile("; emptied tile" : "+v"(v[q].x), "+v"(v[q].y)); }
        }
        __syncthreads();   // the LDS image is free for the next tile
        g += 1;
        cur = nxt;
    }
}

// ---- launch helpers, one instantiation set per line length --------------------------------------
// columns per workgroup of the column pass: keep 512..1024 threads and <= ~80 KiB of LDS
template <int N>
constexpr int col_tile()
{
    return N >= 4096 ? 2 : (N >= 1024 ? 8 : (N == 512 ? 16 : (N == 256 ? 32 : 64)));  // N = 2048: 4-column tiles (2 WG/CU) measured slower (3.4 vs 2.6 ms)
}

template <int N>
constexpr size_t row_lds_bytes()
{
    return sizeof(c32) * (PassTables<N>::slots() + (ROW_THREADS / Plan<N>::TPL) * LdsRow::stride(N)) + 8 * sizeof(double);
}
template <int N, int MODE>
constexpr size_t row_lds_bytes_mode()
{
    return sizeof(c32) * (PassTables<N>::slots() + (row_threads<N, MODE>() / Plan<N>::TPL) * LdsRow::stride(N)) + 16 * sizeof(double);
}
template <int N>
constexpr size_t col_lds_bytes()
{
    constexpr int T = col_tile<N>();
    constexpr int CW = T < 8 ? T : 8;
    return sizeof(c32) * (ColTables<N>::slots() + (size_t)(T / CW) * LdsColW<CW>::stride(N));
}

template <class K>
inline hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int N, int MODE, bool BITS>
hipError_t launch_row_one(const RowArgs& a, hipStream_t st)
{
    constexpr int LB = row_threads<N, MODE>() / Plan<N>::TPL;
    const dim3 grid((a.n1 + LB - 1) / LB, a.nslices);
    constexpr size_t lds = row_lds_bytes_mode<N, MODE>();
    hipError_t e = allow_lds(row_kernel<N, MODE, BITS>, lds);
    if (e != hipSuccess) return e;
    row_kernel<N, MODE, BITS><<<grid, row_threads<N, MODE>(), lds, st>>>(a);
    return hipGetLastError();
}

// one launch of the wave-uniform persistent row pass (row_pipe64_kernel) in mode `pm` on a device with `cus` compute units;
// hipErrorNotSupported where the kernel does not apply (the caller falls back to the generic kernels)
template <int N>
hipError_t launch_row_pipe64(int pm, const RowArgs& a, int cus, hipStream_t st)
{
    if constexpr (Plan<N>::TPL >= 8 && Plan<N>::TPL <= 256 && Plan<N>::PPT == 16) {
        constexpr int RPW64 = Plan<N>::TPL >= 64 ? 1 : 64 / Plan<N>::TPL;
        if ((double)a.nslices * (double)wk_slice_stride(a.n1, N) >= 4294967296.0 || a.n1 % RPW64 != 0 || cus < 1) return hipErrorNotSupported;
        if (a.adaptive || a.write_out || a.only_done || a.plain) return hipErrorNotSupported;   // APOCS, per-iteration store, finalize, fft2 hook
        const bool tables = a.bits64 != nullptr && a.cbase != nullptr;
        if (pm == PIPE_FIRST) {
            if (a.x == nullptr || (a.xc != nullptr && !tables)) return hipErrorNotSupported;   // (no tables: the transform only)
        } else {
            if (a.bits == nullptr || a.xc == nullptr || !tables) return hipErrorNotSupported;
        }
        constexpr int LB64 = pipe64_rows<N>();
        constexpr size_t lds64 = pipe64_lds_bytes<N>();
        constexpr int UPB64 = LB64 / RPW64;
        const long groups64 = ((long)a.nslices * (a.n1 / RPW64) + UPB64 - 1) / UPB64;
        int per_cu64 = (int)((160 * 1024) / lds64);            // workgroups a CU holds: LDS ...
        const int by_waves = 16 / (pipe64_threads<N>() / 64);  // ... and 4 waves per SIMD
        if (per_cu64 > by_waves) per_cu64 = by_waves;
        if (per_cu64 < 1) per_cu64 = 1;
        const long wgs64 = (long)cus * per_cu64;
        const dim3 grid64((unsigned)(groups64 < wgs64 ? groups64 : wgs64));
        hipError_t e = hipSuccess;
#define P3D_PIPE64(DT, SP, PM)                                                                                  \
    do {                                                                                                        \
        if ((e = allow_lds(row_pipe64_kernel<N, DT, SP, PM>, lds64)) != hipSuccess) return e;                   \
        row_pipe64_kernel<N, DT, SP, PM><<<grid64, pipe64_threads<N>(), lds64, st>>>(a);                        \
    } while (0)
        const bool sp = a.nzl != nullptr;
        if (pm == PIPE_FIRST) {
            if (a.dtype == 0) P3D_PIPE64(0, false, PIPE_FIRST); else P3D_PIPE64(1, false, PIPE_FIRST);
        } else if (pm == PIPE_LAST) {
            if (a.dtype == 0) { if (sp) P3D_PIPE64(0, true, PIPE_LAST); else P3D_PIPE64(0, false, PIPE_LAST); }
            else { if (sp) P3D_PIPE64(1, true, PIPE_LAST); else P3D_PIPE64(1, false, PIPE_LAST); }
        } else if (pm == PIPE_MID) {
            bool ts = false;
            if constexpr (pipe64_can_tstore<N>() && LB64 == 16) {
                ts = a.tstore && a.n1 % 16 == 0;
                if (ts) {
#define P3D_PIPE64_TS(DT, SP)                                                                                   \
    do {                                                                                                        \
        if ((e = allow_lds(row_pipe64_kernel<N, DT, SP, PIPE_MID, true>, lds64)) != hipSuccess) return e;       \
        row_pipe64_kernel<N, DT, SP, PIPE_MID, true><<<grid64, pipe64_threads<N>(), lds64, st>>>(a);            \
    } while (0)
                    if (a.dtype == 0) { if (sp) P3D_PIPE64_TS(0, true); else P3D_PIPE64_TS(0, false); }
                    else { if (sp) P3D_PIPE64_TS(1, true); else P3D_PIPE64_TS(1, false); }
#undef P3D_PIPE64_TS
                }
            }
            if (!ts) {
                if (a.dtype == 0) { if (sp) P3D_PIPE64(0, true, PIPE_MID); else P3D_PIPE64(0, false, PIPE_MID); }
                else { if (sp) P3D_PIPE64(1, true, PIPE_MID); else P3D_PIPE64(1, false, PIPE_MID); }
            }
        } else {
            return hipErrorInvalidValue;
        }
#undef P3D_PIPE64
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

// persistent steady-state row pass on a device with `cus` compute units
template <int N>
hipError_t launch_row_pipe(const RowArgs& a, int cus, hipStream_t st)
{
    if ((double)a.nslices * (double)wk_slice_stride(a.n1, N) >= 4294967296.0) return hipErrorNotSupported;  // 32-bit offsets
    const bool bits = a.bits != nullptr;
    const bool extra = a.adaptive || a.write_out || a.done != nullptr;
    const bool compact = bits && a.xc != nullptr;
    hipError_t e = launch_row_pipe64<N>(PIPE_MID, a, cus, st);   // the wave-uniform variant (it also honours the per-slice `done` flags)
    if (e != hipErrorNotSupported) return e;
    e = hipSuccess;
    if constexpr (Plan<N>::TPL > 64) {
        return hipErrorNotSupported;
    } else {
        constexpr int LB = ROW_THREADS / Plan<N>::TPL;
        // resident workgroups per CU: LDS (160 KiB per CU) and the register budget of the variant
        const int by_lds = (int)((160 * 1024) / row_lds_bytes<N>());
        const int wpe = (compact && P3D_COMPACT_LATE) ? P3D_PIPE_WAVES_PER_EU_COMPACT : P3D_PIPE_WAVES_PER_EU;
        const int by_regs = (wpe * 4 * 64) / ROW_THREADS;
        const int per_cu = by_lds < by_regs ? by_lds : by_regs;
        if (per_cu < 1) return hipErrorNotSupported;
        const long wgs = (long)cus * per_cu;
        const long groups = ((long)a.nslices * a.n1 + LB - 1) / LB;
        const dim3 grid((unsigned)(groups < wgs ? groups : wgs));
        constexpr size_t lds = row_lds_bytes<N>();
#define P3D_PIPE(BITS, DT, EXTRA, COMPACT)                                                                  \
    do {                                                                                                    \
        if ((e = allow_lds(row_pipe_kernel<N, BITS, DT, EXTRA, COMPACT>, lds)) != hipSuccess) return e;     \
        row_pipe_kernel<N, BITS, DT, EXTRA, COMPACT><<<grid, ROW_THREADS, lds, st>>>(a);                    \
    } while (0)
        if (a.dtype == 0) {
            if (compact) { if (extra) P3D_PIPE(true, 0, true, true); else P3D_PIPE(true, 0, false, true); }
            else if (bits) { if (extra) P3D_PIPE(true, 0, true, false); else P3D_PIPE(true, 0, false, false); }
            else { if (extra) P3D_PIPE(false, 0, true, false); else P3D_PIPE(false, 0, false, false); }
        } else {
            if (compact) { if (extra) P3D_PIPE(true, 1, true, true); else P3D_PIPE(true, 1, false, true); }
            else if (bits) { if (extra) P3D_PIPE(true, 1, true, false); else P3D_PIPE(true, 1, false, false); }
            else { if (extra) P3D_PIPE(false, 1, true, false); else P3D_PIPE(false, 1, false, false); }
        }
#undef P3D_PIPE
        return hipGetLastError();
    }
}

// real (float32) cubes, rows of 128 ... 1024 samples: the row-pair passes over the half-spectrum work buffer
template <int N>
hipError_t launch_row_real(int mode, const RowArgs& a, int cus, hipStream_t st)
{
    if constexpr ((64 % Plan<N>::TPL == 0 || Plan<N>::TPL % 64 == 0) && Plan<N>::TPL >= 8 && Plan<N>::TPL <= 256 && Plan<N>::PPT == 16) {
        constexpr int PW = Plan<N>::TPL >= 64 ? 1 : 64 / Plan<N>::TPL, WPL = Plan<N>::TPL >= 64 ? Plan<N>::TPL / 64 : 1;
        // Rows of 2048 samples (two wavefronts per pair: workgroup barriers around the split and the sums on top of those of the
        // transforms, 8 waves per CU) lose what the half spectrum gains: 1024 x 2048 x 256 float32, 20 iterations, 71.3 ms against
        // 68.4 on the complex path; 4096 samples: 37.1 against 43.0.  The kernel handles both; only the latter is switched on.
        if (WPL == 2 && !a.real_2048) return hipErrorNotSupported;
        if (a.n1 % (2 * PW) != 0 || a.bits64 == nullptr || a.cbase == nullptr || a.dtype != 1) return hipErrorNotSupported;
        if ((double)a.nslices * (double)wk_slice_stride(a.n1, N / 2 + 1) >= 4294967296.0) return hipErrorNotSupported;
        constexpr size_t lds = pipe64_lds_bytes<N>();
        constexpr int UPB = pipe64_threads<N>() / 64 / WPL;
        const long groups = ((long)a.nslices * (a.n1 / (2 * PW)) + UPB - 1) / UPB;
        int per_cu = (int)((160 * 1024) / lds);               // workgroups a CU holds: LDS and 4 waves per SIMD
        if (per_cu > 16 / (pipe64_threads<N>() / 64)) per_cu = 16 / (pipe64_threads<N>() / 64);
        if (per_cu < 1) per_cu = 1;
        const long wgs = (long)cus * per_cu;
        const dim3 grid((unsigned)(groups < wgs ? groups : wgs));
        hipError_t e = hipSuccess;
#define P3D_REAL(MODE, SP)                                                                      \
    do {                                                                                        \
        if ((e = allow_lds(row_real_kernel<N, MODE, SP>, lds)) != hipSuccess) return e;         \
        row_real_kernel<N, MODE, SP><<<grid, pipe64_threads<N>(), lds, st>>>(a);                \
    } while (0)
        const bool sp = a.nzl != nullptr;
        switch (mode) {
            case REAL_FIRST: P3D_REAL(REAL_FIRST, false); break;
            case REAL_MID: if (sp) P3D_REAL(REAL_MID, true); else P3D_REAL(REAL_MID, false); break;
            case REAL_LAST: if (sp) P3D_REAL(REAL_LAST, true); else P3D_REAL(REAL_LAST, false); break;
            default: return hipErrorInvalidValue;
        }
#undef P3D_REAL
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int N>
hipError_t launch_row(int mode, const RowArgs& a, hipStream_t st)
{
    const bool bits = a.bits != nullptr;
    switch (mode) {
        case ROW_FIRST: return bits ? launch_row_one<N, ROW_FIRST, true>(a, st) : launch_row_one<N, ROW_FIRST, false>(a, st);
        case ROW_MID: return bits ? launch_row_one<N, ROW_MID, true>(a, st) : launch_row_one<N, ROW_MID, false>(a, st);
        case ROW_LAST: return bits ? launch_row_one<N, ROW_LAST, true>(a, st) : launch_row_one<N, ROW_LAST, false>(a, st);
        case ROW_SPREAD_INV: return launch_row_one<N, ROW_SPREAD_INV, false>(a, st);
        case ROW_GATHER_FWD: return launch_row_one<N, ROW_GATHER_FWD, false>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int N, int MODE>
hipError_t launch_col_one(const ColArgs& a, hipStream_t st)
{
    constexpr int T = col_tile<N>();
    constexpr int THREADS = T * Plan<N>::TPL;
    const dim3 grid((a.n2 + T - 1) / T, a.nslices);
    constexpr size_t lds = col_lds_bytes<N>();
    hipError_t e = allow_lds(col_kernel<N, T, MODE>, lds);
    if (e != hipSuccess) return e;
    col_kernel<N, T, MODE><<<grid, THREADS, lds, st>>>(a);
    return hipGetLastError();
}

// persistent column pass of the iteration (column-blocked buffer in and out), `wgs` resident workgroups
template <int N>
hipError_t launch_col_pipe(const ColArgs& a, int cus, hipStream_t st)
{
    constexpr int T = col_tile<N>();
    if constexpr (T % 8 == 0 && Plan<N>::PPT == 16) {
        if (a.in_std || a.out_std || a.in != a.out || cus < 1) return hipErrorNotSupported;
        if ((double)wk_slice_stride(N, a.n2) * 8.0 >= 2147483648.0) return hipErrorNotSupported;
        constexpr size_t lds = col_lds_bytes<N>();
        int per_cu = (int)((160 * 1024) / lds);
        const int by_waves = (P3D_COLPIPE_WAVES_PER_EU * 4) / (T * Plan<N>::TPL / 64);
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        const long total = (long)a.nslices * ((a.n2 + T - 1) / T);
        const long wgs = (long)cus * per_cu;
        const dim3 grid((unsigned)(total < wgs ? total : wgs));
        hipError_t e = hipSuccess;
#define P3D_COLPIPE(OP)                                                                       \
    do {                                                                                      \
        if ((e = allow_lds(col_pipe_kernel<N, T, OP>, lds)) != hipSuccess) return e;          \
        col_pipe_kernel<N, T, OP><<<grid, T * Plan<N>::TPL, lds, st>>>(a);                    \
    } while (0)
        if (a.sh.tau != nullptr) {   // the column pass of a SHEARLET iteration
            if ((e = allow_lds(col_pipe_kernel<N, T, 0, true>, lds)) != hipSuccess) return e;
            col_pipe_kernel<N, T, 0, true><<<grid, T * Plan<N>::TPL, lds, st>>>(a);
        } else if (a.op == 1) P3D_COLPIPE(1); else if (a.op == 2) P3D_COLPIPE(2); else P3D_COLPIPE(0);
#undef P3D_COLPIPE
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int N>
hipError_t launch_col(int mode, const ColArgs& a, hipStream_t st)
{
    switch (mode) {
        case COL_ITER:
            if (a.op == 1) return launch_col_one<N, COL_ITER_SOFT>(a, st);
            if (a.op == 2) return launch_col_one<N, COL_ITER_GARROTE>(a, st);
            return launch_col_one<N, COL_ITER>(a, st);
        case COL_STATS: return launch_col_one<N, COL_STATS>(a, st);
        case COL_FWD: return launch_col_one<N, COL_FWD>(a, st);
        case COL_INV: return launch_col_one<N, COL_INV>(a, st);
        case COL_SHRINK: return launch_col_one<N, COL_SHRINK>(a, st);
        default: return hipErrorInvalidValue;
    }
}

// what the API layer sees of one line length
struct LineOps {
    int n;
    int col_tile;
    int tpl;  // threads per line (layout of the packed mask words)
    int ppt;
    hipError_t (*row)(int mode, const RowArgs&, hipStream_t);
    hipError_t (*col)(int mode, const ColArgs&, hipStream_t);
    hipError_t (*row_pipe)(const RowArgs&, int cus, hipStream_t);  // hipErrorNotSupported when a line spans waves
    size_t row_lds;
    int row_tw_slots;                    // length of the row pass's twiddle tables ...
    void (*build_row_tw)(c32* out);      // ... and their builder
    int col_tw_slots;                    // the same for the column pass (ColTables)
    void (*build_col_tw)(c32* out);
    hipError_t (*row_real)(int mode, const RowArgs&, int cus, hipStream_t);   // REAL_* passes (hipErrorNotSupported where absent)
    hipError_t (*row_pipe64)(int pm, const RowArgs&, int cus, hipStream_t);   // PIPE_FIRST / PIPE_MID / PIPE_LAST (hipErrorNotSupported where absent)
    hipError_t (*col_pipe)(const ColArgs&, int cus, hipStream_t);             // persistent COL_ITER (hipErrorNotSupported where absent)
};

}  // namespace p3d
