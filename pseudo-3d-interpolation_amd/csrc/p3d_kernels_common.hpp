// p3d_kernels_common.hpp -- knobs, pass modes, RowArgs / ColArgs and the helpers every pass shares (split out of p3d_kernels.hpp by pass in round 3: same text, same bits).
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
    // Rows on which a shearlet's spectrum vanishes altogether carry nothing through the whole iteration (Psi_s * F is zero there, and
    // what the column pass would produce there is multiplied by Psi_s = 0 again): bit g of sup[s * sup_words + g / 32 ...] says that
    // rows 8 g ... 8 g + 7 of Psi_s hold a non-zero sample.  The three passes skip the other row groups -- neither computed,
    // stored nor read; a Parseval frame covers every frequency about twice, so well over half of the (shearlet, row) pairs go
    // (configs[4]: 57 % of the groups).  nullptr: dense.
    const unsigned* sup;
    int sup_words;
    // float32 cubes, symmetric spectra: the coefficients are real, so the work slices are Hermitian along the rows (row N - k = the
    // conjugate of row k) and only rows 0 ... n1/2 are computed, stored and read; the column pass (p3d_col_shear.hpp) rebuilds the
    // other half while it loads, the gather pass's output is completed by mirroring.  0: all rows.
    int half;
};
__host__ __device__ inline int shear_rows(const ShearArgs& sh, int n1) { return sh.half ? n1 / 2 + 1 : n1; }
// row group of 8 rows `g` of shearlet `s` holds a non-zero spectrum sample (dense when there is no table)
__device__ __forceinline__ bool shear_group_on(const ShearArgs& sh, int s, int g)
{
    return sh.sup == nullptr || ((sh.sup[(size_t)s * sh.sup_words + (g >> 5)] >> (g & 31)) & 1u) != 0u;
}

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
    int only_done_lo;      // LAST, only_done > 0: the finalize launch takes the slices with only_done_lo < done <= only_done (it runs every few iterations).
                           // INVARIANT every pass of the loop must keep (row, column, flexible, chirp-z, percentile-fused ...): a slice with done != 0 is
                           // left ALONE -- its work rows hold the forward row transform of its converged iterate until the finalize launch reads them,
                           // up to 8 iterations later.  A pass that ignores `done` corrupts exactly the eps > 0 jobs whose slices converge at different
                           // iterations; P3D_CHECK_DONE_ROWS=1 (p3d_pocs_run_dev) checksums those rows every iteration, tests/test_gpu_parity.py runs it
    int only_done;         // LAST, > 0: "finalize" launch of the early exit -- only slices whose done is in (only_done_lo, only_done]; their work
                           // rows hold the forward row transform of the converged iterate, which is handed to `out`
    const unsigned long long* bits64;  // rows of whole wavefronts: the mask as lane masks, word pipe64_word(row, TPL/64, wsub, q) bit l =
                                       // mask[row][64*wsub + l + TPL*q]
    const unsigned long long* nzl;     // the same for nzm, per slice (pipe64_word); nullptr: dense
    const unsigned* cbase;             // observed traces of the slice before the first column of each word (pipe64_word)
    const unsigned long long* bits32;  // row_pipe32_kernel (rows of 1024 samples, a row pair per wavefront): word (unit u, register k) bit l =
                                       // mask[2u + (l >> 5)][(l & 31) + 32 k]; [n1 / 2][32]
    const c32* tw32;                   // ... and its twiddles (P32::build_tw, device)
    const unsigned long long* mbits;   // mixed-radix row pass (p3d_mix.hpp), binary masks: word (row, tl) bit q = mask[row][tl + TPL q]; [n1][TPL]
    const unsigned* mbase;             // ... observed positions of the slice before thread (row, tl)'s own, in (row, tl) order: the compact samples
                                       //     `xc` of that pass are stored thread by thread (thread-major), entry [n1 * TPL] = nobs
    float alpha;
    float scale;           // 1/(n1*N)
    int len;               // N, the row length (the tuned kernels know it at compile time; p3d_flex.hip reads it here)
    int real_2048;         // host side only: the row-pair path for rows of 2048 samples is on (off: experiment switch P3D_NO_REAL_2048)
    int tstore;            // host side only: rows of one wavefront hand their transforms round through LDS and store 1-KiB runs (P3D_NO_TSTORE unset)
    int host_sw;           // host side only: P3D_SW_* experiment switches of the plan (read from the environment once per plan)
    ShearArgs sh;          // ROW_SPREAD_INV, ROW_GATHER_FWD
};

struct ColArgs {
    const c32* in;
    c32* out;           // may alias `in`
    const c32* tw;      // the column pass's twiddle tables (ColTables<N>, device)
    const c32* tau;     // [nslices][niter] (COL_ITER, optional for COL_FWD)
    const int* done;    // per slice: != 0 -> the column pass must not touch the slice (see RowArgs::only_done_lo for the invariant)
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
    int herm_n2;        // COL_STATS on the half-spectrum work buffer of a float32 cube (columns 0 ... herm_n2/2 of herm_n2): the statistics
                        // of the WHOLE Hermitian spectrum -- a column with a mirror image counts twice and offers its conjugates to the
                        // lexicographic maximum (0: plain)
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

}  // namespace p3d
