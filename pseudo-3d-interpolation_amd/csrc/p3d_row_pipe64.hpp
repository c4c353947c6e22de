// p3d_row_pipe64.hpp -- the wave-uniform persistent row pass (rows of 128 ... 4096 samples, binary mask, compact observed samples): first, steady-state and last pass of a complex64 job.
// Part of the two fused passes of one POCS iteration; the overview (pass structure, work-buffer layout) is at the top of p3d_kernels.hpp.
#pragma once

#include "p3d_kernels_common.hpp"

namespace p3d {

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
// ADAPT (steady state only): APOCS -- the forward transform takes the mix of iterate and observation (POCS.py:574-575) instead of the
// iterate; the mix needs nothing but the observed sample and the mask bit of the element it replaces, both at hand in the re-insertion loop
template <int N, int DT, bool SPARSE, int PM, bool TS = false, bool ADAPT = false>
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
        // (ADAPT + write_out: the row's place in the result cube, as store_cube addresses it)
        __amdgpu_buffer_rsrc_t wo_srd = sums_srd;
        unsigned wo_so = 0u, wo_vo = BUF_OOB;
        if constexpr (ADAPT && PM == PIPE_MID) {
            if (a.write_out) {
                wo_srd = buf_srd(reinterpret_cast<const char*>(a.out) + (size_t)cur.slice * cube_slice_bytes, cube_slice_bytes);
                wo_so = cube_soff(cur);
                wo_vo = cur.on ? lane_c : BUF_OOB;
            }
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            c32 xn = v[q] * a.scale;
            const float w = __builtin_amdgcn_inverse_ballot_w64(mw_cur[q]) ? w_obs : 1.0f;
            xn = axpby(xn, w, raw_c32(bx[q]), a.alpha);        // POCS.py:616-619
            acc += abs_c32(xn);
            if constexpr (ADAPT && PM == PIPE_MID) {
                // APOCS with the early exit keeps every iterate (RowArgs::write_out): a slice that converges is simply left alone afterwards
                if (a.write_out) {
                    if (DT == 0) buf_store_c32(wo_srd, wo_vo, wo_so, xn);
                    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xn.x), wo_srd, (int)wo_vo, (int)wo_so, 0);
                    wo_so += qc;
                }
                // x_input of the next iteration (POCS.py:574-575), the expressions of row_kernel (same bits)
                c32 xo = raw_c32(bx[q]);
                asm volatile("" : "+v"(xo.x), "+v"(xo.y), "+v"(xn.x), "+v"(xn.y));   // one element at a time (not gathered into vectors across q: the kernel has no registers for that)
                const float m = __builtin_amdgcn_inverse_ballot_w64(mw_cur[q]) ? 1.0f : 0.0f;
                const c32 blend = xo * a.alpha + xn * w;
                c32 mix = blend + (xo - xn * m) * (1.0f - a.alpha);
                asm volatile("" : "+v"(mix.x), "+v"(mix.y));
                v[q] = mix;
            } else {
                v[q] = xn;
            }
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

}  // namespace p3d
