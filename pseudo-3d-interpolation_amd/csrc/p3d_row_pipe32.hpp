// p3d_row_pipe32.hpp -- the persistent row pass of 1024-sample rows with ONE exchange per transform (round 4): first, steady-state
// and last pass of a job on the complex path, the same role as row_pipe64_kernel<1024> (p3d_row_pipe64.hpp).
//
// Why.  row_pipe64_kernel<1024> was the kernel furthest below its roofline on the headline cube (1.55 ms moving 5.4 GB), and its own
// counters and stamps said what held it: the two 1024-point transforms of a row -- radix 4-16-16 / 16-16-4 on 16 points per thread,
// FOUR trips through LDS per row -- take 1.0 of the 1.55 ms, vector ALU and LDS taking turns under a 16-wave lock-step
// (profiles/r03_rowpass_crosslane.txt).  1024 = 32 x 32: with 32 points per thread a transform is
// two in-register radix-32 butterflies around ONE exchange.
//
// Shape.  A row lives in HALF a wavefront (lane j of 32 holds elements j + 32 k, k = 0 ... 31); a wavefront works on a UNIT = two
// adjacent rows 2u, 2u + 1 of one slice (lanes 0-31 / 32-63), a workgroup of 8 wavefronts on 16 adjacent rows, one workgroup per CU
// (147 KiB of LDS, 512 threads at up to 256 registers).  Per register k a wavefront touches four column blocks x (row 2u | row 2u + 1):
// four whole 128-byte lines of the column-blocked work buffer per instruction, without any help from its neighbours.
// Everything per unit is wave-uniform, as in row_pipe64_kernel: slice / unit in scalar registers, the trace mask of register k as one
// 64-bit lane mask (low half row 2u, high half row 2u + 1: RowArgs::bits32), the emptied-block flags from the per-slice words of the
// 64-lane layout (RowArgs::nzl: both halves of a wavefront see the same 32 columns), ranks among the observed traces from v_mbcnt,
// every access a buffer instruction whose switched-off lanes are out of range.
//
// Conventions with the other kernels.  The work buffer is the common column-blocked one (the column pass does not care who wrote it).
// The compact observed samples are ordered unit by unit, inside a unit register by register, inside a register row 2u's lanes then row
// 2u + 1's (the order of a wavefront's lanes): rowbase[2u] + running population count.  That order is private to this kernel family:
// whoever writes RowArgs::xc with PIPE_FIRST here must read it with PIPE_MID / PIPE_LAST here (p3d_api.hip decides per job, before
// the first pass).  The forward transform of the first pass is a different sequence of roundings from line_fft<1024>'s, so a plan
// uses THIS first pass for every job and every statistics pass whenever it can use it at all (p3d_plan::use32) -- the schedule's
// statistics and the first iteration's spectrum must be the same bits (a threshold that equals the largest coefficient keeps it).
#pragma once

#include "p3d_kernels_common.hpp"
#include "p3d_fft32.hpp"
#include "p3d_row_pipe64.hpp"   // PipeMode, the ablation switches

namespace p3d {

#ifndef P3D_PIPE32_LOCKSTEP
#define P3D_PIPE32_LOCKSTEP 0   // 1 (experiment): one workgroup barrier per unit in front of the stores, the eight waves write sixteen adjacent rows together --
                                // what row_pipe64_kernel needs (its waves hold 64-byte HALVES of the lines) costs here, where a wave writes whole lines: 1.90 against 1.53 ms
#endif

#if defined(__HIPCC__)
P3D_D void p32_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// one row per half wavefront; LDS operations of a wavefront execute in program order, so the exchange needs no barrier
template <int DIR>
P3D_D void fft1024_32(c32 (&v)[32], c32* row, const c32* tw, int j)
{
    dft32<DIR>(v);
    p32_wave_sync();   // (the previous reads of this buffer are behind us)
    {
        float4* const p = reinterpret_cast<float4*>(row + 34 * j);
#pragma unroll
        for (int m = 0; m < 16; ++m) p[m] = float4{v[2 * m].x, v[2 * m].y, v[2 * m + 1].x, v[2 * m + 1].y};
    }
    p32_wave_sync();
    p32_half2<DIR>(v, row, tw, j);
}

template <int DT, bool SPARSE, int PM, bool ADAPT = false>
__global__ __launch_bounds__(P32::THREADS, 2) void row_pipe32_kernel(const RowArgs a)
{
    constexpr int N = P32::N, PPT = 32, UPB = P32::UPB;
    constexpr unsigned ES = DT == 0 ? 8u : 4u;    // bytes per observed sample
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* const twl = reinterpret_cast<c32*>(smem_raw);
    c32* const data = twl + P32::TW;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // the unit of this wavefront inside its workgroup
    const int lane = tid & 63;
    const unsigned half = (unsigned)lane >> 5, j = (unsigned)lane & 31u;
    for (int i = tid; i < P32::TW; i += P32::THREADS) twl[i] = a.tw32[i];
    __syncthreads();
    c32* const row_lds = data + (size_t)(wave * 2 + (int)half) * P32::LSTR;

    const unsigned upslice = (unsigned)a.n1 >> 1;                      // units per slice (n1 even)
    const unsigned total = (unsigned)a.nslices * upslice;
    const unsigned wblk8 = (unsigned)a.n1 * 64u;                       // bytes of one column block of a slice
    const size_t wstride = wk_slice_stride(a.n1, N);
    const unsigned slice_bytes = (unsigned)(wstride * 8);              // at most 128 MiB
    // element j + 32 k of row 2u + half: column block (j >> 3) + 4 k, byte (row * 8 + (j & 7)) * 8 inside it
    const unsigned lane_w = (j >> 3) * wblk8 + (j & 7u) * 8u + half * 64u;
    const unsigned qs = 4u * wblk8;                                    // bytes from register k to k + 1 (32 columns)

    typedef const unsigned long long __attribute__((address_space(4))) * kmask_t;
    typedef const unsigned __attribute__((address_space(4))) * kuint_t;
    typedef const int __attribute__((address_space(4))) * kint_t;
    const kmask_t k_bits = (kmask_t)a.bits32, k_nzl = (kmask_t)a.nzl;
    const kuint_t k_rowbase = (kuint_t)a.rowbase;
    const kint_t k_done = (kint_t)a.done;   // early exit (eps > 0): set between launches, constant during one

    struct Where { unsigned slice, row; bool on, zero; };   // row: the unit's index inside its slice
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
    auto work_soff = [&](const Where& w) -> unsigned { return w.row * 128u; };
    // by[] <- the unit's elements of the work buffer; emptied column blocks (SPARSE) read as zero without a memory access
    auto issue_work = [&](raw64 (&dst)[PPT], const Where& w) {
        const __amdgpu_buffer_rsrc_t srd = work_srd(w);
        unsigned so = work_soff(w);
        const kmask_t nz = k_nzl + (size_t)w.slice * 16;   // word q bit l: the block of column 64 q + l kept something (pipe64_word of one wavefront)
        unsigned long long nzw[16];
        if (SPARSE) {
#pragma unroll
            for (int q = 0; q < 16; ++q) nzw[q] = nz[q];
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (SPARSE) {
                const unsigned m32 = (unsigned)(nzw[k >> 1] >> (32 * (k & 1)));
                const unsigned long long both = (unsigned long long)m32 | ((unsigned long long)m32 << 32);   // the two rows of the unit see the same columns
                raw64 r = 0;
                // (a register whose 32 columns were all emptied skips the instruction behind a wave-uniform branch, as in row_pipe64_kernel)
                if (m32 != 0u && !P3D_ABL_NOWORK) r = buf_load_raw64(srd, __builtin_amdgcn_inverse_ballot_w64(both) ? lane_w : BUF_OOB, so);
                dst[k] = r;
            } else {
                dst[k] = buf_load_raw64(srd, P3D_ABL_NOWORK ? BUF_OOB : lane_w, so);
            }
            so += qs;
        }
    };
    // the unit's 32 mask words: bit l of word k = mask[2u + (l >> 5)][(l & 31) + 32 k]
    auto mask_words = [&](unsigned long long (&mw)[PPT], const Where& w) {
        kmask_t m = k_bits + (size_t)w.row * PPT;
        asm volatile("" : "+s"(m));
#pragma unroll
        for (int k = 0; k < PPT; ++k) mw[k] = m[k];
    };
    // bx[] <- the unit's observed samples from the compact array (zero where the trace is missing)
    auto issue_obs = [&](raw64 (&dst)[PPT], const Where& w, const unsigned long long (&mw)[PPT]) {
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.xc) + (size_t)w.slice * a.nobs * ES, a.nobs * ES);
        unsigned cb = k_rowbase[2u * w.row] * ES;   // the unit's first sample; then register by register
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const unsigned long long m = mw[k];
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned vo = (__builtin_amdgcn_inverse_ballot_w64(m) && !P3D_ABL_NOOBS) ? rank * ES : BUF_OOB;
            if (DT == 0) dst[k] = buf_load_raw64(srd, vo, cb);
            else dst[k] = (raw64)__builtin_amdgcn_raw_buffer_load_b32(srd, (int)vo, (int)cb, 0);   // (imaginary part: zero bits)
            cb += (unsigned)__builtin_popcountll(m) * ES;
        }
    };
    auto store_work = [&](const c32 (&src)[PPT], const Where& w, bool really) {
        const __amdgpu_buffer_rsrc_t srd = work_srd(w);
        unsigned so = work_soff(w);
        const unsigned vo = (really && !P3D_ABL_NOSTORE) ? lane_w : BUF_OOB;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            buf_store_c32(srd, vo, so, src[k]);
            so += qs;
        }
    };
    // row-major cubes (observed cube `x`, result cube `out`; complex64 or float32): element j + 32 k of row 2u + half
    const unsigned cube_slice_bytes = (unsigned)a.n1 * (unsigned)N * ES;
    const unsigned lane_c = (half * (unsigned)N + j) * ES;
    const unsigned qc = 32u * ES;
    auto cube_soff = [&](const Where& w) -> unsigned { return w.row * (unsigned)(2 * N) * ES; };
    auto issue_cube = [&](raw64 (&dst)[PPT], const Where& w) {
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.x) + (size_t)w.slice * cube_slice_bytes, cube_slice_bytes);
        unsigned so = cube_soff(w);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (DT == 0) dst[k] = buf_load_raw64(srd, lane_c, so);
            else dst[k] = (raw64)__builtin_amdgcn_raw_buffer_load_b32(srd, (int)lane_c, (int)so, 0);
            so += qc;
        }
    };
    auto store_cube = [&](const c32 (&src)[PPT], const Where& w) {   // np.real for float32 cubes, POCS.py:656
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.out) + (size_t)w.slice * cube_slice_bytes, cube_slice_bytes);
        unsigned so = cube_soff(w);
        const unsigned vo = w.on ? lane_c : BUF_OOB;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const c32 val = w.zero ? c32{0.f, 0.f} : src[k];
            if (DT == 0) buf_store_c32(srd, vo, so, val);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val.x), srd, (int)vo, (int)so, 0);
            so += qc;
        }
    };
    // per-row sums of |x|: [nslices][n1] doubles; a null table swallows the stores
    const __amdgpu_buffer_rsrc_t sums_srd = buf_srd(a.sums, a.sums != nullptr ? (unsigned)a.nslices * (unsigned)a.n1 * 8u : 0u);
    const float w_obs = 1.0f - a.alpha * 1.0f;   // POCS.py:616 at an observed trace
    const float ws_obs = w_obs * a.scale;
    auto store_row_sum = [&](float acc, const Where& w) {
        double ws = (double)acc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) ws += __shfl_down(ws, o, 32);
        const unsigned so = (w.slice * (unsigned)a.n1 + 2u * w.row) * 8u;
        buf_store_f64(sums_srd, (j == 0u && w.on && !w.zero) ? half * 8u : BUF_OOB, so, ws);
    };

    const unsigned step = gridDim.x * UPB;
    // workgroups b, b + 8, ... share an XCD (round-robin dispatch: speed only): give the workgroups of one XCD adjacent row groups
    unsigned wg = blockIdx.x;
    if (gridDim.x % 8 == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    unsigned g = wg * UPB + (unsigned)wave;
    Where cur = locate(g);
    c32 v[PPT];
    raw64 bx[PPT], by[PPT];
    unsigned long long mw_cur[PPT];

    if constexpr (PM == PIPE_FIRST) {
        // ---- first pass of a job: observed cube -> (compact samples, sum |x_obs|) and forward row transform -> work buffer ----
        const bool tables = k_bits != nullptr && k_rowbase != nullptr;   // uniform for the launch; the compact copy also needs a.xc
        const bool compact = tables && a.xc != nullptr;
        const __amdgpu_buffer_rsrc_t none = buf_srd(nullptr, 0u);
        issue_cube(bx, cur);
        if (tables) mask_words(mw_cur, cur);
        {   // as many (dropped) stores as one trip of the loop issues: the compiler's wait counts at the loop header are then the same along
            // both edges into it (vector-memory operations retire in issue order; see row_pipe64_kernel)
#pragma unroll
            for (int k = 0; k < PPT; ++k) v[k] = c32{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < PPT; ++k) buf_store_c32(none, BUF_OOB, 0u, v[k]);
            buf_store_f64(sums_srd, BUF_OOB, 0u, 0.0);
            store_work(v, cur, false);
        }
        for (unsigned g0 = wg * UPB; g0 < total; g0 += step) {
            const Where nxt = locate(g + step);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < PPT; ++k) v[k] = DT == 0 ? raw_c32(bx[k]) : c32{__uint_as_float((unsigned)bx[k]), 0.f};
            __builtin_amdgcn_sched_barrier(0);
            float acc = 0.f;
            {
                const __amdgpu_buffer_rsrc_t xsrd = compact ? buf_srd(reinterpret_cast<const char*>(a.xc) + (size_t)cur.slice * a.nobs * ES, a.nobs * ES) : none;
                unsigned cb = compact ? k_rowbase[2u * cur.row] * ES : 0u;
                bool bad = false;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const unsigned long long m = tables ? mw_cur[k] : 0ull;
                    const bool set = __builtin_amdgcn_inverse_ballot_w64(m);
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    const unsigned vo = (set && cur.on && compact) ? rank * ES : BUF_OOB;
                    if (DT == 0) buf_store_c32(xsrd, vo, cb, v[k]);
                    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k].x), xsrd, (int)vo, (int)cb, 0);
                    cb += (unsigned)__builtin_popcountll(m) * ES;
                    bad = bad || (!set && (v[k].x != 0.f || v[k].y != 0.f));
                    acc += abs_c32(v[k]);
                    if constexpr (ADAPT) {   // APOCS: x_old = x at the first iteration (POCS.py:549, 574-575), the expressions of row_kernel<ROW_FIRST>
                        c32 x = v[k];
                        asm volatile("" : "+v"(x.x), "+v"(x.y));
                        const float mk = set ? 1.0f : 0.0f;
                        const float wk = 1.0f - a.alpha * mk;
                        const c32 blend = x * a.alpha + x * wk;
                        c32 mix = blend + (x - x * mk) * (1.0f - a.alpha);
                        asm volatile("" : "+v"(mix.x), "+v"(mix.y));
                        v[k] = mix;
                    }
                }
                if (compact && cur.on && __any(bad)) {
                    if (lane == 0) atomicOr(a.violation, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            store_row_sum(acc, cur);
            __builtin_amdgcn_sched_barrier(0);
            issue_cube(bx, nxt);
            unsigned long long mw_nxt[PPT];
            if (tables) mask_words(mw_nxt, nxt);
            __builtin_amdgcn_sched_barrier(0);
            fft1024_32<FWD>(v, row_lds, twl, (int)j);
            __builtin_amdgcn_sched_barrier(0);
#if P3D_PIPE32_LOCKSTEP
            __builtin_amdgcn_s_barrier();
#endif
            store_work(v, cur, cur.on);
            __builtin_amdgcn_sched_barrier(0);
            if (tables) {
#pragma unroll
                for (int k = 0; k < PPT; ++k) mw_cur[k] = mw_nxt[k];
            }
            g += step;
            cur = nxt;
        }
        return;
    } else {
        issue_work(by, cur);
        {   // (dropped stores: exact wait counts at the loop header, see the first pass)
#pragma unroll
            for (int k = 0; k < PPT; ++k) v[k] = c32{0.f, 0.f};
            buf_store_f64(sums_srd, BUF_OOB, 0u, 0.0);
            store_work(v, cur, false);
        }
        {
            unsigned long long mw0[PPT];
            mask_words(mw0, cur);
            issue_obs(bx, cur, mw0);
        }
        for (unsigned g0 = wg * UPB; g0 < total; g0 += step) {
            const Where nxt = locate(g + step);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < PPT; ++k) v[k] = raw_c32(by[k]);
            __builtin_amdgcn_sched_barrier(0);
            if (!P3D_ABL_NOFFT) fft1024_32<INV>(v, row_lds, twl, (int)j);
            __builtin_amdgcn_sched_barrier(0);
            // (the samples are first touched HERE: otherwise the compiler starts on bx * alpha in the middle of the transform and waits there)
#pragma unroll
            for (int k = 0; k < PPT; ++k) asm volatile("" : "+v"(bx[k]));
            // (the unit's mask words are fetched again here rather than carried from the request of its samples across both transforms:
            // 64 scalar registers held that long spill into vector lanes, a v_readlane per use)
            mask_words(mw_cur, cur);
            float acc = 0.f;
            __amdgpu_buffer_rsrc_t wo_srd = sums_srd;
            unsigned wo_so = 0u, wo_vo = BUF_OOB;
            if constexpr (ADAPT && PM == PIPE_MID) {
                if (a.write_out) {   // APOCS with the early exit keeps every iterate
                    wo_srd = buf_srd(reinterpret_cast<const char*>(a.out) + (size_t)cur.slice * cube_slice_bytes, cube_slice_bytes);
                    wo_so = cube_soff(cur);
                    wo_vo = cur.on ? lane_c : BUF_OOB;
                }
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool obs = __builtin_amdgcn_inverse_ballot_w64(mw_cur[k]);
                const float w = obs ? w_obs : 1.0f;
                // POCS.py:616-619 with the 1 / (N1 N2) of the inverse transform folded into the weight (one multiply per sample less)
                c32 xn = axpby(v[k], obs ? ws_obs : a.scale, raw_c32(bx[k]), a.alpha);
                acc += abs_c32(xn);
                if constexpr (ADAPT && PM == PIPE_MID) {
                    if (a.write_out) {
                        if (DT == 0) buf_store_c32(wo_srd, wo_vo, wo_so, xn);
                        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xn.x), wo_srd, (int)wo_vo, (int)wo_so, 0);
                        wo_so += qc;
                    }
                    // x_input of the next iteration (POCS.py:574-575), the expressions of row_kernel (same bits)
                    c32 xo = raw_c32(bx[k]);
                    asm volatile("" : "+v"(xo.x), "+v"(xo.y), "+v"(xn.x), "+v"(xn.y));
                    const float mk = obs ? 1.0f : 0.0f;
                    const c32 blend = xo * a.alpha + xn * w;
                    c32 mix = blend + (xo - xn * mk) * (1.0f - a.alpha);
                    asm volatile("" : "+v"(mix.x), "+v"(mix.y));
                    v[k] = mix;
                } else {
                    v[k] = xn;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!P3D_ABL_NOSUMS) store_row_sum(acc, cur);
            __builtin_amdgcn_sched_barrier(0);
            // the next unit's work-buffer elements are requested BEFORE the forward transform, its observed samples behind it
            issue_work(by, nxt);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PM == PIPE_MID) {
                if (!P3D_ABL_NOFFT) fft1024_32<FWD>(v, row_lds, twl, (int)j);
            }
            __builtin_amdgcn_sched_barrier(0);
            // A wavefront can have 63 vector-memory operations in flight.  With 32 points per thread a unit issues 32 + 32 + 33 of them, so
            // the ORDER decides what a full counter waits for: stores behind the freshly requested samples stall on a memory latency
            // per unit (measured: +0.39 ms per pass); samples behind the stores wait, a whole inverse transform later, for
            // acknowledgements that have long arrived.
            if constexpr (PM == PIPE_MID) {
#if P3D_PIPE32_LOCKSTEP
                __builtin_amdgcn_s_barrier();
#endif
                store_work(v, cur, cur.on);
            } else {
                store_cube(v, cur);   // last pass of a job: whole rows of the result cube
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                unsigned long long mw_nxt[PPT];
                mask_words(mw_nxt, nxt);
                issue_obs(bx, nxt, mw_nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
            g += step;
            cur = nxt;
        }
    }
}
#endif  // __HIPCC__

}  // namespace p3d
