// p3d_row_kernels.hpp -- the space (row) pass: one-launch kernel (first / steady state / last pass, SHEARLET row passes) and the persistent, software-pipelined pass with per-lane bookkeeping (rows below 128 samples, APOCS, odd row counts).
// Part of the two fused passes of one POCS iteration; the overview (pass structure, work-buffer layout) is at the top of p3d_kernels.hpp.
#pragma once

#include "p3d_kernels_common.hpp"

namespace p3d {

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
    int nrows = a.n1;
    if constexpr (MODE == ROW_SPREAD_INV || MODE == ROW_GATHER_FWD) {
        nrows = shear_rows(a.sh, a.n1);
        if (rgroup * LB >= nrows) return;   // (the grid is rounded up to whole groups of eight workgroups for the XCD placement)
    }
    const bool valid = row < nrows;

    const int dn = a.done ? a.done[slice] : 0;
    if (MODE == ROW_LAST && a.only_done) {
        if (dn <= a.only_done_lo || dn > a.only_done) return;
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
        // (row groups of 8 on which Psi_s vanishes are skipped, see ShearArgs::sup: a workgroup's rows span the groups g0 ... g1)
        const int g0 = (rgroup * LB) >> 3, g1 = min(rgroup * LB + LB - 1, nrows - 1) >> 3;
        for (int s = 0; s < a.sh.nsh; ++s) {
            bool any = false;
            for (int g = g0; g <= g1; ++g) any = any || shear_group_on(a.sh, s, g);
            if (!any) continue;   // workgroup-uniform
            const bool mine = valid && shear_group_on(a.sh, s, vrow >> 3);
            const float* const w = a.sh.psi + (size_t)s * a.n1 * N;
            c32* const ws = a.work + ((size_t)slice * a.sh.nsh + s) * wk_slice_stride(a.n1, N);
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = fr[q] * (mine ? w[off + TPL * q] : 0.f);
            line_fft<N, INV, WAVE>(v, lds, tw, tl);
            __syncthreads();   // adjacent rows share 128-byte lines of the work buffer: store them together
            if (mine) {
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
        const int g0 = (rgroup * LB) >> 3, g1 = min(rgroup * LB + LB - 1, nrows - 1) >> 3;
        for (int s = 0; s < a.sh.nsh; ++s) {
            bool any = false;
            for (int g = g0; g <= g1; ++g) any = any || shear_group_on(a.sh, s, g);
            if (!any) continue;   // Psi_s = 0 on all rows of this workgroup: they add nothing (and the column pass did not store them)
            const bool mine = valid && shear_group_on(a.sh, s, vrow >> 3);
            const c32* const ws = a.work + ((size_t)slice * a.sh.nsh + s) * wk_slice_stride(a.n1, N);
            const float* const w = a.sh.psi + (size_t)s * a.n1 * N;
#pragma unroll
            for (int q = 0; q < PPT; ++q) v[q] = mine ? wk_q_ptr<TPL>(ws, q, tl, wblk)[wlane] : c32{0.f, 0.f};
            line_fft<N, FWD, WAVE>(v, lds, tw, tl);
            // the weights are fetched after the transform, four at a time, to keep the register count of the transform
#pragma unroll
            for (int g = 0; g < PPT; g += 4) {
                float wq[4];
#pragma unroll
                for (int i = 0; i < 4 && g + i < PPT; ++i) wq[i] = mine ? w[off + TPL * (g + i)] : 0.f;
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

}  // namespace p3d
