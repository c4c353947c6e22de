// p3d_mix.hpp -- the register-resident line-FFT engine for 7-smooth line lengths that are not powers of two, and the two fused POCS passes on it.
//
// numpy.fft.fft2 / ifft2 take every length (cube_POCS_interpolation_3D.py:255-257, called at functions/POCS.py:535, 592, 613) and survey grids
// are rarely powers of two.  p3d_fft.hpp holds a line of 2^m points in registers, 16 per thread; p3d_flex.hip runs any other length as an
// LDS image with one butterfly per thread and pass, radix and strides read at run time (0.27 - 0.40 of the tuned rate).  This file is the
// engine of p3d_fft.hpp for N = R0 R1 ... with radices 2 ... 32 built from 2, 3, 5 and 7:
//
//   * in pass p a line is held by TPL_p = N / PPT_p threads, PPT_p = R_p B_p points each, layout  register q of thread tl <-> element
//     tl + TPL_p q  (global loads / stores coalesce over tl); butterfly s of thread tl is butterfly j = tl + s TPL_p of the Stockham recipe,
//     its R_p inputs in[j + t N / R_p] are the registers s + B_p t -- no data moves before a pass; a pass of radix R (Ns = product of the
//     radices before it) computes  v[t] = in[j + t N/R] w^(t (j mod Ns)),  out[(j div Ns) Ns R + (j mod Ns) + k Ns] = DFT_R(v)[k]  and hands
//     its outputs round through LDS (scatter, barrier, gather in the next pass's layout); the last pass leaves its own layout by itself.
//     The split (B_p) is chosen per pass, and the inverse transform runs the passes in reversed order, so that transforms chain through
//     registers (MixPlan below);
//   * everything about a plan is a compile-time constant (MixPlan<N, PPT, R...>): strides fold into instruction offsets, the division by
//     Ns is a multiply, the small DFTs are straight-line code (Cooley-Tukey on 4 / 2 / 3 / 5 / 7, root constants as literals);
//   * twiddles in LDS: one master table exp(-2 pi i k / N) for the last pass of either direction, ordered rows for the middle passes
//     (MixPlan::build_tw), conjugated inside the multiply for the inverse transform;
//   * LDS image of a line: one padding slot per R0 positions, so that the first scatter (stride R0) walks the banks with an odd stride.
//
// The plans are chosen by tools/gen_mix_plans.py (fewest passes, ~16-24 points per thread in every pass) and listed in p3d_mix_plans.inc; the
// kernels are instantiated per plan in p3d_mix_inst.hip (several translation units) and reached through the launchers of p3d_flex.hip,
// i.e. behind the same LineOps / RowArgs / ColArgs interface and on the same column-blocked work buffer as every other length.
#pragma once

#include "p3d_kernels_common.hpp"
#include "p3d_mix_engine.hpp"
#include "p3d_mix_entry.hpp"

namespace p3d {
namespace mix {

__device__ __forceinline__ bool lex_gt(float ar, float ai, float br, float bi) { return (ar > br) || (ar == br && ai > bi); }

// =========================================================================================================================================
// spectrum (column) pass: forward transform, threshold, inverse transform of a tile of COLT columns (modes as col_kernel / flex_col_kernel)
// =========================================================================================================================================
template <class PL>
__global__ __launch_bounds__(PL::COLT* PL::TMAX) void mix_col_kernel(const ColArgs a, const c32* __restrict__ tab, int mode, int ntiles)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, TMAX = PL::TMAX, T = PL::COLT, THREADS = T * TMAX;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ float red[((THREADS + 63) / 64) * 5];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + PL::TW_SLOTS;

    const int tid = threadIdx.x;
    const int c_lo = tid % T, tl = tid / T;
    const int slice = blockIdx.y;
    // Tiles narrower than a 64-byte column block (long columns): the 8 / T tiles of one block go to workgroups g, g + 8, ..., which the
    // dispatcher places on the same XCD one after the other (workgroup g of a grid whose x extent is a multiple of 8 runs on XCD g % 8):
    // the block's cache lines come from HBM once, the sibling tiles hit that XCD's L2.  The launcher rounds the grid up for this.
    int tile = blockIdx.x;
    if constexpr (T < 8) {
        constexpr int G = 8 / T;
        const int xcd = tile & 7, j = tile >> 3;
        tile = ((j / G) * 8 + xcd) * G + (j % G);
    }
    if (tile >= ntiles) return;
    const int col = tile * T + c_lo;
    const bool valid = col < a.n2;
    if (a.done && a.done[slice] != 0) return;

    for (int i = tid; i < PL::TW_SLOTS; i += THREADS) twl[i] = tab[i];
    __syncthreads();

    c32* const image = data + c_lo;
    const bool iter = mode == COL_ITER || mode == COL_ITER_SOFT || mode == COL_ITER_GARROTE;
    // unconditional buffer accesses; a lane that must not take part (column past the edge, thread outside the layout) carries an offset beyond
    // the descriptor's range
    const size_t in_slice = a.in_std ? (size_t)N * a.n2 : wk_slice_stride(N, a.n2), out_slice = a.out_std ? (size_t)N * a.n2 : wk_slice_stride(N, a.n2);
    const __amdgpu_buffer_rsrc_t isrd = buf_srd(a.in + (size_t)slice * in_slice, (unsigned)(in_slice * sizeof(c32)));
    const __amdgpu_buffer_rsrc_t osrd = buf_srd(a.out + (size_t)slice * out_slice, (unsigned)(out_slice * sizeof(c32)));
    const unsigned blk0 = ((unsigned)(col >> 3) * N) * 8 + (col & 7);   // column-blocked: + row * 8
    const unsigned in_org = a.in_std ? (unsigned)col : blk0, in_pitch = a.in_std ? (unsigned)a.n2 : 8u;
    const unsigned out_org = a.out_std ? (unsigned)col : blk0, out_pitch = a.out_std ? (unsigned)a.n2 : 8u;

    c32 v[VMAX];
#pragma unroll
    for (int q = 0; q < VMAX; ++q) v[q] = c32{0.f, 0.f};
    if (mode == COL_INV) {   // the inverse transform starts in layout B
        const unsigned org = (valid && tl < TPL_B) ? (in_org + (unsigned)tl * in_pitch) * 8u : BUF_OOB;
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) v[q] = buf_load_c32(isrd, org + (unsigned)(TPL_B * q) * in_pitch * 8u, 0u);
    } else {
        const unsigned org = (valid && tl < TPL_A) ? (in_org + (unsigned)tl * in_pitch) * 8u : BUF_OOB;
#pragma unroll
        for (int q = 0; q < PPT_A; ++q) v[q] = buf_load_c32(isrd, org + (unsigned)(TPL_A * q) * in_pitch * 8u, 0u);
        line_fft<PL, FWD, T>(v, image, twl, tl);
    }
    // (layout B from here to the inverse transform)
    const bool live_b = valid && tl < TPL_B;
    const unsigned out_b = live_b ? (out_org + (unsigned)tl * out_pitch) * 8u : BUF_OOB;

    if (iter || (mode == COL_FWD && a.tau != nullptr)) {
        const c32 tau = a.tau[(size_t)slice * a.niter + a.iter];
        const int op = mode == COL_ITER_SOFT ? 1 : (mode == COL_ITER_GARROTE ? 2 : a.op);   // callers pass COL_ITER + a.op
        const Shrink shr(tau, op);
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            v[q] = shr(v[q]);
            bits |= (__float_as_uint(v[q].x) | __float_as_uint(v[q].y)) & 0x7fffffffu;
        }
        if (iter && a.nzflag != nullptr) {
            // a tile the threshold emptied stays zeros after the inverse transform: say so instead of transforming and storing it
            const int kept = __syncthreads_or((live_b && bits != 0u) ? 1 : 0);
            if (tid == 0) a.nzflag[(size_t)slice * ntiles + tile] = kept ? 1 : 0;
            if (!kept) {
                // the row pass skips whole 8-column BLOCKS: an empty tile narrower than a block must leave zeros behind for the case that a
                // sibling tile of its block kept something (see col_kernel)
                if constexpr (T < 8) {
#pragma unroll
                    for (int q = 0; q < PPT_B; ++q) buf_store_c32(osrd, out_b + (unsigned)(TPL_B * q) * out_pitch * 8u, 0u, c32{0.f, 0.f});
                }
                return;
            }
        }
    }

    if (mode == COL_STATS) {
        // lexicographic complex max, max|X|, min|X|, sum|X|^2 of this tile (POCS.py:261-262, 288, 299)
        float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY, sq = 0.f;
        if (live_b) {
#pragma unroll
            for (int q = 0; q < PPT_B; ++q) {
                const float p = v[q].x * v[q].x + v[q].y * v[q].y;
                if (lex_gt(v[q].x, v[q].y, lr, li)) { lr = v[q].x; li = v[q].y; }
                mx = fmaxf(mx, p);
                mn = fminf(mn, p);
                sq += p;
            }
        }
        const int lane = tid & 63;
        const int nact = (THREADS - (tid & ~63)) < 64 ? (THREADS - (tid & ~63)) : 64;   // the last wavefront may be partial
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
            const float omx = __shfl_down(mx, o, 64), omn = __shfl_down(mn, o, 64), osq = __shfl_down(sq, o, 64);
            if (lane + o < nact) {
                if (lex_gt(orr, oi, lr, li)) { lr = orr; li = oi; }
                mx = fmaxf(mx, omx);
                mn = fminf(mn, omn);
                sq += osq;
            }
        }
        const int wave = tid >> 6, nw = (THREADS + 63) >> 6;
        if (lane == 0) {
            red[wave * 5 + 0] = lr; red[wave * 5 + 1] = li; red[wave * 5 + 2] = mx; red[wave * 5 + 3] = mn; red[wave * 5 + 4] = sq;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nw; ++w) {
                if (lex_gt(red[w * 5], red[w * 5 + 1], lr, li)) { lr = red[w * 5]; li = red[w * 5 + 1]; }
                mx = fmaxf(mx, red[w * 5 + 2]);
                mn = fminf(mn, red[w * 5 + 3]);
                sq += red[w * 5 + 4];
            }
            float* p = a.partials + ((size_t)slice * ntiles + tile) * STATS_PARTIAL;
            p[0] = lr; p[1] = li; p[2] = sqrtf(mx); p[3] = sqrtf(mn); p[4] = sq;
        }
        return;
    }

    if (iter || mode == COL_INV) {
        line_fft<PL, INV, T>(v, image, twl, tl);   // -> layout A
        const unsigned out_a = (valid && tl < TPL_A) ? (out_org + (unsigned)tl * out_pitch) * 8u : BUF_OOB;
#pragma unroll
        for (int q = 0; q < PPT_A; ++q) buf_store_c32(osrd, out_a + (unsigned)(TPL_A * q) * out_pitch * 8u, 0u, v[q]);
    } else {
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) buf_store_c32(osrd, out_b + (unsigned)(TPL_B * q) * out_pitch * 8u, 0u, v[q]);
    }
}

// =========================================================================================================================================
// row pass: inverse transform, re-insertion of the observed traces (POCS.py:616-619), cost sums, forward transform -- ROWLB rows per workgroup
// (modes ROW_FIRST / ROW_MID / ROW_LAST as row_kernel / flex_row_kernel)
// =========================================================================================================================================
template <class PL>
__global__ __launch_bounds__(PL::ROWLB* PL::TMAX) void mix_row_kernel(const RowArgs a, const c32* __restrict__ tab, int mode)
{
    constexpr int N = PL::N, VMAX = PL::VMAX, TMAX = PL::TMAX, LB = PL::ROWLB, THREADS = LB * TMAX;
    constexpr int PPT_A = PL::PPT_A, TPL_A = PL::TPL_A, PPT_B = PL::PPT_B, TPL_B = PL::TPL_B;
    // Rows are taken in PAIRS (2 p, 2 p + 1) whose threads alternate lane by lane: the two rows' 64-byte pieces of a column block are
    // neighbours in the column-blocked work buffer, so one load or store instruction of a wavefront then moves whole 128-byte lines (a row
    // alone touches half of every line it reads or writes, the other half moved by another wavefront at another time); in LDS the pair is
    // a two-line tile, element-major, like the column pass's.
    // (A persistent form -- runs of row groups per workgroup, tables copied once, the next group's input requested into registers ahead of the
    // transforms -- was built in round 5: 256 VGPRs, or 800 bytes of scratch per lane at three waves per SIMD; 0.82 -> 1.5 ms.  Dropped.)
    static_assert(LB % 2 == 0, "rows are worked on in pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ uint8_t nzl[N + 8];   // the slice's tile flags, one byte per column tile (a tile is 1 ... 8 columns wide: at most N of them)
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + PL::TW_SLOTS;
    const int tid = threadIdx.x, par = tid & 1, tl = (tid >> 1) % TMAX, pair = (tid >> 1) / TMAX, line = 2 * pair + par;
    const int slice = blockIdx.y, row = blockIdx.x * LB + line;
    const bool valid = row < a.n1;
    const bool live_a = valid && tl < TPL_A, live_b = valid && tl < TPL_B;   // this thread holds elements in layout A / B

    const int dn = a.done ? a.done[slice] : 0;   // (uniform over the workgroup: every early return below is taken by all of its threads)
    // Every access to the cubes and the work buffer is an UNCONDITIONAL buffer instruction; a lane that must not take part carries an offset
    // beyond the descriptor's range (loads return zero, stores are dropped).  One branch per predicated access would end every wait of the
    // unrolled loops at vmcnt(0): twenty exposed memory latencies in a row instead of one (measured: 1.5 -> 0.9 ms for this pass).
    const unsigned esz = a.dtype == 0 ? 8u : 4u;
    const unsigned cube_bytes = (unsigned)a.n1 * N * esz;            // one slice of x / out (< 2 GiB: extents <= 4096)
    const __amdgpu_buffer_rsrc_t xsrd = buf_srd(reinterpret_cast<const char*>(a.x) + (size_t)slice * cube_bytes, cube_bytes);
    const __amdgpu_buffer_rsrc_t osrd = buf_srd(reinterpret_cast<const char*>(a.out) + (size_t)slice * cube_bytes, a.out ? cube_bytes : 0u);
    const __amdgpu_buffer_rsrc_t msrd = buf_srd(a.mask, a.mask ? (unsigned)a.n1 * N * 4u : 0u);
    const unsigned wbytes = (unsigned)(wk_slice_stride(a.n1, N) * sizeof(c32));
    const __amdgpu_buffer_rsrc_t wsrd = buf_srd(a.work + (size_t)slice * wk_slice_stride(a.n1, N), wbytes);
    const unsigned el0 = (unsigned)row * N + tl;                     // element (row, tl) of a row-major slice
    const unsigned xoff = live_a ? el0 * esz : BUF_OOB, moff = live_a ? el0 * 4u : BUF_OOB;
    const unsigned wblk = (unsigned)a.n1 * 8;
    auto woff = [&](int q) -> unsigned {   // byte offset of element (row, tl + TPL_B q) in the column-blocked slice
        const int i = tl + TPL_B * q;
        return ((unsigned)(i >> 3) * wblk + (unsigned)row * 8 + (unsigned)(i & 7)) * 8u;
    };

    if (mode == ROW_LAST && a.only_done) {
        if (dn <= a.only_done_lo || dn > a.only_done) return;
    } else if (mode == ROW_LAST) {
        if (dn > 0) return;   // converged earlier: `out` already holds that iterate
        if (dn < 0) {         // an all-zero slice is handed back untouched (POCS.py:515-521)
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) {
                if (a.dtype == 0) buf_store_c32(osrd, xoff, (unsigned)(TPL_A * q) * 8u, c32{0.f, 0.f});
                else __builtin_amdgcn_raw_buffer_store_b32(0u, osrd, (int)xoff, (int)((unsigned)(TPL_A * q) * 4u), 0);
            }
            return;
        }
    } else if (dn != 0) {
        return;
    }

    // The slice's tile flags go to LDS first (column blocks the column pass found empty were not stored: they read as zeros); the twiddle
    // tables follow BEHIND the requests for the rows (tables_to_lds() below), so that the two latencies run side by side.
    const uint8_t* const nzf = (mode != ROW_FIRST && a.nzflag && !a.only_done) ? a.nzflag + (size_t)slice * a.nz_tiles : nullptr;
    if (nzf) {
        for (int i = tid; i < a.nz_tiles; i += THREADS) nzl[i] = nzf[i];
        __syncthreads();
    }
    auto tables_to_lds = [&]() {
        for (int i = tid; i < PL::TW_SLOTS; i += THREADS) twl[i] = tab[i];
        __syncthreads();
    };
    c32* const image = data + pair * (2 * PL::LINE) + par;

    // binary masks travel as one 64-bit word per thread and row (bit q = mask[row][tl + TPL_A q]); the observed samples of the steady state
    // then come from the COMPACT array the first pass wrote: thread by thread, a thread's samples in q order (RowArgs::mbits / mbase / xc)
    unsigned long long mbits = 0;
    unsigned cbase = 0;
    if (a.mbits != nullptr && live_a) {
        mbits = a.mbits[(size_t)row * TPL_A + tl];
        cbase = a.mbase[(size_t)row * TPL_A + tl];
    }
    const bool compact = a.xc != nullptr && a.mbits != nullptr;
    const __amdgpu_buffer_rsrc_t csrd = buf_srd(reinterpret_cast<const char*>(a.xc) + (size_t)slice * a.nobs * esz, compact ? a.nobs * esz : 0u);
    auto coff = [&](int q) -> unsigned {   // byte offset of this thread's sample q in the slice's compact array (switched off where the trace is missing)
        const unsigned below = (unsigned)__popcll(mbits & ((1ull << q) - 1ull));
        return ((mbits >> q) & 1ull) ? (cbase + below) * esz : BUF_OOB;
    };
    auto load_obs = [&](c32 (&xo)[VMAX]) {   // layout A
        if (compact && mode != ROW_FIRST) {
            if (a.dtype == 0) {
#pragma unroll
                for (int q = 0; q < PPT_A; ++q) xo[q] = buf_load_c32(csrd, coff(q), 0u);
            } else {
#pragma unroll
                for (int q = 0; q < PPT_A; ++q) xo[q] = c32{buf_load_f32(csrd, coff(q), 0u), 0.f};
            }
        } else if (a.dtype == 0) {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) xo[q] = buf_load_c32(xsrd, xoff, (unsigned)(TPL_A * q) * 8u);
        } else {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) xo[q] = c32{buf_load_f32(xsrd, xoff, (unsigned)(TPL_A * q) * 4u), 0.f};
        }
    };
    auto load_mask = [&](float (&mk)[VMAX]) {   // (no mask: the descriptor is empty, every lane reads zero)
        if (a.mbits != nullptr) {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) mk[q] = ((mbits >> q) & 1ull) ? 1.0f : 0.0f;
        } else {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) mk[q] = buf_load_f32(msrd, moff, (unsigned)(TPL_A * q) * 4u);
        }
    };

    float acc = 0.f;
    c32 v[VMAX];
#pragma unroll
    for (int q = 0; q < VMAX; ++q) v[q] = c32{0.f, 0.f};
    if (mode == ROW_FIRST) {
        load_obs(v);
        tables_to_lds();
        if (compact) {
            // the compact copy of the observed samples; a non-zero sample where the mask says "missing" makes it unusable (RowArgs::violation)
            bool viol = false;
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) {
                if (a.dtype == 0) __builtin_amdgcn_raw_buffer_store_b64(p3d_u2{__float_as_uint(v[q].x), __float_as_uint(v[q].y)}, csrd, (int)coff(q), 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].x), csrd, (int)coff(q), 0, 0);
                viol = viol || (!((mbits >> q) & 1ull) && (v[q].x != 0.0f || v[q].y != 0.0f));
            }
            if (viol && a.violation) *a.violation = 1;
        }
        if (a.adaptive) {   // x_old = x at the first iteration (POCS.py:549, 574-575)
            float mk[VMAX];
            load_mask(mk);
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) {
                const c32 x = v[q];
                acc += abs_c32(x);
                const float w = 1.0f - a.alpha * mk[q];
                const c32 blend = x * a.alpha + x * w;
                v[q] = blend + (x - x * mk[q]) * (1.0f - a.alpha);
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) acc += abs_c32(v[q]);
        }
    } else {
        const int tsh = 31 - __builtin_clz((unsigned)a.nz_col_t);
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) {
            bool kept = live_b;
            if (nzf) kept = kept && nzl[(tl + TPL_B * q) >> tsh] != 0;
            v[q] = buf_load_c32(wsrd, kept ? woff(q) : BUF_OOB, 0u);
        }
        tables_to_lds();
        line_fft<PL, INV, 2>(v, image, twl, tl);   // layout B -> layout A
        c32 xo[VMAX];
        float mk[VMAX];
        if (!a.plain) {
            load_obs(xo);
            load_mask(mk);
        } else {
#pragma unroll
            for (int q = 0; q < PPT_A; ++q) { xo[q] = c32{0.f, 0.f}; mk[q] = 0.f; }
        }
        const bool handback = mode == ROW_LAST && a.only_done;
        const bool store_out = mode == ROW_LAST || a.write_out;
#pragma unroll
        for (int q = 0; q < PPT_A; ++q) {
            c32 xn = v[q] * a.scale;
            float m = 0.f;
            if (handback) {
                // the converged iterate up to one row-transform round trip; an observed trace with alpha = 1 IS the observation
                if (a.alpha == 1.0f && (a.mask || a.mbits) && mk[q] == 1.0f) xn = xo[q];
            } else if (!a.plain) {
                m = mk[q];
                const float w = 1.0f - a.alpha * m;        // POCS.py:616
                xn = axpby(xn, w, xo[q], a.alpha);         // POCS.py:619
            }
            acc += abs_c32(xn);
            if (a.dtype == 0) buf_store_c32(osrd, store_out ? xoff : BUF_OOB, (unsigned)(TPL_A * q) * 8u, xn);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xn.x), osrd, (int)(store_out ? xoff : BUF_OOB), (int)((unsigned)(TPL_A * q) * 4u), 0);   // np.real(), POCS.py:656
            if (mode == ROW_MID && a.adaptive) {   // x_input of the next iteration (POCS.py:574-575)
                const float w = 1.0f - a.alpha * m;
                const c32 blend = xo[q] * a.alpha + xn * w;
                v[q] = blend + (xo[q] - xn * m) * (1.0f - a.alpha);
            } else {
                v[q] = xn;
            }
        }
    }
    if (a.sums != nullptr) {
        // per-row sum of |x| in a fixed order: the threads' partial sums go through the pair's LDS image, one thread per row adds them up in double
        __syncthreads();   // (the image is free: the inverse transform's last gather is behind every thread)
        float* part = reinterpret_cast<float*>(data + pair * (2 * PL::LINE)) + par * TMAX;
        part[tl] = live_a ? acc : 0.f;
        __syncthreads();
        if (tl == 0 && valid) {
            double t = 0.0;
            for (int i = 0; i < TPL_A; ++i) t += (double)part[i];
            a.sums[(size_t)slice * a.n1 + row] = t;
        }
    }
    if (mode != ROW_LAST) {
        line_fft<PL, FWD, 2>(v, image, twl, tl);   // layout A -> layout B
#pragma unroll
        for (int q = 0; q < PPT_B; ++q) buf_store_c32(wsrd, live_b ? woff(q) : BUF_OOB, 0u, v[q]);
    }
}

}  // namespace mix
}  // namespace p3d
