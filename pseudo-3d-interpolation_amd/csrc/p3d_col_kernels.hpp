// p3d_col_kernels.hpp -- the spectrum (column) pass: one-launch kernel (all modes) and the persistent pass with the next tile prefetched.
// Part of the two fused passes of one POCS iteration; the overview (pass structure, work-buffer layout) is at the top of p3d_kernels.hpp.
#pragma once

#include "p3d_kernels_common.hpp"

namespace p3d {

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
        // rows on which Psi_s vanishes were not stored by the spread pass and are not read by the gather pass (ShearArgs::sup)
        unsigned rows_on = 0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) rows_on |= (shear_group_on(a.sh, s, (tl + TPL * q) >> 3) ? 1u : 0u) << q;
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = ((rows_on >> q) & 1u) ? inb[eoff(0, tl + TPL * q)] : c32{0.f, 0.f};
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
            for (int q = 0; q < PPT; ++q)
                if ((rows_on >> q) & 1u) outb[eoff(0, tl + TPL * q)] = v[q];
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
            // half-spectrum buffer of a real cube (ColArgs::herm_n2): column k2 stands for column herm_n2 - k2 (its conjugate, mirrored in
            // k1) as well, unless it is its own mirror image
            const bool twice = a.herm_n2 > 0 && col != 0 && 2 * col != a.herm_n2;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const float p = v[q].x * v[q].x + v[q].y * v[q].y;
                const float vi = twice ? fabsf(v[q].y) : v[q].y;
                if (lex_greater(v[q].x, vi, lr, li)) { lr = v[q].x; li = vi; }
                mx = fmaxf(mx, p);
                mn = fminf(mn, p);
                sq += p;
            }
            if (twice) sq += sq;
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
    // SHEAR: bit q of the result = rows tl + TPL q of shearlet `slice % nsh` belong to a row group on which its spectrum does not vanish
    // (ShearArgs::sup; the 8 rows of a wavefront's register q are one group, so the words come through the scalar path -- a vector
    // load would sit behind the prefetched tile in the in-order vmcnt queue, like the threshold below).  Others: all rows.
    typedef const unsigned __attribute__((address_space(4))) * ksup_t;
    const ksup_t k_sup = (ksup_t)a.sh.sup;
    auto rows_of = [&](const Tile& t) -> unsigned {
        if constexpr (!SHEAR) return 0xffffu;
        if (k_sup == nullptr) return 0xffffu;
        const unsigned sh = t.slice % (unsigned)a.sh.nsh;
        const unsigned g0 = (unsigned)__builtin_amdgcn_readfirstlane(tl >> 3);
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const unsigned g = g0 + (unsigned)(TPL / 8) * q;
            m |= ((k_sup[(size_t)sh * a.sh.sup_words + (g >> 5)] >> (g & 31u)) & 1u) << q;
        }
        return m;
    };
    auto issue = [&](raw64 (&dst)[PPT], const Tile& t) {
        bool valid;
        const unsigned vo = lane_off(t, valid);
        const __amdgpu_buffer_rsrc_t srd = buf_srd(reinterpret_cast<const char*>(a.in) + (size_t)t.slice * slice_bytes, slice_bytes);
        const unsigned on = rows_of(t);
#pragma unroll
        for (int q = 0; q < PPT; ++q) dst[q] = buf_load_raw64(srd, (!SHEAR || ((on >> q) & 1u)) ? vo : BUF_OOB, (unsigned)(TPL * q) * 64u);   // out of range: zero, no access
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
            const unsigned on = rows_of(cur);
#pragma unroll
            for (int q = 0; q < PPT; ++q) buf_store_c32(osrd, (!SHEAR || ((on >> q) & 1u)) ? so_v : BUF_OOB, (unsigned)(TPL * q) * 64u, v[q]);
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

}  // namespace p3d
