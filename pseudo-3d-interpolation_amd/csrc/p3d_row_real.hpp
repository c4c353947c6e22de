// p3d_row_real.hpp -- the row-pair passes of float32 cubes (half-spectrum work buffer).
// Part of the two fused passes of one POCS iteration; the overview (pass structure, work-buffer layout) is at the top of p3d_kernels.hpp.
#pragma once

#include "p3d_row_pipe64.hpp"

namespace p3d {

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
// (register budget = what the workgroup's own wavefronts need per SIMD, as in row_pipe64_kernel: rows of 2048 / 4096 samples leave one or two 256- / 512-thread
// workgroups per CU for their LDS anyway -- at a flat "4 waves per SIMD" the 4096-sample kernel spilled 17 registers for an occupancy it cannot have)
__global__ __launch_bounds__((pipe64_threads<N>()), ((pipe64_threads<N>() / 64 + 3) / 4)) void row_real_kernel(const RowArgs a)
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
        const bool in_range = u < total;
        const unsigned uu = in_range ? u : 0u;
        const unsigned slice = uu / pps, pr = uu - slice * pps, ua = 2 * pr;   // ua, ua + 1: the table units of the a- and b-rows
        const unsigned ra = ua * PW + (unsigned)sub;                           // this lane's row a
        int dn = 0;
        if (k_done != nullptr) dn = k_done[slice];
        bool on = in_range;
        if (MODE == REAL_MID) on = on && dn == 0;
        if (MODE == REAL_LAST) on = on && (a.only_done ? (dn > a.only_done_lo && dn <= a.only_done) : dn <= 0);
        if (MODE == REAL_FIRST) on = on && dn == 0;
        // lock-step (adjacent row pairs complete the 128-byte lines of a column block) -- and a trip in which NONE of the workgroup's units has anything
        // to do is skipped as a whole: the "finalize" launch of the early exit (only_done) touches the few slices that have just converged, and ran a full
        // pass's loads and transforms for everything else (float32 job of 50 iterations on the headline cube: 443 instead of 661 it/s)
        if (k_done != nullptr) {
            if (!__syncthreads_or(on ? 1 : 0)) continue;
        } else if (MODE != REAL_FIRST) {
            __syncthreads();
        }
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
            if (a.sums != nullptr) {   // (not for an all-zero slice, dn < 0: it is only handed back -- its work rows hold nothing of this job)
                double da = (double)sa, db = (double)sb;
                constexpr int SEG = TPL >= 64 ? 64 : TPL;
#pragma unroll
                for (int o = SEG / 2; o > 0; o >>= 1) { da += __shfl_down(da, o, SEG); db += __shfl_down(db, o, SEG); }
                if constexpr (WAVE) {
                    if (tl == 0 && on && dn >= 0) { a.sums[(size_t)slice * a.n1 + ra] = da; a.sums[(size_t)slice * a.n1 + ra + PW] = db; }
                } else {   // the wavefronts of a pair, in order
                    __syncthreads();
                    if (lane == 0) { red[2 * wave] = da; red[2 * wave + 1] = db; }
                    __syncthreads();
                    if (tl == 0 && on && dn >= 0) {
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

}  // namespace p3d
