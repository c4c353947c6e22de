// p3d_chirp.hip -- the two POCS passes for line lengths with a prime factor above 13 (numpy.fft takes every length and so does the
// reference, cube_POCS_interpolation_3D.py:255-257; functions/POCS.py:535, 592, 613): chirp-z (Bluestein) on the register-resident
// engine of p3d_fft.hpp.
//
// A DFT of n points is a convolution with the chirp c_k = exp(-i pi k^2 / n):  X_k = c_k sum_j (x_j c_j) conj(c)_(k-j).  The
// convolution runs as two transforms of M points, M the power of two >= 2n - 1, through the spectrum of conj c (computed in double
// on the host, flex_build_table).  Round 2 ran those transforms as mixed-radix passes over an LDS image of the padded line
// (p3d_flex.hip); here a line of M points sits in the registers of its M/16 threads exactly like a tuned power-of-two line, so the
// kernels below are the tuned column and row kernels (p3d_col_kernels.hpp, p3d_row_kernels.hpp) with
//   * `chirp_dft` (multiply by the chirp, forward M-point transform, multiply by the spectrum, inverse transform, multiply by the
//     chirp) in the place of `line_fft`,
//   * loads, stores, statistics and the work-buffer layout on the TRUE length n:  n <= M/2, so only registers q < 8 of a thread
//     (elements tl + TPL q < M/2) ever hold samples -- the other eight are literal zeros going into the first pass and are not
//     looked at coming out of the last one.
// Same LineOps / RowArgs / ColArgs contract as p3d_flex.hip, which hands its chirp-z lengths here (flex_row / flex_col).
// Arithmetic is float32; the chirp and its spectrum are rounded once from double.  Results agree with numpy's to the same 1e-6
// relative as the other paths (tests/test_gpu_parity.py: test_fft2_matches_numpy, test_flexible_lengths_against_the_oracle).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "p3d_chirp.hpp"
#include "p3d_kernels.hpp"

namespace p3d {

namespace {

struct ChirpDev {
    const c32* chirp;
    const c32* bhat;
    int n;
    float inv_m;
};

template <int DIR> __device__ __forceinline__ c32 cj(c32 w) { return DIR > 0 ? c32{w.x, -w.y} : w; }

// DFT (DIR = FWD) or unnormalised inverse DFT (INV: conj(DFT(conj x))) of the n-point line held in registers q < PPT/2 (canonical
// layout: register q of thread tl = element tl + TPL q); result in the same registers.  Every thread of the line's workgroup
// (WAVE = false) / wavefront (WAVE = true) must call this.
template <int M, int DIR, bool WAVE, class LDS, class TW>
__device__ __forceinline__ void chirp_dft(c32 (&v)[Plan<M>::PPT], LDS lds, TW tw, int tl, const ChirpDev& t)
{
    constexpr int TPL = Plan<M>::TPL, PPT = Plan<M>::PPT, H = PPT / 2;
    {
        c32 c[H];
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            c[q] = i < t.n ? t.chirp[i] : c32{0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            const c32 p = cj<DIR>(v[q]) * c[q];
            v[q] = i < t.n ? p : c32{0.f, 0.f};
        }
    }
#pragma unroll
    for (int q = H; q < PPT; ++q) v[q] = c32{0.f, 0.f};
    line_fft<M, FWD, WAVE>(v, lds, tw, tl);
    // (the spectrum is fetched in two halves: sixteen of its samples at once would cost the transform's register budget)
#pragma unroll
    for (int g = 0; g < PPT; g += H) {
        c32 b[H];
#pragma unroll
        for (int q = 0; q < H; ++q) b[q] = t.bhat[tl + TPL * (g + q)];
#pragma unroll
        for (int q = 0; q < H; ++q) v[g + q] = v[g + q] * b[q];
    }
    line_fft<M, INV, WAVE>(v, lds, tw, tl);
    {
        c32 c[H];
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            c[q] = i < t.n ? t.chirp[i] : c32{0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            const c32 p = cj<DIR>((v[q] * c[q]) * t.inv_m);
            v[q] = i < t.n ? p : c32{0.f, 0.f};
        }
    }
}

// Launch shapes.  A workgroup holds the padded lines of its T columns / LB rows in LDS between the passes of a transform plus the
// twiddle tables, and a CU should hold at least TWO workgroups (one covers the other's barriers and table loads): transforms of 2048
// points run four columns / two rows per workgroup on the half table of p3d_col_shear.hpp (70 + 11 KiB: two per CU; rows 35 + 11: three),
// transforms of 4096 points one line per workgroup (35 + 20 KiB: two per CU).  Measured on 1009 x 1013 x 128 and 1101 x 1451 x 128:
// profiles/r03_chirp_shape_sweep.txt.
template <int M>
struct ChirpCfg {
    static constexpr bool HALF = M >= 2048;   // ColTablesHalf / TwColHalf in both passes
    static constexpr int T = M == 2048 ? 4 : (M == 4096 ? 1 : col_tile<M>());
    static constexpr int LB = M >= 4096 ? 1 : (M == 2048 ? 2 : ROW_THREADS / Plan<M>::TPL);
    static constexpr int ROW_WAVES = M >= 4096 ? 2 : (M >= 1024 ? 3 : 4);   // waves per SIMD the LDS footprint allows anyway: the register budget follows it
    using ColTab = std::conditional_t<HALF, ColTablesHalf<M>, ColTables<M>>;
    using ColTw = std::conditional_t<HALF, TwColHalf, TwCol>;
    static constexpr int row_slots() { return HALF ? ColTablesHalf<M>::slots() : PassTables<M>::slots(); }
    using RowTw = std::conditional_t<HALF, TwColHalf, TwOrdered>;
    static constexpr size_t col_lds() { return sizeof(c32) * (ColTab::slots() + (size_t)(T / (T < 8 ? T : 8)) * LdsColW<(T < 8 ? T : 8)>::stride(M)); }
    static constexpr size_t row_lds() { return sizeof(c32) * (row_slots() + (size_t)LB * LdsRow::stride(M)) + 32 * sizeof(double); }
};
template <int M, int THREADS>
__device__ __forceinline__ void chirp_load_col_tables(c32* twl, const c32* tab, int tid)
{
    if constexpr (ChirpCfg<M>::HALF) ColTablesHalf<M>::template load<THREADS>(twl, tab, tid);
    else for (int i = tid; i < ColTables<M>::slots(); i += THREADS) twl[i] = tab[i];
}
template <int M, int THREADS>
__device__ __forceinline__ void chirp_load_row_tables(c32* twl, const c32* tab, int tid)   // tab: ColTables image (HALF) or PassTables image
{
    if constexpr (ChirpCfg<M>::HALF) ColTablesHalf<M>::template load<THREADS>(twl, tab, tid);
    else for (int i = tid; i < PassTables<M>::slots(); i += THREADS) twl[i] = tab[i];
}

// ---- column pass ------------------------------------------------------------------------------------------------------------------
// col_kernel (p3d_col_kernels.hpp) on columns of n points: T columns per workgroup, CW = min(T, 8) of them from one 64-byte column
// block.  MODE: COL_ITER (the operator comes in a.op), COL_STATS, COL_FWD (with the threshold when a.tau is given), COL_INV.
template <int M, int MODE>
__global__ __launch_bounds__(ChirpCfg<M>::T * Plan<M>::TPL, P3D_WAVES_PER_EU)
void chirp_col_kernel(const ColArgs a, const c32* __restrict__ tab, const ChirpDev cz)
{
    using PL = Plan<M>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT, H = PPT / 2;
    constexpr int T = ChirpCfg<M>::T;
    constexpr int THREADS = T * TPL;
    constexpr int CW = T < 8 ? T : 8;
    using LDS = LdsColW<CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + ChirpCfg<M>::ColTab::slots();
    const typename ChirpCfg<M>::ColTw tw{twl};
    const int n = cz.n;

    const int tid = threadIdx.x;
    const int c_lo = tid % CW;
    const int tl = (tid / CW) % TPL;
    const int cbl = tid / (CW * TPL);
    const int slice = blockIdx.y;
    int tile = blockIdx.x;
    const int ntiles = (a.n2 + T - 1) / T;
    if constexpr (T < 8) {
        // the tiles of one 64-byte block on one XCD, back to back (see col_kernel); the launcher rounds the grid up to whole groups of
        // 8 x 8/T tiles so that this always applies -- a block fetched by 8/T workgroups on different XCDs comes from HBM 8/T times
        constexpr int G = 8 / T;
        if (P3D_XCD_PAIR && (gridDim.x % (8 * G)) == 0) {
            const int xcd = tile & 7, j = tile >> 3;
            tile = ((j / G) * 8 + xcd) * G + (j % G);
        }
    }
    if (tile >= ntiles) return;
    const int col = tile * T + cbl * CW + c_lo;
    const bool valid = col < a.n2;
    if (a.done && a.done[slice] != 0) return;

    const LDS lds{data + cbl * LDS::stride(M) + c_lo};
    constexpr bool ITER = MODE == COL_ITER;
    const int vcol = valid ? col : 0;
    const bool in_std = !ITER && a.in_std, out_std = !ITER && a.out_std;
    const c32* const inb = a.in + (size_t)slice * (in_std ? (size_t)n * a.n2 : wk_slice_stride(n, a.n2));
    c32* const outb = a.out + (size_t)slice * (out_std ? (size_t)n * a.n2 : wk_slice_stride(n, a.n2));
    const unsigned blk0 = ((unsigned)(vcol >> 3) * (unsigned)n) * 8 + (vcol & 7);   // column-blocked: + row * 8
    const unsigned in_org = in_std ? (unsigned)vcol : blk0, in_pitch = in_std ? (unsigned)a.n2 : 8u;
    const unsigned out_org = out_std ? (unsigned)vcol : blk0, out_pitch = out_std ? (unsigned)a.n2 : 8u;
    // element offsets: 32 bits inside the column-blocked buffer (checked by the launcher); 64 bits for the row-major cubes of the fft2 / time-axis
    // hooks, whose trace count is the caller's (never in the iteration: ITER is compile time)
    using off_t = std::conditional_t<ITER, unsigned, size_t>;

    c32 v[PPT];
#pragma unroll
    for (int q = 0; q < H; ++q) {
        const int r = tl + TPL * q;
        v[q] = r < n ? inb[(off_t)in_org + (off_t)r * in_pitch] : c32{0.f, 0.f};   // (columns past the edge re-read column 0)
    }
    chirp_load_col_tables<M, THREADS>(twl, tab, tid);   // (under the latency of the tile's loads)
    __syncthreads();
    if (MODE != COL_INV) chirp_dft<M, FWD, false>(v, lds, tw, tl, cz);

    if (ITER || (MODE == COL_FWD && a.tau != nullptr)) {
        const c32 tau = a.tau[(size_t)slice * a.niter + a.iter];
        const Shrink shr(tau, a.op);
#pragma unroll
        for (int q = 0; q < H; ++q) v[q] = shr(v[q]);   // (the padding is zero and stays zero)
        if (ITER && a.nzflag != nullptr) {
            // a tile the threshold emptied is all zeros after the inverse transform too: say so instead of transforming and storing it
            unsigned bits = 0;
#pragma unroll
            for (int q = 0; q < H; ++q) bits |= __float_as_uint(v[q].x) | __float_as_uint(v[q].y);
            const int kept = __syncthreads_or(bits != 0u ? 1 : 0);
            if (tid == 0) a.nzflag[(size_t)slice * ntiles + tile] = kept ? 1 : 0;
            if (!kept) {
                // the row pass skips whole 8-column blocks: an empty tile narrower than a block leaves zeros behind for the case that a
                // sibling tile of its block kept something
                if constexpr (T < 8) {
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < H; ++q) {
                            const int r = tl + TPL * q;
                            if (r < n) outb[(off_t)out_org + (off_t)r * out_pitch] = c32{0.f, 0.f};
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
            for (int q = 0; q < H; ++q) {
                if (tl + TPL * q >= n) continue;
                const float p = v[q].x * v[q].x + v[q].y * v[q].y;
                if (lex_greater(v[q].x, v[q].y, lr, li)) { lr = v[q].x; li = v[q].y; }
                mx = fmaxf(mx, p);
                mn = fminf(mn, p);
                sq += p;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {   // (workgroups are whole wavefronts: T TPL >= 256)
            const float orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
            const float omx = __shfl_down(mx, o, 64), omn = __shfl_down(mn, o, 64), osq = __shfl_down(sq, o, 64);
            if (lex_greater(orr, oi, lr, li)) { lr = orr; li = oi; }
            mx = fmaxf(mx, omx);
            mn = fminf(mn, omn);
            sq += osq;
        }
        __syncthreads();   // the line images are free again
        float* red = reinterpret_cast<float*>(data);
        const int wave = tid >> 6, nw = THREADS >> 6;
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
            float* p = a.partials + ((size_t)slice * ntiles + tile) * STATS_PARTIAL;
            p[0] = lr; p[1] = li; p[2] = sqrtf(mx); p[3] = sqrtf(mn); p[4] = sq;
        }
        return;
    }

    if (ITER || MODE == COL_INV) chirp_dft<M, INV, false>(v, lds, tw, tl, cz);

    if (valid) {
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int r = tl + TPL * q;
            if (r < n) outb[(off_t)out_org + (off_t)r * out_pitch] = v[q];
        }
    }
}

// ---- row pass ----------------------------------------------------------------------------------------------------------------------
// row_kernel (p3d_row_kernels.hpp) on rows of n samples with the float weights of the trace mask (no packed words, no compact
// samples: the contract of the flexible row pass).  LB = 256 / TPL rows per workgroup.
template <int M>
constexpr int chirp_row_lines() { return ChirpCfg<M>::LB; }
template <int M>
constexpr int chirp_row_waves() { return ChirpCfg<M>::ROW_WAVES; }
template <int M>
constexpr int chirp_row_threads() { return chirp_row_lines<M>() * Plan<M>::TPL; }
template <int M>
constexpr size_t chirp_row_lds() { return ChirpCfg<M>::row_lds(); }

template <int M, int MODE>
__global__ __launch_bounds__(chirp_row_threads<M>(), chirp_row_waves<M>()) void chirp_row_kernel(const RowArgs a, const c32* __restrict__ tab, const ChirpDev cz)
{
    using PL = Plan<M>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT, H = PPT / 2;
    constexpr int LB = chirp_row_lines<M>(), THREADS = chirp_row_threads<M>();
    constexpr int LSTR = LdsRow::stride(M);
    constexpr bool WAVE = TPL <= 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + ChirpCfg<M>::row_slots();
    const typename ChirpCfg<M>::RowTw tw{twl};
    const int n = cz.n;

    const int tid = threadIdx.x;
    const int line = tid / TPL;
    const int tl = tid - line * TPL;
    const int slice = blockIdx.y, row = blockIdx.x * LB + line;
    const bool valid = row < a.n1;
    const int vrow = valid ? row : 0;
    const size_t sbase = ((size_t)slice * a.n1 + vrow) * n;   // row-major cubes (x, out)

    const int dn = a.done ? a.done[slice] : 0;
    if (MODE == ROW_LAST && a.only_done) {
        if (dn <= a.only_done_lo || dn > a.only_done) return;
    } else if (MODE == ROW_LAST) {
        if (dn > 0) return;   // converged earlier: `out` already holds that iterate
        if (dn < 0) {         // all-zero slice is handed back untouched (POCS.py:515-521)
            if (valid) {
#pragma unroll
                for (int q = 0; q < H; ++q) {
                    const int i = tl + TPL * q;
                    if (i >= n) continue;
                    if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[sbase + i] = c32{0.f, 0.f};
                    else reinterpret_cast<float*>(a.out)[sbase + i] = 0.f;
                }
            }
            return;
        }
    } else if (dn != 0) {
        return;
    }

    const LdsRow lds{data + line * LSTR};
    c32* const wslice = a.work + (size_t)slice * wk_slice_stride(a.n1, n);
    const unsigned wblk = (unsigned)a.n1 * 8;
    const unsigned wlane = wk_lane_off<TPL>(tl, vrow, wblk);
    auto obs_at = [&](int i) -> c32 {
        if (a.dtype == 0) return reinterpret_cast<const c32*>(a.x)[sbase + i];
        return c32{reinterpret_cast<const float*>(a.x)[sbase + i], 0.f};
    };
    const float* const mrow = a.mask ? a.mask + (size_t)vrow * n : nullptr;

    c32 v[PPT];
    float acc = 0.f;
    if (MODE == ROW_FIRST) {
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            const c32 x = (valid && i < n) ? obs_at(i) : c32{0.f, 0.f};
            acc += abs_c32(x);
            if (a.adaptive) {   // x_old = x at the first iteration (POCS.py:549, 574-575)
                const float m = (mrow && i < n) ? mrow[i] : 0.f;
                const float w = 1.0f - a.alpha * m;
                const c32 blend = x * a.alpha + x * w;
                v[q] = blend + (x - x * m) * (1.0f - a.alpha);
            } else {
                v[q] = x;
            }
        }
        chirp_load_row_tables<M, THREADS>(twl, tab, tid);   // (under the latency of the row's loads)
        __syncthreads();
    } else {
        // column blocks the column pass found empty were not stored (RowArgs::nzflag): they read as zeros.  The flags first, all of
        // them, then the loads they allow, then the copy of the tables: three latencies one after the other otherwise
        const uint8_t* const nzf = (a.nzflag && !a.only_done) ? a.nzflag + (size_t)slice * a.nz_tiles : nullptr;
        const int tsh = 31 - __builtin_clz((unsigned)a.nz_col_t);
        unsigned keep = 0;
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            bool kept = valid && i < n;
            if (nzf && kept) kept = nzf[i >> tsh] != 0;
            keep |= (kept ? 1u : 0u) << q;
        }
#pragma unroll
        for (int q = 0; q < H; ++q) v[q] = ((keep >> q) & 1u) ? wk_q_ptr<TPL>(wslice, q, tl, wblk)[wlane] : c32{0.f, 0.f};
        chirp_load_row_tables<M, THREADS>(twl, tab, tid);
        __syncthreads();
        // observed samples and weights: requested in front of the inverse transform and used after it where the registers are there
        // (transforms of 2048 points and more run at 128 registers per thread: they fetch them afterwards, the other workgroup covers)
        constexpr bool EARLY = M < 2048;
        c32 xo[H];
        float mk[H];
        const bool need_obs = !a.plain && valid;
        auto fetch = [&] {
#pragma unroll
            for (int q = 0; q < H; ++q) {
                const int i = tl + TPL * q;
                xo[q] = (need_obs && i < n) ? obs_at(i) : c32{0.f, 0.f};
                mk[q] = (mrow && !a.plain && i < n) ? mrow[i] : 0.f;
            }
        };
        if constexpr (EARLY) fetch();
        chirp_dft<M, INV, WAVE>(v, lds, tw, tl, cz);
        if constexpr (!EARLY) {
            __builtin_amdgcn_sched_barrier(0);   // (not hoisted into the transform)
            fetch();
        }
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            c32 xn = v[q] * a.scale;
            float m = 0.f;
            if (MODE == ROW_LAST && a.only_done) {
                // the converged iterate up to one row-transform round trip; an observed trace with alpha = 1 IS the observation
                if (a.alpha == 1.0f && mrow && mk[q] == 1.0f) xn = xo[q];
            } else if (!a.plain) {
                m = mk[q];
                const float w = 1.0f - a.alpha * m;        // POCS.py:616
                xn = axpby(xn, w, xo[q], a.alpha);         // POCS.py:619
            }
            acc += abs_c32(xn);   // (zero beyond the row's end)
            if ((MODE == ROW_LAST || a.write_out) && valid && i < n) {
                if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[sbase + i] = xn;
                else reinterpret_cast<float*>(a.out)[sbase + i] = xn.x;   // np.real(), POCS.py:656
            }
            if (MODE == ROW_MID) {
                if (a.adaptive) {   // x_input of the next iteration (POCS.py:574-575)
                    const float w = 1.0f - a.alpha * m;
                    const c32 blend = xo[q] * a.alpha + xn * w;
                    v[q] = blend + (xo[q] - xn * m) * (1.0f - a.alpha);
                } else {
                    v[q] = xn;
                }
            }
        }
    }

    if (a.sums != nullptr) {
        double ws = valid ? (double)acc : 0.0;
        if constexpr (TPL <= 64) {
#pragma unroll
            for (int o = TPL / 2; o > 0; o >>= 1) ws += __shfl_down(ws, o, TPL);
            if (tl == 0 && valid) a.sums[(size_t)slice * a.n1 + row] = ws;
        } else {
            ws = wave_sum(ws);
            double* red = reinterpret_cast<double*>(data + LB * LSTR);   // scratch behind the line buffers
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
        chirp_dft<M, FWD, WAVE>(v, lds, tw, tl, cz);
        if (valid) {
#pragma unroll
            for (int q = 0; q < H; ++q)
                if (tl + TPL * q < n) wk_q_ptr<TPL>(wslice, q, tl, wblk)[wlane] = v[q];
        }
    }
}

// Real (float32) cubes with the hard operator (row_real_kernel, flex_row_real_kernel): rows 2p and 2p + 1 share one complex transform,
// z = r_a + i r_b, and the work buffer holds columns 0 ... n/2 of the two row spectra.  The split needs Z[k] next to Z[n - k]: the
// forward transform's result goes through the line's LDS image once.
template <int M, int MODE>
__global__ __launch_bounds__(chirp_row_threads<M>(), chirp_row_waves<M>()) void chirp_row_real_kernel(const RowArgs a, const c32* __restrict__ tab, const ChirpDev cz)
{
    using PL = Plan<M>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT, H = PPT / 2;
    constexpr int LB = chirp_row_lines<M>(), THREADS = chirp_row_threads<M>();
    constexpr int LSTR = LdsRow::stride(M);
    constexpr bool WAVE = TPL <= 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + ChirpCfg<M>::row_slots();
    const typename ChirpCfg<M>::RowTw tw{twl};
    const int n = cz.n, Hn = n / 2;

    const int tid = threadIdx.x;
    const int line = tid / TPL;
    const int tl = tid - line * TPL;
    const int slice = blockIdx.y, pair = blockIdx.x * LB + line;
    const bool valid = 2 * pair + 1 < a.n1;
    const int ra = valid ? 2 * pair : 0;
    const size_t sbase = ((size_t)slice * a.n1 + ra) * n;   // row a of the row-major cubes; row b follows
    const float* const xa = reinterpret_cast<const float*>(a.x) + sbase;
    float* const oa = reinterpret_cast<float*>(a.out) + sbase;

    const int dn = a.done ? a.done[slice] : 0;
    if (MODE == ROW_LAST && a.only_done) {
        if (dn <= a.only_done_lo || dn > a.only_done) return;
    } else if (MODE == ROW_LAST) {
        if (dn > 0) return;
        if (dn < 0) {
            if (valid) {
#pragma unroll
                for (int q = 0; q < H; ++q) {
                    const int i = tl + TPL * q;
                    if (i < n) { oa[i] = 0.f; oa[n + i] = 0.f; }
                }
            }
            return;
        }
    } else if (dn != 0) {
        return;
    }

    const LdsRow lds{data + line * LSTR};
    c32* const wrow = a.work + (size_t)slice * wk_slice_stride(a.n1, Hn + 1) + (size_t)ra * 8;   // row a; row b is 8 elements on
    const size_t wblk = (size_t)a.n1 * 8;
    const float* const ma = a.mask ? a.mask + (size_t)ra * n : nullptr;

    c32 v[PPT];
    float sa = 0.f, sb = 0.f;
    if (MODE == ROW_FIRST) {
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            const bool on = valid && i < n;
            const float va = on ? xa[i] : 0.f, vb = on ? xa[n + i] : 0.f;
            sa += fabsf(va);
            sb += fabsf(vb);
            v[q] = c32{va, vb};
        }
        chirp_load_row_tables<M, THREADS>(twl, tab, tid);
        __syncthreads();
    } else {
        const uint8_t* const nzf = (a.nzflag && !a.only_done) ? a.nzflag + (size_t)slice * a.nz_tiles : nullptr;
        const int tsh = 31 - __builtin_clz((unsigned)a.nz_col_t);
        unsigned keep = 0;   // (flags, then loads, then tables: see chirp_row_kernel)
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            const int k = i <= Hn ? i : n - i;
            bool kept = valid && i < n;
            if (nzf && kept) kept = nzf[k >> tsh] != 0;
            keep |= (kept ? 1u : 0u) << q;
        }
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            const int k = i <= Hn ? i : n - i;
            const bool kept = ((keep >> q) & 1u) != 0;
            c32 r0{0.f, 0.f}, r1{0.f, 0.f};
            if (kept) {
                const c32* w = wrow + (size_t)(k >> 3) * wblk + (k & 7);
                r0 = w[0];
                r1 = w[8];
            }
            if (i > Hn) { r0.y = -r0.y; r1.y = -r1.y; }
            if (k == 0 || 2 * k == n) { r0.y = 0.f; r1.y = 0.f; }   // self-mirrored columns of a real row are real
            v[q] = c32{r0.x - r1.y, r0.y + r1.x};
        }
        chirp_load_row_tables<M, THREADS>(twl, tab, tid);
        __syncthreads();
        constexpr bool EARLY = M < 2048;   // (see chirp_row_kernel)
        float xoa[H], xob[H], mka[H], mkb[H];
        auto fetch = [&] {
#pragma unroll
            for (int q = 0; q < H; ++q) {
                const int i = tl + TPL * q;
                const bool on = valid && i < n;
                xoa[q] = on ? xa[i] : 0.f;
                xob[q] = on ? xa[n + i] : 0.f;
                mka[q] = (ma && i < n) ? ma[i] : 0.f;
                mkb[q] = (ma && i < n) ? ma[n + i] : 0.f;
            }
        };
        if constexpr (EARLY) fetch();
        chirp_dft<M, INV, WAVE>(v, lds, tw, tl, cz);
        if constexpr (!EARLY) {
            __builtin_amdgcn_sched_barrier(0);   // (not hoisted into the transform)
            fetch();
        }
        const bool handback = MODE == ROW_LAST && a.only_done != 0;
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            const c32 z = v[q] * a.scale;
            float va = z.x, vb = z.y;
            if (handback) {
                if (a.alpha == 1.0f && mka[q] == 1.0f) va = xoa[q];
                if (a.alpha == 1.0f && mkb[q] == 1.0f) vb = xob[q];
            } else {
                va = __builtin_fmaf(va, 1.0f - a.alpha * mka[q], xoa[q] * a.alpha);   // POCS.py:616-619
                vb = __builtin_fmaf(vb, 1.0f - a.alpha * mkb[q], xob[q] * a.alpha);
            }
            sa += fabsf(va);
            sb += fabsf(vb);
            if (MODE == ROW_LAST && valid && i < n) { oa[i] = va; oa[n + i] = vb; }
            if (MODE == ROW_MID) v[q] = c32{va, vb};
        }
    }
    if (a.sums != nullptr) {
        double da = valid ? (double)sa : 0.0, db = valid ? (double)sb : 0.0;
        if constexpr (TPL <= 64) {
#pragma unroll
            for (int o = TPL / 2; o > 0; o >>= 1) { da += __shfl_down(da, o, TPL); db += __shfl_down(db, o, TPL); }
            if (tl == 0 && valid) { a.sums[(size_t)slice * a.n1 + ra] = da; a.sums[(size_t)slice * a.n1 + ra + 1] = db; }
        } else {
            da = wave_sum(da);
            db = wave_sum(db);
            double* red = reinterpret_cast<double*>(data + LB * LSTR);
            __syncthreads();
            if ((tid & 63) == 0) { red[2 * (tid >> 6)] = da; red[2 * (tid >> 6) + 1] = db; }
            __syncthreads();
            if (tl == 0 && valid) {
                double ta = 0.0, tb = 0.0;
                for (int w = 0; w < TPL / 64; ++w) { ta += red[2 * (line * (TPL / 64) + w)]; tb += red[2 * (line * (TPL / 64) + w) + 1]; }
                a.sums[(size_t)slice * a.n1 + ra] = ta;
                a.sums[(size_t)slice * a.n1 + ra + 1] = tb;
            }
            __syncthreads();
        }
    }
    if (MODE != ROW_LAST) {
        chirp_dft<M, FWD, WAVE>(v, lds, tw, tl, cz);
        exchange_sync<WAVE>();   // everybody is done with the transform's last exchange
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const int i = tl + TPL * q;
            if (i < n) lds.at(i) = v[q];
        }
        exchange_sync<WAVE>();
        if (valid) {
#pragma unroll
            for (int q = 0; q <= H / 2; ++q) {   // k <= n/2 < M/4
                const int k = tl + TPL * q;
                if (k > Hn) continue;
                const c32 z = lds.at(k), pz = lds.at(k == 0 ? 0 : n - k);
                c32* w = wrow + (size_t)(k >> 3) * wblk + (k & 7);
                w[0] = c32{0.5f * (z.x + pz.x), 0.5f * (z.y - pz.y)};      // R_a = (Z[k] + conj Z[n-k]) / 2
                w[8] = c32{0.5f * (z.y + pz.y), -0.5f * (z.x - pz.x)};     // R_b = (Z[k] - conj Z[n-k]) / 2i
            }
        }
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------------------------
template <int M>
hipError_t launch_chirp_col(int mode, const ColArgs& a, const ChirpTabs& t, hipStream_t st)
{
    constexpr int T = ChirpCfg<M>::T;
    const ChirpDev cz{t.chirp, t.bhat, t.n, 1.0f / (float)M};
    unsigned gx = (unsigned)((a.n2 + T - 1) / T);
    if (T < 8 && P3D_XCD_PAIR) { const unsigned g = 8u * (8u / T); gx = (gx + g - 1) / g * g; }   // (see the kernel: surplus workgroups leave at once)
    const dim3 grid(gx, a.nslices);
    constexpr size_t lds = ChirpCfg<M>::col_lds();
    static_assert(lds >= sizeof(float) * 5 * 16, "scratch of the statistics");
    hipError_t e = hipSuccess;
    ColArgs c = a;
#define P3D_CHIRP_COL(MODE)                                                                         \
    do {                                                                                            \
        if ((e = allow_lds(chirp_col_kernel<M, MODE>, lds)) != hipSuccess) return e;                \
        chirp_col_kernel<M, MODE><<<grid, T * Plan<M>::TPL, lds, st>>>(c, t.coltab, cz);            \
    } while (0)
    switch (mode) {
        case COL_ITER: P3D_CHIRP_COL(COL_ITER); break;
        case COL_ITER_SOFT: c.op = 1; P3D_CHIRP_COL(COL_ITER); break;
        case COL_ITER_GARROTE: c.op = 2; P3D_CHIRP_COL(COL_ITER); break;
        case COL_STATS: P3D_CHIRP_COL(COL_STATS); break;
        case COL_FWD: P3D_CHIRP_COL(COL_FWD); break;
        case COL_INV: P3D_CHIRP_COL(COL_INV); break;
        default: return hipErrorNotSupported;
    }
#undef P3D_CHIRP_COL
    return hipGetLastError();
}

template <int M, bool REAL>
hipError_t launch_chirp_row(int mode, const RowArgs& a, const ChirpTabs& t, hipStream_t st)
{
    constexpr int LB = chirp_row_lines<M>();
    const ChirpDev cz{t.chirp, t.bhat, t.n, 1.0f / (float)M};
    const int lines = REAL ? a.n1 / 2 : a.n1;
    const dim3 grid((lines + LB - 1) / LB, a.nslices);
    constexpr size_t lds = chirp_row_lds<M>();
    const c32* const rtab = ChirpCfg<M>::HALF ? t.coltab : t.rowtab;   // (the half table is cut from the column pass's image)
    hipError_t e = hipSuccess;
#define P3D_CHIRP_ROW(MODE)                                                                                     \
    do {                                                                                                        \
        if constexpr (REAL) {                                                                                   \
            if ((e = allow_lds(chirp_row_real_kernel<M, MODE>, lds)) != hipSuccess) return e;                   \
            chirp_row_real_kernel<M, MODE><<<grid, chirp_row_threads<M>(), lds, st>>>(a, rtab, cz);         \
        } else {                                                                                                \
            if ((e = allow_lds(chirp_row_kernel<M, MODE>, lds)) != hipSuccess) return e;                        \
            chirp_row_kernel<M, MODE><<<grid, chirp_row_threads<M>(), lds, st>>>(a, rtab, cz);              \
        }                                                                                                       \
    } while (0)
    switch (mode) {
        case ROW_FIRST: P3D_CHIRP_ROW(ROW_FIRST); break;
        case ROW_MID: P3D_CHIRP_ROW(ROW_MID); break;
        case ROW_LAST: P3D_CHIRP_ROW(ROW_LAST); break;
        default: return hipErrorNotSupported;
    }
#undef P3D_CHIRP_ROW
    return hipGetLastError();
}

#define P3D_CHIRP_SIZES(X) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096)

}  // namespace

bool chirp_supported(int m)
{
    switch (m) {
#define P3D_CASE(MM) case MM: return true;
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: return false;
    }
}

int chirp_col_tile(int m)
{
    switch (m) {
#define P3D_CASE(MM) case MM: return ChirpCfg<MM>::T;
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: return 0;
    }
}

size_t chirp_rowtab_slots(int m)
{
    switch (m) {
#define P3D_CASE(MM) case MM: return (size_t)PassTables<MM>::slots();
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: return 0;
    }
}

size_t chirp_table_slots(int m)
{
    switch (m) {
#define P3D_CASE(MM) case MM: return (size_t)PassTables<MM>::slots() + (size_t)ColTables<MM>::slots();
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: return 0;
    }
}

void chirp_build_tables(int m, c32* out)
{
    switch (m) {
#define P3D_CASE(MM) case MM: PassTables<MM>::build(out); ColTables<MM>::build(out + PassTables<MM>::slots()); break;
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: break;
    }
}

hipError_t chirp_col(int mode, const ColArgs& a, const ChirpTabs& t, hipStream_t st)
{
    if (t.n < 2 || 2 * t.n > t.m || (double)wk_slice_stride(t.n, a.n2) >= 4294967296.0) return hipErrorNotSupported;   // 32-bit element offsets inside a work slice
    switch (t.m) {
#define P3D_CASE(MM) case MM: return launch_chirp_col<MM>(mode, a, t, st);
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: return hipErrorNotSupported;
    }
}

hipError_t chirp_row(int mode, const RowArgs& a, const ChirpTabs& t, hipStream_t st)
{
    if (t.n < 2 || 2 * t.n > t.m || (double)wk_slice_stride(a.n1, t.n) >= 4294967296.0) return hipErrorNotSupported;   // (32-bit element offsets inside a work slice)
    switch (t.m) {
#define P3D_CASE(MM) case MM: return launch_chirp_row<MM, false>(mode, a, t, st);
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: return hipErrorNotSupported;
    }
}

hipError_t chirp_row_real(int mode, const RowArgs& a, const ChirpTabs& t, hipStream_t st)
{
    if (t.n < 2 || 2 * t.n > t.m) return hipErrorNotSupported;
    switch (t.m) {
#define P3D_CASE(MM) case MM: return launch_chirp_row<MM, true>(mode, a, t, st);
        P3D_CHIRP_SIZES(P3D_CASE)
#undef P3D_CASE
        default: return hipErrorNotSupported;
    }
}

}  // namespace p3d
