// p3d_col_shear.hpp -- the column pass of a SHEARLET iteration for float32 cubes: TWO columns per complex transform.
// Part of the fused passes of the SHEARLET loop (p3d_shearlet.hip); the general form is COL_SHRINK of col_kernel / col_pipe_kernel.
//
// For a real slice and symmetric real spectra (Psi_s(-k) = Psi_s(k): FFST's realCoefficients=True) the coefficients
// c_s = ifft2(Psi_s fft2(x)) are real, and the reference keeps their real part (FFST returns ST.real).  A column of the work buffer,
// W[k1] (row-transformed, still a spectrum along k1), is then Hermitian in k1, and two columns A, B travel through ONE complex
// transform each way:
//     Z = W_A + i W_B   --inverse-->   z = c_A + i c_B   (threshold Re and Im separately)   --forward-->   Z' = W'_A + i W'_B
//     W'_A[k] = (Z'[k] + conj Z'[N-k]) / 2,      W'_B[k] = (Z'[k] - conj Z'[N-k]) / (2 i).
// Half the transforms per byte, and a workgroup of four packed lines needs half the LDS of an 8-column tile: two workgroups share a CU
// at 2048-point columns where the one-transform-per-column pass has room for one (one covers the other's loads and barriers).
// The imaginary rounding residue of c_A that the general pass drops lands in Re c_B here (and vice versa): the two passes agree to
// float32 rounding (1e-7 of the peak), not bit for bit -- P3D_SHEARLET_NO_PAIR=1 selects the general pass.
// A thread loads / stores the pair (A, B) of a row as one 16-byte access; the eight columns of a tile are one 64-byte column block.
#pragma once

#include "p3d_kernels_common.hpp"

namespace p3d {

// Column-pass tables with HALF the plain table: exp(-2 pi i k / N) for k < N/2 (the other half is its negative), then the ordered
// rows of the middle pass as in ColTables.  7 KiB less at N = 2048 -- what lets two 4-line workgroups fit a CU's 160 KiB.
template <int N>
struct ColTablesHalf {
    using PL = Plan<N>;
    static constexpr int master() { return N / 2; }
    static constexpr int mid_off(int dir) { return master() + (dir == FWD ? 0 : ColTables<N>::mid_len(FWD)); }
    static constexpr int slots() { return master() + ColTables<N>::mid_len(FWD) + ColTables<N>::mid_len(INV); }
    // copy from the plan's full ColTables image (device) into LDS
    template <int THREADS>
    static __device__ __forceinline__ void load(c32* lds, const c32* full, int tid)
    {
        for (int i = tid; i < master(); i += THREADS) lds[i] = full[i];
        constexpr int NM = ColTables<N>::mid_len(FWD) + ColTables<N>::mid_len(INV);
        for (int i = tid; i < NM; i += THREADS) lds[master() + i] = full[ColTables<N>::master() + i];
    }
};
struct TwColHalf {
    const c32* tw;
    template <int N, int DIR, int P, int T>
    __device__ __forceinline__ c32 mul(c32 a, int jm) const
    {
        using PL = Plan<N>;
        if constexpr (P + 1 == PL::NPASS) {
            const int k = T * jm;
            c32 w = tw[k & (N / 2 - 1)];
            if (k & (N / 2)) { w.x = -w.x; w.y = -w.y; }
            return DIR > 0 ? mul_conj(a, w) : a * w;
        } else {
            return a * (tw + (ColTablesHalf<N>::mid_off(DIR) + (T - 1) * PL::ns(DIR, P)))[jm];
        }
    }
};

template <int N>
constexpr size_t col_shear_pair_lds() { return sizeof(c32) * (ColTablesHalf<N>::slots() + (size_t)LdsColW<4>::stride(N)); }
template <int N>
constexpr int col_shear_pair_wgs_per_cu()
{
    constexpr int by_lds = (int)((160 * 1024) / col_shear_pair_lds<N>()), by_waves = 16 / (4 * Plan<N>::TPL / 64 > 0 ? 4 * Plan<N>::TPL / 64 : 1);
    return by_lds < 1 ? 1 : (by_lds < by_waves ? by_lds : by_waves);
}

// grid: (tiles of 8 columns, nb * nsh work slices); 4 * TPL threads
// STATS: the statistics pass of a job instead (p3d_shearlet_stats): inverse transform only, then signed maximum, max c^2, min c^2 and
// sum c^2 of the tile's 8 x N real coefficients -> a.partials[(work slice, tile)][STATS_PARTIAL]
template <int N, bool STATS = false>
__global__ __launch_bounds__(4 * Plan<N>::TPL, (4 * Plan<N>::TPL / 64) * col_shear_pair_wgs_per_cu<N>() / 4 > 0 ? (4 * Plan<N>::TPL / 64) * col_shear_pair_wgs_per_cu<N>() / 4 : 1)
void col_shear_pair_kernel(const ColArgs a)
{
    using PL = Plan<N>;
    constexpr int TPL = PL::TPL, PPT = PL::PPT, THREADS = 4 * TPL;
    static_assert(PPT == 16 && TPL >= 32, "columns of 512 points and more");
    using LDS = LdsColW<4>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c32* twl = reinterpret_cast<c32*>(smem_raw);
    c32* data = twl + ColTablesHalf<N>::slots();
    const TwColHalf tw{twl};
    const int tid = threadIdx.x, j = tid & 3, tl = tid >> 2;
    const int slice = blockIdx.y, tile = blockIdx.x;
    const int b = slice / a.sh.nsh, s = slice - b * a.sh.nsh;
    // Rows on which Psi_s vanishes were not stored by the spread pass and are not read by the gather pass (ShearArgs::sup).  With
    // Hermitian work slices (ShearArgs::half) row k > N/2 is the conjugate of row N - k: a thread loads THAT row (bit of its group).
    // The bitmap words of the shearlet are wave-uniform and come through the scalar path with compile-time word indices, so that the
    // tile's loads can be issued right away -- the copy of the tables into LDS then runs under their latency.
    //   row k = tl + TPL q, group (tl >> 3) + G q, G = TPL / 8 (a divisor or a multiple of 32: the word depends on q alone);
    //   mirrored row N - k, group N/8 - G q - c with c = (tl + 7) >> 3: word of N/8 - G (q + 1) for c >= 1, of N/8 - G q for tl = 0.
    constexpr int G = TPL / 8, NG = N / 8, WORDS = (NG + 31) / 32;
    unsigned supw[WORDS];
#pragma unroll
    for (int w = 0; w < WORDS; ++w) supw[w] = a.sh.sup ? a.sh.sup[(size_t)s * a.sh.sup_words + w] : 0xffffffffu;
    const bool half = a.sh.half != 0;
    const int c8 = (tl + 7) >> 3;
    unsigned rows_on = 0, mirrored = 0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const unsigned direct = (supw[(G * q) >> 5] >> (((G * q) & 31) + (tl >> 3))) & 1u;
        unsigned bit = direct;
        if (q >= PPT / 2) {
            const bool mir = half && !(q == PPT / 2 && tl == 0);
            const int gz = NG - G * q > 0 ? NG - G * q : 0, gl = NG - G * (q + 1) > 0 ? NG - G * (q + 1) : 0;   // (tl = 0 / tl > 0; constants once the loop is unrolled)
            const unsigned m0 = (supw[(gz >> 5) < WORDS ? (gz >> 5) : 0] >> (gz & 31)) & 1u;
            const unsigned m1 = (supw[gl >> 5] >> ((NG - G * q - c8) & 31)) & 1u;
            if (mir) bit = tl == 0 ? m0 : m1;
            mirrored |= (mir ? 1u : 0u) << q;
        }
        rows_on |= bit << q;
    }
    c32 tau{0.f, 0.f};
    if constexpr (!STATS) tau = a.sh.tau[((size_t)b * a.sh.niter + a.sh.iter) * a.sh.nsh + s];

    const LDS lds{data + j};
    c32* const base = a.out + (size_t)slice * wk_slice_stride(N, a.n2);   // in place (a.in == a.out)
    const unsigned org = ((unsigned)tile * N + (unsigned)tl) * 8u + 2u * (unsigned)j;   // element (row tl, column pair j) of the tile's block; + TPL q rows
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 raw[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        raw[q] = f4{0.f, 0.f, 0.f, 0.f};
        if ((rows_on >> q) & 1u) {
            const int k = tl + TPL * q;
            const unsigned off = ((mirrored >> q) & 1u) ? ((unsigned)tile * N + (unsigned)(N - k)) * 8u + 2u * (unsigned)j : org + (unsigned)(TPL * q) * 8u;
            raw[q] = *reinterpret_cast<const f4*>(base + off);
        }
    }
    ColTablesHalf<N>::template load<THREADS>(twl, a.tw, tid);
    __syncthreads();
    c32 v[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        f4 ab = raw[q];
        if ((mirrored >> q) & 1u) { ab.y = -ab.y; ab.w = -ab.w; }   // row N - k holds the conjugates
        v[q] = add_ib(c32{ab.x, ab.y}, c32{ab.z, ab.w});             // Z = W_A + i W_B
    }
    line_fft<N, INV, false>(v, lds, tw, tl);
    if constexpr (STATS) {
        const float scale = 1.0f / ((float)N * (float)a.n2);
        float smax = -INFINITY, mx = 0.f, mn = INFINITY;
        double sq = 0.0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const float ca = v[q].x * scale, cb = v[q].y * scale;     // the two columns' real coefficients
            smax = fmaxf(smax, fmaxf(ca, cb));
            const float pa = ca * ca, pb = cb * cb;
            mx = fmaxf(mx, fmaxf(pa, pb));
            mn = fminf(mn, fminf(pa, pb));
            sq += (double)(pa + pb);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            smax = fmaxf(smax, __shfl_down(smax, o, 64));
            mx = fmaxf(mx, __shfl_down(mx, o, 64));
            mn = fminf(mn, __shfl_down(mn, o, 64));
            sq += __shfl_down(sq, o, 64);
        }
        __syncthreads();   // the line image is free: partial results of the wavefronts go through it
        double* red = reinterpret_cast<double*>(data);
        constexpr int NW = THREADS / 64;
        if ((tid & 63) == 0) { red[(tid >> 6) * 4 + 0] = smax; red[(tid >> 6) * 4 + 1] = mx; red[(tid >> 6) * 4 + 2] = mn; red[(tid >> 6) * 4 + 3] = sq; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w) {
                smax = fmaxf(smax, (float)red[w * 4]);
                mx = fmaxf(mx, (float)red[w * 4 + 1]);
                mn = fminf(mn, (float)red[w * 4 + 2]);
                sq += red[w * 4 + 3];
            }
            float* p = a.partials + ((size_t)slice * gridDim.x + tile) * STATS_PARTIAL;
            p[0] = smax; p[1] = 0.f; p[2] = mx; p[3] = mn; p[4] = (float)sq;
        }
        return;
    }
    {
        const Shrink shr(tau, a.sh.op);
        const float scale = 1.0f / ((float)N * (float)a.n2);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {   // the operators of the general pass on (c, 0): c_A = Re z, c_B = Im z
            const c32 ca = shr(c32{v[q].x * scale, 0.f}), cb = shr(c32{v[q].y * scale, 0.f});
            v[q] = c32{ca.x, cb.x};
        }
    }
    line_fft<N, FWD, false>(v, lds, tw, tl);
    // Z'[N - k] sits in another thread of the line: through the LDS image once more (canonical positions)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PPT; ++q) lds.at(tl + TPL * q) = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (((rows_on & ~mirrored) >> q) & 1u) {    // (Hermitian work slices: rows 0 ... N/2 only)
            const int k = tl + TPL * q;
            const c32 p = lds.at((N - k) & (N - 1));
            const c32 z = v[q];
            // W'_A = (Z + conj P) / 2,  W'_B = (Z - conj P) / (2 i)
            const f4 o{0.5f * (z.x + p.x), 0.5f * (z.y - p.y), 0.5f * (z.y + p.y), 0.5f * (p.x - z.x)};
            *reinterpret_cast<f4*>(base + org + (unsigned)(TPL * q) * 8u) = o;
        }
    }
}

// hipErrorNotSupported where the pass does not apply (the caller takes the general one)
template <int N>
hipError_t launch_col_shear_pair(const ColArgs& a, hipStream_t st)
{
    if constexpr (Plan<N>::PPT == 16 && Plan<N>::TPL >= 32 && 4 * Plan<N>::TPL <= 1024) {
        const bool stats = a.partials != nullptr;     // the statistics pass of a job (no thresholds yet)
        if ((!stats && a.sh.tau == nullptr) || !a.sh.real_only || a.in != a.out || a.in_std || a.out_std || a.n2 % 8 != 0) return hipErrorNotSupported;
        if ((double)wk_slice_stride(N, a.n2) >= 4294967296.0 / 8.0) return hipErrorNotSupported;
        constexpr size_t lds = col_shear_pair_lds<N>();
        hipError_t e = hipSuccess;
        const dim3 grid(a.n2 / 8, a.nslices);
        if (stats) {
            if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(col_shear_pair_kernel<N, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
            col_shear_pair_kernel<N, true><<<grid, 4 * Plan<N>::TPL, lds, st>>>(a);
        } else {
            if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(col_shear_pair_kernel<N, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
            col_shear_pair_kernel<N, false><<<grid, 4 * Plan<N>::TPL, lds, st>>>(a);
        }
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

}  // namespace p3d
