// p3d_flex.hip -- the two-pass POCS pipeline for line lengths that are NOT powers of two.
//
// numpy.fft accepts every length and so does the reference (cube_POCS_interpolation_3D.py:255-257); survey grids are rarely
// powers of two.  The register-resident engine of p3d_fft.hpp is specialised per power-of-two length at compile time; this file
// provides the same two kernels -- column pass (forward transform, threshold, inverse transform) and row pass (inverse
// transform, re-insertion, cost sums, forward transform) -- for any length whose lines fit LDS, behind the same LineOps /
// RowArgs / ColArgs interface and on the same column-blocked work buffer, so that every caller in p3d_api.hip (POCS loop,
// statistics, fft2 hooks, early exit, APOCS) runs unchanged and a slice may mix a tuned axis with a flexible one.
//
// Line FFT: mixed-radix Stockham autosort in LDS.  The factor list AND everything a pass derives from it (strides, twiddle step,
// the multiplier of its one division) are computed on the host and sit behind the line's twiddle table (FlexFactors); radices are
// in-register butterflies: 2 ... 16 folded from two factors where possible, 11 and 13 on their own (odd radices first: the strided
// writes of a pass with small stride then have an odd stride in banks); lengths with a larger prime factor run in the chirp-z form on a
// power of two -- NOT here: flex_row / flex_col hand them to p3d_chirp.hip (round 3) -- or, where that does not apply, as direct O(p^2) passes.  The direction of a transform is a template parameter, the first pass
// multiplies by no twiddles.  The twiddle table exp(-2 pi i k / n) sits in LDS.
//   column pass: a workgroup owns T columns (one 64-byte column block for T = 8); element i of column c lives at X[i*T + c],
//                which is exactly the tile's layout in the work buffer: loads and stores are linear copies.  Long columns (one
//                workgroup per CU) run persistent workgroups that request the next tile into registers ahead of the transforms.
//   row pass:    one, two or four wavefronts per row, LB rows per workgroup; in place (one LDS buffer per row, one butterfly per
//                thread and pass) wherever every pass fits the row's threads, two buffers otherwise.
// 7-smooth lengths with a compile-time plan (p3d_mix_plans.inc) leave this file at the launchers: flex_row / flex_col hand them to the mixed-radix
// register engine of p3d_mix.hpp (round 5; 1.4 - 1.6 x the rate of the LDS-image passes below), whose tables sit behind this file's in the device table.
// Arithmetic is float32 like the tuned path; results agree with it to rounding (different pass structure).
// What the round-1 version of this file did wrong, and the measurements: profiles/r02_flex_shape_sweep.txt.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "p3d_chirp.hpp"
#include "p3d_flex.hpp"
#include "p3d_kernels.hpp"
#include "p3d_mix_entry.hpp"

namespace p3d {

namespace {

constexpr int FLEX_MAX_PASSES = 12;   // radices are folded into 16 ... 2 where possible: lengths up to GEN_MAX_N need at most 6

struct FlexFactors {
    int n;      // line length
    int m;      // length of the transforms that are actually run: n, or the power of two M >= 2n-1 of the chirp-z form
    int blue;   // 1: chirp-z (Bluestein) -- lines whose length has a large prime factor
    int nf;
    int f[FLEX_MAX_PASSES];     // radices of the passes of a length-m transform
    // what a pass needs besides its radix R, computed once on the host (the device code used to divide for them in every pass,
    // and the compiler hoisted the divisions of all thirteen radix cases in front of the dispatch):
    int ns[FLEX_MAX_PASSES];    // product of the earlier radices = distance of a butterfly's outputs
    int nb[FLEX_MAX_PASSES];    // butterflies per line, m / R = distance of a butterfly's inputs
    int ts[FLEX_MAX_PASSES];    // twiddle step m / (ns R)
    unsigned mg[FLEX_MAX_PASSES];   // j / ns = umulhi(j, mg) for j < 2^16 (ns > 1)
};

int largest_prime_factor(int n)
{
    int best = 1;
    for (int p = 2; (long)p * p <= n; ++p)
        while (n % p == 0) { best = p; n /= p; }
    return n > 1 ? n : best;
}

constexpr int FLEX_REGISTER_PRIME_MAX = 13;   // prime factors up to here: in-register butterflies
constexpr int FLEX_DIRECT_PRIME_MAX = 23;     // larger prime factors: chirp-z on a power of two instead of an O(p^2) pass

// Radices of the passes.  Every pass is one trip through LDS and one barrier, so two prime factors are folded into one
// in-register butterfly wherever a supported product exists (16 = 4x4 ... 6 = 2x3); 11 and 13 are in-register butterflies of
// their own (paired form, (p - 1)^2 / 2 real multiply-adds per output pair: a 13-point pass costs about 1.5 16-point passes --
// the chirp-z form costs six transforms' worth of passes on twice the length).  17 ... 23 as in-register butterflies take
// 160 - 256 registers, and a kernel is allocated for its hungriest case: they stay direct O(p^2) passes or go chirp-z.
// Odd radices go first: the scattered writes of an early pass (small stride) then have an odd stride in LDS banks.
FlexFactors flex_factors(int n)
{
    FlexFactors p{};
    p.n = p.m = n;
    if (gen_make_plan(n).nf <= 0) { p.nf = -1; return p; }
    const int lp = largest_prime_factor(n);
    if (lp > FLEX_REGISTER_PRIME_MAX) {
        // the convolution length of the chirp-z form: the power of two >= 2n - 1, on which the register-resident engine runs
        // (p3d_chirp.hip).  Columns beyond 1024 points leave two columns per workgroup there: primes 17 ... 23 stay direct O(p^2)
        // passes for those.
        int M = 1;
        while (M < 2 * n - 1) M *= 2;
        if (M < 64) M = 64;
        if (chirp_supported(M) && (M <= 2048 || lp > FLEX_DIRECT_PRIME_MAX)) { p.blue = 1; p.m = M; }
    }
    int rem = p.m, k = 0;
    int radices[32];
    for (int r : {16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2})
        while (rem % r == 0 && rem > 1) { radices[k++] = r; rem /= r; }
    for (int q = 11; rem > 1; q += 2)
        while (rem % q == 0) { radices[k++] = q; rem /= q; }
    if (k > FLEX_MAX_PASSES) { p.nf = -1; return p; }
    p.nf = 0;
    for (int i = 0; i < k; ++i) if (radices[i] % 2) p.f[p.nf++] = radices[i];
    for (int i = 0; i < k; ++i) if (radices[i] % 2 == 0) p.f[p.nf++] = radices[i];
    int ns = 1;
    for (int i = 0; i < p.nf; ++i) {
        const int R = p.f[i];
        p.ns[i] = ns;
        p.nb[i] = p.m / R;
        p.ts[i] = p.m / (ns * R);
        p.mg[i] = ns > 1 ? (unsigned)((0x100000000ull + (unsigned)ns - 1) / (unsigned)ns) : 0u;   // ceil(2^32 / ns): exact for j ns < 2^32
        ns *= R;
    }
    return p;
}

constexpr size_t FLEX_LDS_MAX = 150 * 1024;
constexpr size_t FLEX_LDS_TWO = 80 * 1024;   // two workgroups per CU

size_t col_lds(int n, int T) { const size_t L = flex_factors(n).m; return sizeof(c32) * (2 * T * L + L); }
size_t row_lds(int n, int LB) { const size_t L = flex_factors(n).m; return sizeof(c32) * (2 * LB * L + L); }

int pick_col_tile(int n)
{
    const FlexFactors pl = flex_factors(n);
    if (pl.blue) return chirp_col_tile(pl.m);   // chirp-z lengths: the kernels of p3d_chirp.hip
    if (const mix::Entry* e = mix::find(n)) return e->col_tile;   // 7-smooth lengths: the register engine of p3d_mix.hpp
    if (col_lds(n, 8) <= FLEX_LDS_MAX) return 8;   // a whole 64-byte column block, even at one workgroup per CU (4-column tiles
                                                   // with two workgroups per CU measured 10-18 % slower)
    for (int T : {4, 2, 1}) if (col_lds(n, T) <= FLEX_LDS_MAX) return T;
    return 0;
}
size_t row_lds_inplace(int n, int LB) { const size_t L = flex_factors(n).m; return sizeof(c32) * (LB * L + L); }
// in-place passes (flex_pass_inplace): in-register radices only, at most one butterfly per thread in every pass
bool flex_inplace_ok(const FlexFactors& pl, int tpr)
{
    if (pl.nf <= 0) return false;
    for (int p = 0; p < pl.nf; ++p) {
        const int R = pl.f[p];
        if (R > 16) return false;
        if (pl.m / R > tpr) return false;
    }
    return true;
}
int pick_row_lines(int n)   // rows per workgroup at one wavefront per row
{
    if (flex_factors(n).blue) return 1;   // (p3d_chirp.hip has its own launch shapes)
    for (int LB : {4, 2, 1}) if (row_lds(n, LB) <= FLEX_LDS_TWO) return LB;
    return row_lds(n, 1) <= FLEX_LDS_MAX ? 1 : 0;
}

// ---- butterflies --------------------------------------------------------------------------------------------------------------
// the direction of a transform is a template parameter (DIR = FWD: exp(-...), INV: exp(+...)): as a run-time value it cost a
// select per twiddle and per butterfly constant
template <int DIR> __device__ __forceinline__ c32 conj_if(c32 w) { return DIR > 0 ? c32{w.x, -w.y} : w; }

// exp(2 pi i q / R) for the in-register butterflies: compile-time constants (they used to be read from the twiddle table once per
// pass: wave-uniform values that the compiler parked in scalar registers -- and spilled, one v_readlane per use)
template <int R>
struct Roots;
template <>
struct Roots<3> {
    static constexpr float c[3] = {1.000000000e+00f, -5.000000000e-01f, -5.000000000e-01f};
    static constexpr float s[3] = {0.000000000e+00f, 8.660254040e-01f, -8.660254040e-01f};
};
template <>
struct Roots<5> {
    static constexpr float c[5] = {1.000000000e+00f, 3.090169940e-01f, -8.090169940e-01f, -8.090169940e-01f, 3.090169940e-01f};
    static constexpr float s[5] = {0.000000000e+00f, 9.510565160e-01f, 5.877852520e-01f, -5.877852520e-01f, -9.510565160e-01f};
};
template <>
struct Roots<6> {
    static constexpr float c[6] = {1.000000000e+00f, 5.000000000e-01f, -5.000000000e-01f, -1.000000000e+00f, -5.000000000e-01f, 5.000000000e-01f};
    static constexpr float s[6] = {0.000000000e+00f, 8.660254040e-01f, 8.660254040e-01f, 1.224646800e-16f, -8.660254040e-01f, -8.660254040e-01f};
};
template <>
struct Roots<7> {
    static constexpr float c[7] = {1.000000000e+00f, 6.234898020e-01f, -2.225209340e-01f, -9.009688680e-01f, -9.009688680e-01f, -2.225209340e-01f, 6.234898020e-01f};
    static constexpr float s[7] = {0.000000000e+00f, 7.818314820e-01f, 9.749279120e-01f, 4.338837390e-01f, -4.338837390e-01f, -9.749279120e-01f, -7.818314820e-01f};
};
template <>
struct Roots<8> {
    static constexpr float c[8] = {1.000000000e+00f, 7.071067810e-01f, 6.123234000e-17f, -7.071067810e-01f, -1.000000000e+00f, -7.071067810e-01f, -1.836970200e-16f, 7.071067810e-01f};
    static constexpr float s[8] = {0.000000000e+00f, 7.071067810e-01f, 1.000000000e+00f, 7.071067810e-01f, 1.224646800e-16f, -7.071067810e-01f, -1.000000000e+00f, -7.071067810e-01f};
};
template <>
struct Roots<9> {
    static constexpr float c[9] = {1.000000000e+00f, 7.660444430e-01f, 1.736481780e-01f, -5.000000000e-01f, -9.396926210e-01f, -9.396926210e-01f, -5.000000000e-01f, 1.736481780e-01f, 7.660444430e-01f};
    static constexpr float s[9] = {0.000000000e+00f, 6.427876100e-01f, 9.848077530e-01f, 8.660254040e-01f, 3.420201430e-01f, -3.420201430e-01f, -8.660254040e-01f, -9.848077530e-01f, -6.427876100e-01f};
};
template <>
struct Roots<10> {
    static constexpr float c[10] = {1.000000000e+00f, 8.090169940e-01f, 3.090169940e-01f, -3.090169940e-01f, -8.090169940e-01f, -1.000000000e+00f, -8.090169940e-01f, -3.090169940e-01f, 3.090169940e-01f, 8.090169940e-01f};
    static constexpr float s[10] = {0.000000000e+00f, 5.877852520e-01f, 9.510565160e-01f, 9.510565160e-01f, 5.877852520e-01f, 1.224646800e-16f, -5.877852520e-01f, -9.510565160e-01f, -9.510565160e-01f, -5.877852520e-01f};
};
template <>
struct Roots<11> {
    static constexpr float c[11] = {1.000000000e+00f, 8.412535328e-01f, 4.154150130e-01f, -1.423148383e-01f, -6.548607339e-01f, -9.594929736e-01f, -9.594929736e-01f, -6.548607339e-01f, -1.423148383e-01f, 4.154150130e-01f, 8.412535328e-01f};
    static constexpr float s[11] = {0.000000000e+00f, 5.406408175e-01f, 9.096319954e-01f, 9.898214419e-01f, 7.557495744e-01f, 2.817325568e-01f, -2.817325568e-01f, -7.557495744e-01f, -9.898214419e-01f, -9.096319954e-01f, -5.406408175e-01f};
};
template <>
struct Roots<12> {
    static constexpr float c[12] = {1.000000000e+00f, 8.660254040e-01f, 5.000000000e-01f, 6.123234000e-17f, -5.000000000e-01f, -8.660254040e-01f, -1.000000000e+00f, -8.660254040e-01f, -5.000000000e-01f, -1.836970200e-16f, 5.000000000e-01f, 8.660254040e-01f};
    static constexpr float s[12] = {0.000000000e+00f, 5.000000000e-01f, 8.660254040e-01f, 1.000000000e+00f, 8.660254040e-01f, 5.000000000e-01f, 1.224646800e-16f, -5.000000000e-01f, -8.660254040e-01f, -1.000000000e+00f, -8.660254040e-01f, -5.000000000e-01f};
};
template <>
struct Roots<13> {
    static constexpr float c[13] = {1.000000000e+00f, 8.854560257e-01f, 5.680647467e-01f, 1.205366803e-01f, -3.546048870e-01f, -7.485107482e-01f, -9.709418174e-01f, -9.709418174e-01f, -7.485107482e-01f, -3.546048870e-01f, 1.205366803e-01f, 5.680647467e-01f, 8.854560257e-01f};
    static constexpr float s[13] = {0.000000000e+00f, 4.647231720e-01f, 8.229838659e-01f, 9.927088741e-01f, 9.350162427e-01f, 6.631226582e-01f, 2.393156643e-01f, -2.393156643e-01f, -6.631226582e-01f, -9.350162427e-01f, -9.927088741e-01f, -8.229838659e-01f, -4.647231720e-01f};
};
template <>
struct Roots<14> {
    static constexpr float c[14] = {1.000000000e+00f, 9.009688680e-01f, 6.234898020e-01f, 2.225209340e-01f, -2.225209340e-01f, -6.234898020e-01f, -9.009688680e-01f, -1.000000000e+00f, -9.009688680e-01f, -6.234898020e-01f, -2.225209340e-01f, 2.225209340e-01f, 6.234898020e-01f, 9.009688680e-01f};
    static constexpr float s[14] = {0.000000000e+00f, 4.338837390e-01f, 7.818314820e-01f, 9.749279120e-01f, 9.749279120e-01f, 7.818314820e-01f, 4.338837390e-01f, 1.224646800e-16f, -4.338837390e-01f, -7.818314820e-01f, -9.749279120e-01f, -9.749279120e-01f, -7.818314820e-01f, -4.338837390e-01f};
};
template <>
struct Roots<15> {
    static constexpr float c[15] = {1.000000000e+00f, 9.135454580e-01f, 6.691306060e-01f, 3.090169940e-01f, -1.045284630e-01f, -5.000000000e-01f, -8.090169940e-01f, -9.781476010e-01f, -9.781476010e-01f, -8.090169940e-01f, -5.000000000e-01f, -1.045284630e-01f, 3.090169940e-01f, 6.691306060e-01f, 9.135454580e-01f};
    static constexpr float s[15] = {0.000000000e+00f, 4.067366430e-01f, 7.431448250e-01f, 9.510565160e-01f, 9.945218950e-01f, 8.660254040e-01f, 5.877852520e-01f, 2.079116910e-01f, -2.079116910e-01f, -5.877852520e-01f, -8.660254040e-01f, -9.945218950e-01f, -9.510565160e-01f, -7.431448250e-01f, -4.067366430e-01f};
};
template <>
struct Roots<16> {
    static constexpr float c[16] = {1.000000000e+00f, 9.238795330e-01f, 7.071067810e-01f, 3.826834320e-01f, 6.123234000e-17f, -3.826834320e-01f, -7.071067810e-01f, -9.238795330e-01f, -1.000000000e+00f, -9.238795330e-01f, -7.071067810e-01f, -3.826834320e-01f, -1.836970200e-16f, 3.826834320e-01f, 7.071067810e-01f, 9.238795330e-01f};
    static constexpr float s[16] = {0.000000000e+00f, 3.826834320e-01f, 7.071067810e-01f, 9.238795330e-01f, 1.000000000e+00f, 9.238795330e-01f, 7.071067810e-01f, 3.826834320e-01f, 1.224646800e-16f, -3.826834320e-01f, -7.071067810e-01f, -9.238795330e-01f, -1.000000000e+00f, -9.238795330e-01f, -7.071067810e-01f, -3.826834320e-01f};
};
// W^q with W = exp(DIR * 2 pi i / R)
template <int R, int DIR>
__device__ __forceinline__ c32 root(int q) { return c32{Roots<R>::c[q], DIR > 0 ? Roots<R>::s[q] : -Roots<R>::s[q]}; }

// ---- small DFTs on registers: X[k] = sum_t x[t] W^(t k), W = exp(DIR * 2 pi i / R) -------------------------------------------
template <int R, int DIR>
__device__ __forceinline__ void dft_small(c32 (&x)[R])
{
    if constexpr (R == 2) {
        const c32 a = x[0] + x[1], b = x[0] - x[1];
        x[0] = a; x[1] = b;
    } else if constexpr (R == 4) {
        // b +- DIR i (x1 - x3): one packed add each, the rotation carried by the operand selects (add_ib / sub_ib, p3d_fft.hpp)
        const c32 a = x[0] + x[2], b = x[0] - x[2], s = x[1] + x[3], t = x[1] - x[3];
        x[0] = a + s; x[2] = a - s;
        x[1] = DIR > 0 ? add_ib(b, t) : sub_ib(b, t);
        x[3] = DIR > 0 ? sub_ib(b, t) : add_ib(b, t);
    } else {
        // odd prime: pair x[q] with x[R-q].  W^(qk) = cos + i (DIR sin) gives  X[k], X[R-k] = A_k +- i B_k  with
        // A_k = x0 + sum_q cos(2 pi q k / R) (x[q] + x[R-q]),  B_k = sum_q (DIR sin(2 pi q k / R)) (x[q] - x[R-q]):
        // real coefficients only, a third of the multiplications of the direct form
        constexpr int H = (R - 1) / 2;
        c32 sp[H], dm[H];
        c32 x0 = x[0];
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            sp[q - 1] = x[q] + x[R - q];
            dm[q - 1] = x[q] - x[R - q];
            x0 = x0 + sp[q - 1];
        }
        const c32 xin = x[0];
        x[0] = x0;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            c32 A = xin, B{0.f, 0.f};
#pragma unroll
            for (int q = 1; q <= H; ++q) {
                const c32 wq = root<R, DIR>((q * k) % R);
                A = A + sp[q - 1] * wq.x;
                B = B + dm[q - 1] * wq.y;
            }
            x[k] = add_ib(A, B);        // A + i B
            x[R - k] = sub_ib(A, B);    // A - i B
        }
    }
}

// R = R1 * R2 in registers: t = R2 t1 + t2, k = k1 + R1 k2:  X[k] = sum_t2 [ (sum_t1 x[t] W_R1^(t1 k1)) W_R^(t2 k1) ] W_R2^(t2 k2)
template <int R1, int R2, int DIR>
__device__ __forceinline__ void radix_apply(c32 (&v)[R1 * R2])
{
    constexpr int R = R1 * R2;
    if constexpr (R2 == 1) {
        dft_small<R, DIR>(v);
    } else {
        c32 u[R];   // u[k1 * R2 + t2]
#pragma unroll
        for (int t2 = 0; t2 < R2; ++t2) {
            c32 a[R1];
#pragma unroll
            for (int t1 = 0; t1 < R1; ++t1) a[t1] = v[R2 * t1 + t2];
            dft_small<R1, DIR>(a);
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) u[k1 * R2 + t2] = (k1 * t2) % R == 0 ? a[k1] : a[k1] * root<R, DIR>((k1 * t2) % R);
        }
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            c32 b[R2];
#pragma unroll
            for (int t2 = 0; t2 < R2; ++t2) b[t2] = u[k1 * R2 + t2];
            dft_small<R2, DIR>(b);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = b[k2];
        }
    }
}

// What a pass is told (FlexFactors, one entry per pass): output stride ns, butterflies per line nb, twiddle step ts, the
// multiplier of the division by ns.  The values pass through an empty asm so that what is derived from them (R - 1 input
// offsets, R output offsets ...) is computed where it is used and not, for all thirteen radix cases, in front of the dispatch.
struct PassArgs {
    int ns, nb, ts;
    unsigned mg;
};
__device__ __forceinline__ PassArgs pass_args(const FlexFactors& pl, int p)
{
    PassArgs a{pl.ns[p], pl.nb[p], pl.ts[p], pl.mg[p]};
    return a;
}
__device__ __forceinline__ void pin(PassArgs& a)
{
    // (uniform values the compiler may have parked in vector registers: back to scalar ones first)
    a.ns = __builtin_amdgcn_readfirstlane(a.ns);
    a.nb = __builtin_amdgcn_readfirstlane(a.nb);
    a.ts = __builtin_amdgcn_readfirstlane(a.ts);
    a.mg = (unsigned)__builtin_amdgcn_readfirstlane((int)a.mg);
    asm volatile("" : "+s"(a.ns), "+s"(a.nb), "+s"(a.ts), "+s"(a.mg));
}

// Addressing of the lines a pass works on.  COLS: a column tile, 2^tsh interleaved lines (element i of line l at X[(i << tsh) + l]),
// butterfly index b = (j << tsh) + l.  Otherwise ONE contiguous line (the caller's pointers are those of its line), b = j.
// One pass of radix R = R1 R2: butterfly j reads in[j + t nb], multiplies by w^(t jm ts), writes out[j0 + k ns] with
// jq = j / ns, jm = j mod ns, j0 = jq ns R + jm.  FIRST (ns = 1): jm = 0, all twiddles are 1.
template <int R1, int R2, int DIR, bool FIRST, bool COLS>
__device__ __forceinline__ void flex_pass(const c32* A, c32* B, const c32* tw, PassArgs pa, int tsh, int first, int step)
{
    constexpr int R = R1 * R2;
    pin(pa);
    const int total = COLS ? pa.nb << tsh : pa.nb;
    const int istr = COLS ? 1 << tsh : 1;
    for (int b = first; b < total; b += step) {
        const int j = COLS ? b >> tsh : b, l = COLS ? b & (istr - 1) : 0;
        int j0 = j * R, jm = 0;
        if (!FIRST) {
            const int jq = (int)__umulhi((unsigned)j, pa.mg);
            jm = j - jq * pa.ns;
            j0 = jq * pa.ns * R + jm;
        }
        const c32* in = A + (COLS ? (j << tsh) + l : j);
        c32* out = B + (COLS ? (j0 << tsh) + l : j0);
        const int mstride = COLS ? pa.nb << tsh : pa.nb, ostride = COLS ? pa.ns << tsh : pa.ns;
        c32 v[R];
        v[0] = in[0];
        if (FIRST) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = in[t * mstride];
        } else {
            const int twi = jm * pa.ts;
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = DIR > 0 ? mul_conj(in[t * mstride], tw[t * twi]) : in[t * mstride] * tw[t * twi];
        }
        radix_apply<R1, R2, DIR>(v);
#pragma unroll
        for (int k = 0; k < R; ++k) out[k * ostride] = v[k];
    }
}

// SYNC: 0 = workgroup barrier, 1 = wavefront-level (one wave owns its lines)
template <int SYNC>
__device__ __forceinline__ void flex_sync()
{
    if constexpr (SYNC == 0) __syncthreads();
    else exchange_sync<true>();
}

// The same pass IN PLACE for one line with at most one butterfly per thread (nb <= step): every thread reads the inputs of its
// butterfly into registers, the line's threads synchronise, then the results overwrite the line.  No second buffer: half the
// LDS per row, twice the resident waves of the row pass (which is latency-bound at 8 waves per CU).
template <int R1, int R2, int DIR, bool FIRST, int SYNC>
__device__ __forceinline__ void flex_pass_inplace(c32* A, const c32* tw, PassArgs pa, int first)
{
    constexpr int R = R1 * R2;
    pin(pa);
    c32 v[R];
    const bool act = first < pa.nb;
    const int jc = act ? first : 0;
    int j0 = jc * R, jm = 0;
    if (!FIRST) {
        const int jq = (int)__umulhi((unsigned)jc, pa.mg);
        jm = jc - jq * pa.ns;
        j0 = jq * pa.ns * R + jm;
    }
    {
        const c32* in = A + jc;
        v[0] = in[0];
        if (FIRST) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = in[t * pa.nb];
        } else {
            const int twi = jm * pa.ts;
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = DIR > 0 ? mul_conj(in[t * pa.nb], tw[t * twi]) : in[t * pa.nb] * tw[t * twi];
        }
    }
    flex_sync<SYNC>();   // every input of the pass is in registers
    if (act) {
        radix_apply<R1, R2, DIR>(v);
        c32* out = A + j0;
#pragma unroll
        for (int k = 0; k < R; ++k) out[k * pa.ns] = v[k];
    }
}

// the in-register radices: R -> (R1, R2)
#define P3D_FLEX_RADICES(X) X(2, 2, 1) X(3, 3, 1) X(4, 4, 1) X(5, 5, 1) X(6, 2, 3) X(7, 7, 1) X(8, 2, 4) X(9, 3, 3) X(10, 2, 5) X(12, 3, 4) X(14, 2, 7) X(15, 3, 5) X(16, 4, 4)

// (force-inlined, like everything that is handed the FlexFactors: a call that survives until late makes the compiler copy the
// kernel argument to scratch memory, and its per-pass entries are then read back with vector loads)
template <int SYNC, int DIR>
__device__ __forceinline__ void flex_fft_inplace(c32* A, const c32* tw, const FlexFactors& pl, int first)
{
    // the first pass has no twiddles (ns = 1)
    switch (pl.f[0]) {
#define P3D_FLEX_CASE(R, R1, R2) case R: flex_pass_inplace<R1, R2, DIR, true, SYNC>(A, tw, pass_args(pl, 0), first); break;
        P3D_FLEX_RADICES(P3D_FLEX_CASE)
        P3D_FLEX_CASE(11, 11, 1)
        P3D_FLEX_CASE(13, 13, 1)
#undef P3D_FLEX_CASE
        default: break;   // never: flex_inplace_ok() admits the radices above only
    }
    flex_sync<SYNC>();
    for (int p = 1; p < pl.nf; ++p) {
        switch (pl.f[p]) {
#define P3D_FLEX_CASE(R, R1, R2) case R: flex_pass_inplace<R1, R2, DIR, false, SYNC>(A, tw, pass_args(pl, p), first); break;
            P3D_FLEX_RADICES(P3D_FLEX_CASE)
            P3D_FLEX_CASE(11, 11, 1)
            P3D_FLEX_CASE(13, 13, 1)
#undef P3D_FLEX_CASE
            default: break;
        }
        flex_sync<SYNC>();
    }
}

// flex_transform (below) for one line with in-place passes
template <int SYNC, int DIR>
__device__ __forceinline__ void flex_transform_inplace(c32* A, const c32* tw, const c32*, const FlexFactors& pl, int first, int)
{
    flex_fft_inplace<SYNC, DIR>(A, tw, pl, first);
}

// large prime factor R: direct O(R^2) butterflies, inputs re-read from LDS
template <int DIR, bool COLS>
__device__ __forceinline__ void flex_pass_prime(const c32* A, c32* B, const c32* tw, int R, int n, PassArgs pa, int tsh, int first, int step)
{
    const int total = COLS ? pa.nb << tsh : pa.nb;
    const int istr = COLS ? 1 << tsh : 1;
    for (int b = first; b < total; b += step) {
        const int j = COLS ? b >> tsh : b, l = COLS ? b & (istr - 1) : 0;
        const int jq = pa.ns > 1 ? (int)__umulhi((unsigned)j, pa.mg) : j, jm = j - jq * pa.ns;
        const int j0 = jq * pa.ns * R + jm;
        for (int k = 0; k < R; ++k) {
            c32 acc{0.f, 0.f};
            for (int t = 0; t < R; ++t) {
                const long idx = ((long)t * jm * pa.ts + (long)((long)t * k % R) * pa.nb) % n;
                acc = acc + A[(size_t)(j + t * pa.nb) * istr + l] * conj_if<DIR>(tw[idx]);
            }
            B[(size_t)(j0 + k * pa.ns) * istr + l] = acc;
        }
    }
}

// All passes of one line or of a tile of interleaved lines in LDS (see flex_pass).  first/step: butterfly indices handled by this
// thread.  Returns the buffer that holds the result.  SYNC: 0 = workgroup barrier, 1 = wavefront-level (one wave owns its line).
// BIGP: with the 11- and 13-point in-register butterflies.  They need ~140 registers; a kernel is allocated for its hungriest
// case, so the 1024-thread column kernel (128 registers per thread) has them in an instantiation of its own (FlexFactors::bigp).
template <int DIR, bool FIRST, bool COLS, bool BIGP>
__device__ __forceinline__ void flex_pass_other(const c32* A, c32* B, const c32* tw, int R, int n, PassArgs pa, int tsh, int first, int step)
{
    if constexpr (BIGP) {
        if (R == 11) return flex_pass<11, 1, DIR, FIRST, COLS>(A, B, tw, pa, tsh, first, step);
        if (R == 13) return flex_pass<13, 1, DIR, FIRST, COLS>(A, B, tw, pa, tsh, first, step);
    }
    flex_pass_prime<DIR, COLS>(A, B, tw, R, n, pa, tsh, first, step);
}

template <int SYNC, int DIR, bool COLS, bool BIGP>
__device__ __forceinline__ c32* flex_fft(c32* A, c32* B, const c32* tw, const FlexFactors& pl, int tsh, int first, int step)
{
    for (int p = 0; p < pl.nf; ++p) {
        const int R = pl.f[p];
        if (p == 0) {
            switch (R) {
#define P3D_FLEX_CASE(R, R1, R2) case R: flex_pass<R1, R2, DIR, true, COLS>(A, B, tw, pass_args(pl, 0), tsh, first, step); break;
                P3D_FLEX_RADICES(P3D_FLEX_CASE)
#undef P3D_FLEX_CASE
                default: flex_pass_other<DIR, true, COLS, BIGP>(A, B, tw, R, pl.m, pass_args(pl, 0), tsh, first, step);
            }
        } else {
            switch (R) {
#define P3D_FLEX_CASE(R, R1, R2) case R: flex_pass<R1, R2, DIR, false, COLS>(A, B, tw, pass_args(pl, p), tsh, first, step); break;
                P3D_FLEX_RADICES(P3D_FLEX_CASE)
#undef P3D_FLEX_CASE
                default: flex_pass_other<DIR, false, COLS, BIGP>(A, B, tw, R, pl.m, pass_args(pl, p), tsh, first, step);
            }
        }
        flex_sync<SYNC>();
        c32* t = A; A = B; B = t;
    }
    return A;
}

// DFT of a line (or a tile of lines) of pl.n points.  Unnormalised; the result is returned in one of the two buffers.  (Lengths with a
// prime factor above 13 that run in the chirp-z form never come here: flex_row / flex_col hand them to p3d_chirp.hip.)
template <int SYNC, int DIR, bool COLS, bool BIGP>
__device__ __forceinline__ c32* flex_transform(c32* A, c32* B, const c32* tw, const c32*, const FlexFactors& pl, int tsh, int first, int step)
{
    return flex_fft<SYNC, DIR, COLS, BIGP>(A, B, tw, pl, tsh, first, step);
}

// ---- column pass ------------------------------------------------------------------------------------------------------------------
constexpr int FLEX_COL_THREADS = 1024;   // 16 waves per CU although a tile of long columns allows one workgroup per CU only (512: 28.6 vs 31 Gpt/s)

// Persistent: a workgroup owns a contiguous run of (slice, tile) pairs.  Long columns leave room for one workgroup per CU only
// (two buffers of a tile: 128 KiB at 1000 x 8), so nothing else on the CU would cover a tile's load: the NEXT tile is requested
// into registers (n T / 1024 <= FLEX_COL_PF samples per thread) before the current one is transformed.
constexpr int FLEX_COL_PF = 10;   // 2 T L <= FLEX_LDS_MAX / 8  =>  n T / 1024 <= 9.4

template <bool PERSIST, bool BIGP>   // PERSIST false: one tile per workgroup (per = 1), loaded straight into LDS; BIGP: see flex_fft
__global__ __launch_bounds__(FLEX_COL_THREADS) void flex_col_kernel(const ColArgs a, const FlexFactors* __restrict__ plp, int mode, int tshift, int ntiles, int per)
{
    const FlexFactors& pl = *plp;   // behind the line's twiddle table (flex_build_table): uniform address, scalar loads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ float r[(FLEX_COL_THREADS / 64) * 5];
    const int n = pl.n, L = pl.m, T = 1 << tshift, tid_ = threadIdx.x;
    c32* tw = reinterpret_cast<c32*>(smem_raw);
    c32* A = tw + L;
    c32* B = A + (size_t)L * T;
    const bool iter = mode == COL_ITER || mode == COL_ITER_SOFT || mode == COL_ITER_GARROTE;
    const int total = ntiles * a.nslices, nel = n << tshift;
    const int begin = blockIdx.x * per, end = min(begin + per, total);
    if (begin >= end) return;
    for (int i = tid_; i < L; i += FLEX_COL_THREADS) tw[i] = a.tw[i];

    auto goff = [&](int std_layout, int i, int col) -> size_t {
        return std_layout ? (size_t)i * a.n2 + col : ((size_t)(col >> 3) * n + i) * 8 + (col & 7);
    };
    // request tile `idx` into registers (nothing for a slice that is finished: its tile is skipped below).  A whole column block
    // of the work buffer is one contiguous run of n * 8 samples.
    c32 pre[FLEX_COL_PF];
    auto request = [&](int idx) {
        int t = tid_;
        asm volatile("" : "+v"(t));   // (addresses are computed here, per tile: hoisted out of the tile loop they spilled)
        const int slice = idx / ntiles, col0 = (idx - slice * ntiles) << tshift;
        const bool live = idx < end && !(a.done && a.done[slice] != 0);
        const c32* const inb = a.in + (size_t)slice * (a.in_std ? (size_t)n * a.n2 : wk_slice_stride(n, a.n2));
        if (!a.in_std && T == 8 && col0 + 8 <= a.n2) {
            const c32* const src = inb + (size_t)(col0 >> 3) * n * 8 + t;
#pragma unroll
            for (int k = 0; k < FLEX_COL_PF; ++k) {
                pre[k] = c32{0.f, 0.f};
                if (live && t + k * FLEX_COL_THREADS < nel) pre[k] = src[k * FLEX_COL_THREADS];
            }
        } else {
#pragma unroll
            for (int k = 0; k < FLEX_COL_PF; ++k) {
                const int e = t + k * FLEX_COL_THREADS, c = e & (T - 1), i = e >> tshift, col = col0 + c;
                pre[k] = c32{0.f, 0.f};
                if (live && e < nel && col < a.n2) pre[k] = inb[goff(a.in_std, i, col)];
            }
        }
    };
    if constexpr (PERSIST) request(begin);

    // (one trip when not persistent: the compiler then sees straight-line code)
    for (int idx = begin; idx < (PERSIST ? end : begin + 1); ++idx) {
        // what the passes derive from the thread index belongs to one radix case each: it is computed there, per tile, and not
        // hoisted out of the tile loop for all of them
        int tid = tid_;
        asm volatile("" : "+v"(tid));
        const int slice = idx / ntiles, tile = idx - slice * ntiles, col0 = tile << tshift;
        const bool skip = a.done && a.done[slice] != 0;
        if constexpr (PERSIST) {
            if (!skip) {
#pragma unroll
                for (int k = 0; k < FLEX_COL_PF; ++k) {
                    const int e = tid + k * FLEX_COL_THREADS;
                    if (e < nel) A[e] = pre[k];
                }
            }
        } else {
            if (skip) continue;
            const c32* const inb = a.in + (size_t)slice * (a.in_std ? (size_t)n * a.n2 : wk_slice_stride(n, a.n2));
            for (int e = tid; e < nel; e += FLEX_COL_THREADS) {
                const int c = e & (T - 1), i = e >> tshift, col = col0 + c;
                A[e] = col < a.n2 ? inb[goff(a.in_std, i, col)] : c32{0.f, 0.f};
            }
        }
        __syncthreads();   // the tile (and, the first time, the twiddle table) is in LDS; nobody reads the previous tile any more
        if constexpr (PERSIST) request(idx + 1);
        if (skip) continue;
        c32* const outb = a.out + (size_t)slice * (a.out_std ? (size_t)n * a.n2 : wk_slice_stride(n, a.n2));

        c32* X = A;
        c32* Y = B;
        if (mode != COL_INV) {
            // (a length with a factor 11 or 13 is never run in the chirp-z form: the BIGP instantiation leaves that code out)
            if constexpr (BIGP) X = flex_fft<0, FWD, true, true>(A, B, tw, pl, tshift, tid, FLEX_COL_THREADS);
            else X = flex_transform<0, FWD, true, false>(A, B, tw, a.tw, pl, tshift, tid, FLEX_COL_THREADS);
            Y = X == A ? B : A;
        }
        bool emptied = false;
        if (iter || (mode == COL_FWD && a.tau != nullptr)) {
            const c32 tau = a.tau[(size_t)slice * a.niter + a.iter];
            const int op = mode == COL_ITER_SOFT ? 1 : (mode == COL_ITER_GARROTE ? 2 : a.op);   // callers pass COL_ITER + a.op
            bool any = false;
            const Shrink shr(tau, op);
            for (int e = tid; e < nel; e += FLEX_COL_THREADS) {
                const c32 v = shr(X[e]);
                X[e] = v;
                any = any || v.x != 0.0f || v.y != 0.0f;
            }
            const int kept = __syncthreads_or(any ? 1 : 0);
            if (iter && a.nzflag != nullptr) {   // a tile the threshold emptied stays zeros: say so instead of transforming and storing it
                if (tid == 0) a.nzflag[(size_t)slice * ntiles + tile] = kept ? 1 : 0;
                if (!kept) {
                    // the row pass skips whole 8-column blocks: an empty tile NARROWER than a block must still leave zeros behind for
                    // the case that a sibling tile of its block kept something (see col_kernel)
                    if (T < 8)
                        for (int e = tid; e < nel; e += FLEX_COL_THREADS) {
                            const int c = e & (T - 1), i = e >> tshift, col = col0 + c;
                            if (col < a.n2) outb[goff(a.out_std, i, col)] = c32{0.f, 0.f};
                        }
                    emptied = true;
                }
            }
        }
        if (emptied) continue;
        if (mode == COL_STATS) {
            // lexicographic complex max, max|X|, min|X|, sum|X|^2 of this tile (POCS.py:261-262, 288, 299)
            float lr = -INFINITY, li = -INFINITY, mx = 0.f, mn = INFINITY, sq = 0.f;
            for (int e = tid; e < nel; e += FLEX_COL_THREADS) {
                if (col0 + (e & (T - 1)) >= a.n2) continue;
                const c32 v = X[e];
                const float p = v.x * v.x + v.y * v.y;
                if (lex_greater(v.x, v.y, lr, li)) { lr = v.x; li = v.y; }
                mx = fmaxf(mx, p);
                mn = fminf(mn, p);
                sq += p;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float orr = __shfl_down(lr, o, 64), oi = __shfl_down(li, o, 64);
                const float omx = __shfl_down(mx, o, 64), omn = __shfl_down(mn, o, 64), osq = __shfl_down(sq, o, 64);
                if (lex_greater(orr, oi, lr, li)) { lr = orr; li = oi; }
                mx = fmaxf(mx, omx);
                mn = fminf(mn, omn);
                sq += osq;
            }
            if ((tid & 63) == 0) {
                float* o = r + (tid >> 6) * 5;
                o[0] = lr; o[1] = li; o[2] = mx; o[3] = mn; o[4] = sq;
            }
            __syncthreads();
            if (tid == 0) {
                for (int t = 1; t < FLEX_COL_THREADS / 64; ++t) {
                    const float* o = r + t * 5;
                    if (lex_greater(o[0], o[1], lr, li)) { lr = o[0]; li = o[1]; }
                    mx = fmaxf(mx, o[2]);
                    mn = fminf(mn, o[3]);
                    sq += o[4];
                }
                float* p = a.partials + ((size_t)slice * ntiles + tile) * STATS_PARTIAL;
                p[0] = lr; p[1] = li; p[2] = sqrtf(mx); p[3] = sqrtf(mn); p[4] = sq;
            }
            continue;   // (the barrier at the top of the next tile separates this tile's use of r[] from the next one's)
        }
        if (iter || mode == COL_INV) {
            if constexpr (BIGP) X = flex_fft<0, INV, true, true>(X, Y, tw, pl, tshift, tid, FLEX_COL_THREADS);
            else X = flex_transform<0, INV, true, false>(X, Y, tw, a.tw, pl, tshift, tid, FLEX_COL_THREADS);
        }
        for (int e = tid; e < nel; e += FLEX_COL_THREADS) {
            const int c = e & (T - 1), i = e >> tshift, col = col0 + c;
            if (col < a.n2) outb[goff(a.out_std, i, col)] = X[e];
        }
    }
}

// ---- row pass ------------------------------------------------------------------------------------------------------------------------
// TPR threads per row (64: one wavefront, wave-level synchronisation only; 128: two wavefronts and workgroup barriers -- half the
// rows per workgroup, half the LDS, twice the resident waves), LB rows per workgroup; modes ROW_FIRST / ROW_MID / ROW_LAST as in
// row_kernel (p3d_kernels.hpp)
template <int TPR, bool INPL>
__global__ __launch_bounds__(256) void flex_row_kernel(const RowArgs a, const FlexFactors* __restrict__ plp, int mode, int LB)
{
    const FlexFactors& pl = *plp;
    constexpr bool inpl = INPL;
    constexpr int SYNC = TPR == 64 ? 1 : 0;
    __shared__ double rsum[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = pl.n, L = pl.m, tid = threadIdx.x, lane = tid % TPR, line = tid / TPR;
    c32* tw = reinterpret_cast<c32*>(smem_raw);
    c32* A = tw + L + (size_t)line * (inpl ? 1 : 2) * L;   // inpl: in-place passes, one buffer per row (flex_inplace_ok)
    c32* B = A + L;
    const int slice = blockIdx.y, row = blockIdx.x * LB + line;
    const bool valid = row < a.n1;
    const int vrow = valid ? row : 0;
    for (int i = tid; i < L; i += blockDim.x) tw[i] = a.tw[i];
    __syncthreads();   // TPR = 64: the only workgroup-wide barrier, from here on every wave is on its own

    const int dn = a.done ? a.done[slice] : 0;
    const size_t sbase = ((size_t)slice * a.n1 + vrow) * n;   // row-major cubes (x, out)
    if (mode == ROW_LAST && a.only_done) {
        if (dn <= a.only_done_lo || dn > a.only_done) return;
    } else if (mode == ROW_LAST) {
        if (dn > 0) return;   // converged earlier: `out` already holds that iterate
        if (dn < 0) {         // all-zero slice is handed back untouched (POCS.py:515-521)
            if (valid)
                for (int i = lane; i < n; i += TPR) {
                    if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[sbase + i] = c32{0.f, 0.f};
                    else reinterpret_cast<float*>(a.out)[sbase + i] = 0.f;
                }
            return;
        }
    } else if (dn != 0) {
        return;
    }
    c32* const wrow = a.work + (size_t)slice * wk_slice_stride(a.n1, n) + (size_t)vrow * 8;   // + (i>>3)*n1*8 + (i&7)
    const size_t wblk = (size_t)a.n1 * 8;
    auto obs_at = [&](int i) -> c32 {
        if (a.dtype == 0) return reinterpret_cast<const c32*>(a.x)[sbase + i];
        return c32{reinterpret_cast<const float*>(a.x)[sbase + i], 0.f};
    };
    const float* const mrow = a.mask ? a.mask + (size_t)vrow * n : nullptr;

    float acc = 0.f;
    c32* X = A;
    if (mode == ROW_FIRST) {
        for (int i = lane; i < n; i += TPR) {
            const c32 x = valid ? obs_at(i) : c32{0.f, 0.f};
            acc += abs_c32(x);
            if (a.adaptive) {   // x_old = x at the first iteration (POCS.py:549, 574-575)
                const float m = mrow ? mrow[i] : 0.f;
                const float w = 1.0f - a.alpha * m;
                const c32 blend = x * a.alpha + x * w;
                A[i] = blend + (x - x * m) * (1.0f - a.alpha);
            } else {
                A[i] = x;
            }
        }
        flex_sync<SYNC>();
    } else {
        // column blocks the column pass found empty were not stored (RowArgs::nzflag): they read as zeros
        const uint8_t* const nzf = (a.nzflag && !a.only_done) ? a.nzflag + (size_t)slice * a.nz_tiles : nullptr;
        const int tsh = 31 - __builtin_clz((unsigned)a.nz_col_t);   // column tiles are 1, 2, 4 or 8 columns wide (pick_col_tile)
#pragma unroll 4
        for (int i = lane; i < n; i += TPR) {
            bool kept = valid;
            if (nzf) kept = kept && nzf[i >> tsh] != 0;
            A[i] = kept ? wrow[(size_t)(i >> 3) * wblk + (i & 7)] : c32{0.f, 0.f};
        }
        // The observed samples and weights of the row are requested HERE, in front of the inverse transform, and used after it
        // (rows of up to PFN * TPR samples; longer rows load the rest where they use it): the pass is latency bound -- 70 % of its
        // wave cycles were waits -- and these two loads sat, fully exposed, between the transforms.
        // Only in the two-buffer kernel with one wavefront per row (four waves per SIMD whatever it does: 1024 x 600 69 -> 81 Gpt/s).
        // The in-place kernels run six waves per SIMD on 77 registers and lose more to the 24 registers of the prefetch than the
        // exposed loads cost them (1000 x 1000: 1.02 -> 1.14 ms; 960 x 768 on four wavefronts per row: 57.8 -> 50.5 Gpt/s).
        constexpr int PFN = (!INPL && TPR == 64) ? 8 : 0;
        c32 xo_pre[PFN ? PFN : 1];
        float m_pre[PFN ? PFN : 1];
        const bool need_obs = !a.plain && valid;
#pragma unroll
        for (int q = 0; q < PFN; ++q) {
            const int i = lane + q * TPR;
            xo_pre[q] = c32{0.f, 0.f};
            m_pre[q] = 0.f;
            if (i < n && need_obs) xo_pre[q] = obs_at(i);
            if (i < n && mrow && !a.plain) m_pre[q] = mrow[i];
        }
        flex_sync<SYNC>();
        if constexpr (INPL) flex_transform_inplace<SYNC, INV>(A, tw, a.tw, pl, lane, TPR);
        else X = flex_transform<SYNC, INV, false, true>(A, B, tw, a.tw, pl, 0, lane, TPR);
        auto update = [&](const int i, const c32 xo, const float mk) {
            c32 xn = X[i] * a.scale;
            float m = 0.f;
            if (mode == ROW_LAST && a.only_done) {
                // the converged iterate up to one row-transform round trip; an observed trace with alpha = 1 IS the observation
                if (a.alpha == 1.0f && mrow && mk == 1.0f) xn = xo;
            } else if (!a.plain) {
                m = mk;
                const float w = 1.0f - a.alpha * m;        // POCS.py:616
                xn = axpby(xn, w, xo, a.alpha);            // POCS.py:619
            }
            acc += abs_c32(xn);   // hardware square root: the sums only feed the convergence test
            if ((mode == ROW_LAST || a.write_out) && valid) {
                if (a.dtype == 0) reinterpret_cast<c32*>(a.out)[sbase + i] = xn;
                else reinterpret_cast<float*>(a.out)[sbase + i] = xn.x;   // np.real(), POCS.py:656
            }
            if (mode == ROW_MID) {
                if (a.adaptive) {   // x_input of the next iteration (POCS.py:574-575)
                    const float w = 1.0f - a.alpha * m;
                    const c32 blend = xo * a.alpha + xn * w;
                    X[i] = blend + (xo - xn * m) * (1.0f - a.alpha);
                } else {
                    X[i] = xn;
                }
            }
        };
#pragma unroll
        for (int q = 0; q < PFN; ++q)
            if (lane + q * TPR < n) update(lane + q * TPR, xo_pre[q], m_pre[q]);
        for (int i = lane + PFN * TPR; i < n; i += TPR)
            update(i, need_obs ? obs_at(i) : c32{0.f, 0.f}, (mrow && !a.plain) ? mrow[i] : 0.f);
        flex_sync<SYNC>();
    }
    if (a.sums != nullptr) {
        double ws = valid ? (double)acc : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ws += __shfl_down(ws, o, 64);
        if constexpr (TPR == 64) {
            if (lane == 0 && valid) a.sums[(size_t)slice * a.n1 + row] = ws;
        } else {   // two or four wavefronts per row
            constexpr int W = TPR / 64;
            if ((tid & 63) == 0) rsum[tid >> 6] = ws;
            __syncthreads();
            if (lane == 0 && valid) {
                double t = rsum[W * line];
#pragma unroll
                for (int w = 1; w < W; ++w) t += rsum[W * line + w];
                a.sums[(size_t)slice * a.n1 + row] = t;
            }
        }
    }
    if (mode != ROW_LAST) {
        c32* Y = X == A ? B : A;
        if constexpr (INPL) flex_transform_inplace<SYNC, FWD>(A, tw, a.tw, pl, lane, TPR);
        else X = flex_transform<SYNC, FWD, false, true>(X, Y, tw, a.tw, pl, 0, lane, TPR);
        if (valid)
            for (int i = lane; i < n; i += TPR) wrow[(size_t)(i >> 3) * wblk + (i & 7)] = X[i];
    }
}

// Real (float32) cubes with the hard operator (see row_real_kernel in p3d_kernels.hpp): rows 2p and 2p + 1 share one complex
// transform, z = r_a + i r_b, and the work buffer holds columns 0 ... n/2 of the row spectra.  The line lives in LDS here, so the
// partner Z[n - k] of the split is simply another element of the same buffer.  Any real mask (float weights), no compaction.
template <int TPR, bool INPL>
__global__ __launch_bounds__(256) void flex_row_real_kernel(const RowArgs a, const FlexFactors* __restrict__ plp, int mode, int LB)
{
    const FlexFactors& pl = *plp;
    constexpr int SYNC = TPR == 64 ? 1 : 0;
    __shared__ double rsum[8];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = pl.n, L = pl.m, H = n / 2, tid = threadIdx.x, lane = tid % TPR, line = tid / TPR;
    c32* tw = reinterpret_cast<c32*>(smem_raw);
    c32* A = tw + L + (size_t)line * (INPL ? 1 : 2) * L;
    c32* B = A + L;
    const int slice = blockIdx.y, pair = blockIdx.x * LB + line;
    const bool valid = 2 * pair + 1 < a.n1;
    const int ra = valid ? 2 * pair : 0;
    for (int i = tid; i < L; i += blockDim.x) tw[i] = a.tw[i];
    __syncthreads();

    const int dn = a.done ? a.done[slice] : 0;
    const size_t sbase = ((size_t)slice * a.n1 + ra) * n;   // row a of the row-major cubes; row b follows
    const float* const xa = reinterpret_cast<const float*>(a.x) + sbase;
    float* const oa = reinterpret_cast<float*>(a.out) + sbase;
    if (mode == ROW_LAST && a.only_done) {
        if (dn <= a.only_done_lo || dn > a.only_done) return;
    } else if (mode == ROW_LAST) {
        if (dn > 0) return;
        if (dn < 0) {
            if (valid)
                for (int i = lane; i < n; i += TPR) { oa[i] = 0.f; oa[n + i] = 0.f; }
            return;
        }
    } else if (dn != 0) {
        return;
    }
    c32* const wrow = a.work + (size_t)slice * wk_slice_stride(a.n1, H + 1) + (size_t)ra * 8;   // row a; row b is 8 elements on
    const size_t wblk = (size_t)a.n1 * 8;
    const float* const ma = a.mask ? a.mask + (size_t)ra * n : nullptr;

    float sa = 0.f, sb = 0.f;
    c32* X = A;
    if (mode == ROW_FIRST) {
        for (int i = lane; i < n; i += TPR) {
            const float va = valid ? xa[i] : 0.f, vb = valid ? xa[n + i] : 0.f;
            sa += fabsf(va);
            sb += fabsf(vb);
            A[i] = c32{va, vb};
        }
        flex_sync<SYNC>();
    } else {
        const uint8_t* const nzf = (a.nzflag && !a.only_done) ? a.nzflag + (size_t)slice * a.nz_tiles : nullptr;
        const int tsh = 31 - __builtin_clz((unsigned)a.nz_col_t);
#pragma unroll 4
        for (int i = lane; i < n; i += TPR) {
            const int k = i <= H ? i : n - i;
            bool kept = valid;
            if (nzf) kept = kept && nzf[k >> tsh] != 0;
            c32 r0{0.f, 0.f}, r1{0.f, 0.f};
            if (kept) {
                const c32* w = wrow + (size_t)(k >> 3) * wblk + (k & 7);
                r0 = w[0];
                r1 = w[8];
            }
            if (i > H) { r0.y = -r0.y; r1.y = -r1.y; }
            if (k == 0 || 2 * k == n) { r0.y = 0.f; r1.y = 0.f; }   // self-mirrored columns of a real row are real
            A[i] = c32{r0.x - r1.y, r0.y + r1.x};
        }
        flex_sync<SYNC>();
        if constexpr (INPL) flex_transform_inplace<SYNC, INV>(A, tw, a.tw, pl, lane, TPR);
        else X = flex_transform<SYNC, INV, false, true>(A, B, tw, a.tw, pl, 0, lane, TPR);
        const bool handback = mode == ROW_LAST && a.only_done != 0;
        for (int i = lane; i < n; i += TPR) {
            const c32 z = X[i] * a.scale;
            float va = z.x, vb = z.y;
            const float xoa = valid ? xa[i] : 0.f, xob = valid ? xa[n + i] : 0.f;
            const float mka = ma ? ma[i] : 0.f, mkb = ma ? ma[n + i] : 0.f;
            if (handback) {
                if (a.alpha == 1.0f && mka == 1.0f) va = xoa;
                if (a.alpha == 1.0f && mkb == 1.0f) vb = xob;
            } else {
                va = __builtin_fmaf(va, 1.0f - a.alpha * mka, xoa * a.alpha);   // POCS.py:616-619
                vb = __builtin_fmaf(vb, 1.0f - a.alpha * mkb, xob * a.alpha);
            }
            sa += fabsf(va);
            sb += fabsf(vb);
            if (mode == ROW_LAST && valid) { oa[i] = va; oa[n + i] = vb; }
            if (mode == ROW_MID) X[i] = c32{va, vb};
        }
        flex_sync<SYNC>();
    }
    if (a.sums != nullptr) {
        double da = valid ? (double)sa : 0.0, db = valid ? (double)sb : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { da += __shfl_down(da, o, 64); db += __shfl_down(db, o, 64); }
        if constexpr (TPR == 64) {
            if (lane == 0 && valid) { a.sums[(size_t)slice * a.n1 + ra] = da; a.sums[(size_t)slice * a.n1 + ra + 1] = db; }
        } else {   // two wavefronts per row pair
            if ((tid & 63) == 0) { rsum[2 * (tid >> 6)] = da; rsum[2 * (tid >> 6) + 1] = db; }
            __syncthreads();
            if (lane == 0 && valid) {
                a.sums[(size_t)slice * a.n1 + ra] = rsum[4 * line] + rsum[4 * line + 2];
                a.sums[(size_t)slice * a.n1 + ra + 1] = rsum[4 * line + 1] + rsum[4 * line + 3];
            }
        }
    }
    if (mode != ROW_LAST) {
        c32* Y = X == A ? B : A;
        if constexpr (INPL) flex_transform_inplace<SYNC, FWD>(A, tw, a.tw, pl, lane, TPR);
        else X = flex_transform<SYNC, FWD, false, true>(X, Y, tw, a.tw, pl, 0, lane, TPR);
        if (valid)
            for (int k = lane; k <= H; k += TPR) {
                const c32 z = X[k], pz = X[k == 0 ? 0 : n - k];
                c32* w = wrow + (size_t)(k >> 3) * wblk + (k & 7);
                w[0] = c32{0.5f * (z.x + pz.x), 0.5f * (z.y - pz.y)};      // R_a = (Z[k] + conj Z[n-k]) / 2
                w[8] = c32{0.5f * (z.y + pz.y), -0.5f * (z.x - pz.x)};     // R_b = (Z[k] - conj Z[n-k]) / 2i
            }
    }
}

// The pass list of a length sits behind its device table (flex_build_table).  It is read through a pointer: as a by-value kernel
// argument with run-time indexed arrays it was copied to scratch memory by some instantiations (260 bytes, vector loads per pass).
size_t flex_table_len(const FlexFactors& pl) { return pl.blue ? (size_t)pl.m + pl.n : (size_t)pl.n; }
const FlexFactors* device_factors(const c32* table, const FlexFactors& pl) { return reinterpret_cast<const FlexFactors*>(table + flex_table_len(pl)); }

// the tables of a chirp-z length inside its device table (flex_build_table): [chirp | spectrum of conj chirp | FlexFactors | tables of the
// register-resident engine for M]
ChirpTabs chirp_tabs(const c32* table, const FlexFactors& pl)
{
    const c32* const tuned = table + flex_table_len(pl) + (sizeof(FlexFactors) + sizeof(c32) - 1) / sizeof(c32);
    return ChirpTabs{tuned, tuned + chirp_rowtab_slots(pl.m), table, table + pl.n, pl.n, pl.m};
}

// the tables of the register engine for a 7-smooth length inside its device table: [exp(-2 pi i k / n) | FlexFactors | per-pass twiddle rows]
const c32* mix_tab(const c32* table, const FlexFactors& pl) { return table + flex_table_len(pl) + (sizeof(FlexFactors) + sizeof(c32) - 1) / sizeof(c32); }

// row-pair passes for float32 cubes (mode = ROW_FIRST / ROW_MID / ROW_LAST)
hipError_t flex_row_real(int mode, const RowArgs& a, int, hipStream_t st)
{
    if (mode != ROW_FIRST && mode != ROW_MID && mode != ROW_LAST) return hipErrorNotSupported;
    if (a.dtype != 1 || a.n1 % 2 != 0 || a.adaptive || a.write_out) return hipErrorNotSupported;
    const int n = a.len, LB = pick_row_lines(n);
    if (LB == 0) return hipErrorNotSupported;
    const FlexFactors pl = flex_factors(n);
    if (pl.blue) return chirp_row_real(mode, a, chirp_tabs(a.tw, pl), st);
    // 7-smooth rows: the complex passes of the register engine (p3d_mix.hpp) beat the row pairs of the LDS-image kernel below although they
    // transform twice the columns; "not supported" sends the caller there (p3d_pocs_run_dev falls back to the complex path)
    if (const mix::Entry* e = mix::find(n); e && e->row) return hipErrorNotSupported;
    int widest = 2;
    for (int p = 0; p < pl.nf; ++p) widest = pl.f[p] > widest ? pl.f[p] : widest;
    const bool two = LB >= 2 && pl.m / widest >= 48;
    const bool no_inplace = (a.host_sw & P3D_SW_FLEX_NO_INPLACE) != 0;
    bool two_ = two;
    int inpl = 0;
    if (!no_inplace) {
        if (flex_inplace_ok(pl, two ? 128 : 64)) inpl = 1;
        else if (!two && flex_inplace_ok(pl, 128)) { inpl = 1; two_ = true; }
    }
    int lb = two_ ? LB / 2 : LB;
    if (lb < 1) lb = 1;
    if (inpl) {
        lb = two_ ? 2 : 4;
        while (lb > 1 && row_lds_inplace(n, lb) > FLEX_LDS_MAX) lb /= 2;
    }
    const size_t lds = inpl ? row_lds_inplace(n, lb) : row_lds(n, lb);
    const int pairs = a.n1 / 2;
    const dim3 grid((pairs + lb - 1) / lb, a.nslices);
#define P3D_FLEX_ROW(TPR, IP)                                                                                                  \
    do {                                                                                                                       \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flex_row_real_kernel<TPR, IP>),                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLEX_LDS_MAX);                     \
        if (e != hipSuccess) return e;                                                                                         \
        flex_row_real_kernel<TPR, IP><<<grid, TPR * lb, lds, st>>>(a, device_factors(a.tw, pl), mode, lb);                                           \
    } while (0)
    if (two_) { if (inpl) P3D_FLEX_ROW(128, true); else P3D_FLEX_ROW(128, false); }
    else { if (inpl) P3D_FLEX_ROW(64, true); else P3D_FLEX_ROW(64, false); }
#undef P3D_FLEX_ROW
    return hipGetLastError();
}

hipError_t flex_row(int mode, const RowArgs& a, hipStream_t st)
{
    if (mode != ROW_FIRST && mode != ROW_MID && mode != ROW_LAST) return hipErrorNotSupported;   // no shearlet passes here
    if (a.bits != nullptr) return hipErrorInvalidValue;   // packed masks belong to the tuned row pass
    const int n = a.len, LB = pick_row_lines(n);
    if (LB == 0) return hipErrorNotSupported;
    const FlexFactors pl = flex_factors(n);
    if (pl.blue) return chirp_row(mode, a, chirp_tabs(a.tw, pl), st);
    if (const mix::Entry* e = mix::find(n); e && e->row) return e->row(mode, a, mix_tab(a.tw, pl), st);
    // two wavefronts per row where the narrowest pass still has ~48 butterflies for them (measured: 500, 768, 1000 gain 20-25 %,
    // 600 = 15*10*4 loses 8 %) and rows can be paired
    int widest = 2;
    for (int p = 0; p < pl.nf; ++p) widest = pl.f[p] > widest ? pl.f[p] : widest;
    const bool two = LB >= 2 && pl.m / widest >= 48;
    const bool no_inplace = (a.host_sw & P3D_SW_FLEX_NO_INPLACE) != 0;
    int tpr = two ? 128 : 64;
    int inpl = 0;
    if (!no_inplace) {
        // in-place passes want one butterfly per thread: more wavefronts per row where the passes are wide
        // (four only where the passes keep 40 % of the 256 lanes busy: 600 = 15 x 10 x 4 would use a third of them and lost 9 %)
        int butterflies = 0;
        for (int p = 0; p < pl.nf; ++p) butterflies += pl.m / pl.f[p];
        for (int t = tpr; t <= 256 && !inpl; t *= 2)
            if (flex_inplace_ok(pl, t) && (t < 256 || 5 * butterflies >= 2 * pl.nf * t)) { inpl = 1; tpr = t; }
    }
    int lb = tpr == 128 && !inpl ? LB / 2 : LB;
    if (lb < 1) lb = 1;
    if (inpl) {   // 256 threads per workgroup; the single buffers let four of them share a CU
        lb = 256 / tpr;
        while (lb > 1 && row_lds_inplace(n, lb) > FLEX_LDS_MAX) lb /= 2;
        if (row_lds_inplace(n, lb) > FLEX_LDS_MAX) return hipErrorNotSupported;
    }
    const size_t lds = inpl ? row_lds_inplace(n, lb) : row_lds(n, lb);
    const dim3 grid((a.n1 + lb - 1) / lb, a.nslices);
#define P3D_FLEX_ROW(TPR, IP)                                                                                                  \
    do {                                                                                                                       \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flex_row_kernel<TPR, IP>),                            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLEX_LDS_MAX);                     \
        if (e != hipSuccess) return e;                                                                                         \
        flex_row_kernel<TPR, IP><<<grid, TPR * lb, lds, st>>>(a, device_factors(a.tw, pl), mode, lb);                                                \
    } while (0)
    if (tpr == 256) P3D_FLEX_ROW(256, true);
    else if (tpr == 128) { if (inpl) P3D_FLEX_ROW(128, true); else P3D_FLEX_ROW(128, false); }
    else { if (inpl) P3D_FLEX_ROW(64, true); else P3D_FLEX_ROW(64, false); }
#undef P3D_FLEX_ROW
    return hipGetLastError();
}

hipError_t flex_col(int mode, const ColArgs& a, hipStream_t st)
{
    if (mode == COL_SHRINK) return hipErrorNotSupported;
    const int n = a.len, T = pick_col_tile(n);
    if (T == 0) return hipErrorNotSupported;
    int tshift = 0;
    while ((1 << tshift) < T) ++tshift;
    const FlexFactors pl = flex_factors(n);
    if (pl.blue) return chirp_col(mode, a, chirp_tabs(a.tw, pl), st);
    if (const mix::Entry* e = mix::find(n)) return e->col(mode, a, mix_tab(a.tw, pl), st);
    const size_t lds = col_lds(n, T);
    if ((size_t)n * T > (size_t)FLEX_COL_PF * FLEX_COL_THREADS) return hipErrorNotSupported;   // (never: col_lds <= FLEX_LDS_MAX)
    int cus = a.cus;   // the plan's device (a.cus = 0: a caller without a plan -- ask the current device)
    if (cus <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    // Long columns (one workgroup per CU): contiguous runs of tiles, the next tile prefetched.  Short ones share a CU and cover
    // each other's loads: one tile per workgroup there (measured: 300-point columns 0.23 vs 0.27 ms).
    const int ntiles = (a.n2 + T - 1) / T, total = ntiles * a.nslices;
    const bool no_persist = (a.host_sw & P3D_SW_FLEX_NO_PERSIST) != 0;
    const bool persist = !no_persist && lds + 512 > 80 * 1024;
    // eight workgroups' worth of runs per CU: the hardware hands the next run to the CU that is free (kept tiles cost twice an
    // emptied one and cluster around the low wavenumbers: one run per CU left some CUs with 45 % more work -- 0.90 vs 0.65 ms)
    const int over = a.flex_over > 0 ? a.flex_over : 8;
    const int want = cus * (over > 0 ? over : 1);
    const int grid = !persist ? total : (total < want ? total : want), per = (total + grid - 1) / grid;
    bool bigp = false;
    for (int i = 0; i < pl.nf; ++i) bigp = bigp || pl.f[i] == 11 || pl.f[i] == 13;
    bigp = bigp && !pl.blue;
#define P3D_FLEX_COL(PS, BP)                                                                                                     \
    do {                                                                                                                         \
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flex_col_kernel<PS, BP>),                         \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLEX_LDS_MAX);                 \
        if (e != hipSuccess) return e;                                                                                           \
        flex_col_kernel<PS, BP><<<PS ? (total + per - 1) / per : total, FLEX_COL_THREADS, lds, st>>>(a, device_factors(a.tw, pl), mode, tshift, ntiles, PS ? per : 1); \
    } while (0)
    if (persist) { if (bigp) P3D_FLEX_COL(true, true); else P3D_FLEX_COL(true, false); }
    else { if (bigp) P3D_FLEX_COL(false, true); else P3D_FLEX_COL(false, false); }
#undef P3D_FLEX_COL
    return hipGetLastError();
}

hipError_t flex_no_pipe(const RowArgs&, int, hipStream_t) { return hipErrorNotSupported; }

}  // namespace

// device table of a line length: exp(-2 pi i k / n), or for the chirp-z form [chirp c_k = exp(-i pi k^2 / n) | FFT_M(conj c, wrapped) | ... the
// twiddle tables of the register-resident engine for M points], computed in double precision and rounded once
void flex_build_table(int n, std::vector<c32>& out)
{
    const FlexFactors pl = flex_factors(n);
    auto append_factors = [&] {   // read by the kernels through device_factors()
        const size_t at = out.size();
        out.resize(at + (sizeof(FlexFactors) + sizeof(c32) - 1) / sizeof(c32), c32{0.f, 0.f});
        memcpy(out.data() + at, &pl, sizeof(FlexFactors));
    };
    if (!pl.blue) {
        out.resize(n);
        gen_build_twiddles(n, out.data());
        append_factors();
        if (const mix::Entry* e = mix::find(n)) {   // the per-pass twiddle rows of the register engine (p3d_mix.hpp), read through mix_tab()
            const size_t at = out.size();
            out.resize(at + (size_t)e->tw_slots + 1, c32{0.f, 0.f});
            e->build_tw(out.data() + at);
        }
        return;
    }
    const int M = pl.m;
    const double pi = 3.14159265358979323846;
    out.resize((size_t)M + n);
    std::vector<double> cr(n), ci(n), br(M, 0.0), bi(M, 0.0);
    for (int k = 0; k < n; ++k) {
        const long q = ((long)k * k) % (2L * n);   // k^2 mod 2n keeps the angle small
        cr[k] = std::cos(pi * (double)q / n);
        ci[k] = -std::sin(pi * (double)q / n);
        out[(size_t)k] = c32{(float)cr[k], (float)ci[k]};
        br[k] = cr[k]; bi[k] = -ci[k];
        if (k) { br[M - k] = cr[k]; bi[M - k] = -ci[k]; }
    }
    // transform of b in double precision: the same mixed-radix Stockham passes as on the device, direct butterflies
    {
        std::vector<double> ar(br), ai(bi), cr2(M), ci2(M);
        int ns = 1;
        for (int pi_ = 0; pi_ < pl.nf; ++pi_) {
            const int R = pl.f[pi_], m = M / R;
            for (int j = 0; j < m; ++j) {
                const int jq = j / ns, jm = j - jq * ns, j0 = jq * ns * R + jm;
                for (int k = 0; k < R; ++k) {
                    double sr = 0.0, si = 0.0;
                    for (int t = 0; t < R; ++t) {
                        // twiddle of the pass and of the butterfly: exp(-2 pi i (t jm / (ns R) + t k / R))
                        const double ang = -2.0 * pi * ((double)t * jm / ((double)ns * R) + (double)((t * k) % R) / R);
                        const double wr = std::cos(ang), wi = std::sin(ang);
                        const double xr = ar[j + t * m], xi = ai[j + t * m];
                        sr += xr * wr - xi * wi;
                        si += xr * wi + xi * wr;
                    }
                    cr2[j0 + k * ns] = sr;
                    ci2[j0 + k * ns] = si;
                }
            }
            ar.swap(cr2);
            ai.swap(ci2);
            ns *= R;
        }
        br = ar;
        bi = ai;
    }
    for (int k = 0; k < M; ++k) out[(size_t)n + k] = c32{(float)br[k], (float)bi[k]};
    append_factors();
    const size_t at = out.size();   // the twiddle tables of the register-resident engine for M points (p3d_chirp.hip)
    out.resize(at + chirp_table_slots(M), c32{0.f, 0.f});
    chirp_build_tables(M, out.data() + at);
}

bool flex_supported(int n) { return n >= 2 && n <= GEN_MAX_N && gen_make_plan(n).nf > 0 && flex_factors(n).nf > 0 && pick_col_tile(n) > 0 && pick_row_lines(n) > 0; }
int flex_col_tile(int n) { return pick_col_tile(n); }

const LineOps* get_flex_ops()
{
    // tpl = 0 marks the flexible implementation: no packed mask words, no persistent row pass, no sparse-tile flags
    static const LineOps ops = {0, 0, 0, 0, &flex_row, &flex_col, &flex_no_pipe, 0, 0, nullptr, 0, nullptr, &flex_row_real};
    return &ops;
}

}  // namespace p3d
