// p3d_fft32.hpp -- the 1024-point transform as 32 x 32 (two in-register radix-32 butterflies around one exchange): constants, twiddle table and the
// two halves as one lane sees them.  Host-compilable (tests/csrc/test_fft_host.cpp drives the halves lane by lane on the CPU); the kernels that use it
// are in p3d_row_pipe32.hpp.
#pragma once

#include "p3d_fft.hpp"

namespace p3d {

struct P32 {
    static constexpr int N = 1024, ROWS = 16, THREADS = 512, UPB = 8;   // rows / threads / units (row pairs) per workgroup
    static constexpr int LSTR = 1024 + 2 * 32;   // a row's exchange buffer: butterfly j's 32 outputs at 34 j (two padding slots: 16-byte stores, banks spread)
    static constexpr int TW = 31 * 32;           // exp(-2 pi i t j / 1024), t = 1 ... 31, j = 0 ... 31 at (t - 1) * 32 + j; the inverse conjugates
    static constexpr size_t lds_bytes() { return sizeof(c32) * (TW + (size_t)ROWS * LSTR); }
    static void build_tw(c32* out)
    {
        for (int t = 1; t < 32; ++t)
            for (int j = 0; j < 32; ++j) {
                const double ang = -6.283185307179586476925286766559 * double(t) * double(j) / 1024.0;
                out[(t - 1) * 32 + j] = c32{float(__builtin_cos(ang)), float(__builtin_sin(ang))};
            }
    }
};

// ---- in-register radix-32 butterfly, natural order in and out: two radix-16 butterflies (even / odd inputs) and one radix-2 stage ----
// The stage's twiddles W32^k, k = 1 ... 15, are written with SEVEN complex constants: the inverse multiplies by the conjugate through the
// operand modifiers of the same instruction pair (mul_conj), and W32^(k + 8) = W32^k * (-+i) folds into the butterfly's sum (a -+ i b is
// one packed add).  A packed multiply takes its constant from a scalar register PAIR, the compiler keeps every distinct pair alive across the
// loop, and thirty of them -- one per (k, direction) -- pushed the kernel's mask words out of the scalar file into v_readlane traffic.
// multiply by a FORWARD twiddle w, or by its conjugate for the inverse (same instruction pair, the sign in the operand modifiers)
template <int DIR>
P3D_HD c32 mul_tw(c32 a, c32 w) { return DIR > 0 ? mul_conj(a, w) : a * w; }

// Dft<16, DIR> (p3d_fft.hpp) with its nine twiddles written on five forward constants shared by both directions (same reason as below;
// output k at position digit_rev<16>(k) as there).  Not the same roundings as Dft<16>: this kernel family has no bitwise contract with the others.
template <int DIR>
P3D_HD void dft16s(c32* a)
{
    constexpr float H = 0.70710678118654752440f, C = 0.92387953251128675613f, S = 0.38268343236508977173f;
    dft4<DIR>(a[0], a[4], a[8], a[12]);
    dft4<DIR>(a[1], a[5], a[9], a[13]);
    dft4<DIR>(a[2], a[6], a[10], a[14]);
    dft4<DIR>(a[3], a[7], a[11], a[15]);
    a[5] = mul_tw<DIR>(a[5], c32{C, -S});     // W16^1
    a[9] = mul_tw<DIR>(a[9], c32{H, -H});     // W16^2
    a[13] = mul_tw<DIR>(a[13], c32{S, -C});   // W16^3
    a[6] = mul_tw<DIR>(a[6], c32{H, -H});     // 2
    a[10] = mul_i<DIR>(a[10]);                // 4
    a[14] = mul_tw<DIR>(a[14], c32{-H, -H});  // 6
    a[7] = mul_tw<DIR>(a[7], c32{S, -C});     // 3
    a[11] = mul_tw<DIR>(a[11], c32{-H, -H});  // 6
    a[15] = mul_tw<DIR>(a[15], c32{-C, S});   // 9
    dft4<DIR>(a[0], a[1], a[2], a[3]);
    dft4<DIR>(a[4], a[5], a[6], a[7]);
    dft4<DIR>(a[8], a[9], a[10], a[11]);
    dft4<DIR>(a[12], a[13], a[14], a[15]);
}

template <int DIR>
P3D_HD void dft32(c32 (&x)[32])
{
    constexpr float C[8] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                            0.38268343236508977173f, 0.19509032201612826785f};   // cos(k pi / 16)
    constexpr float S[8] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f, 0.83146961230254523708f,
                            0.92387953251128675613f, 0.98078528040323044913f};   // sin(k pi / 16)
    c32 e[16], o[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { e[m] = x[2 * m]; o[m] = x[2 * m + 1]; }
    dft16s<DIR>(e);
    dft16s<DIR>(o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // outputs k, k + 16 (twiddle W32^k) and k + 8, k + 24 (twiddle W32^k * W32^8, W32^8 = DIR * i)
        const c32 ea = e[digit_rev<16>(k)], eb = e[digit_rev<16>(k + 8)];
        c32 oa = o[digit_rev<16>(k)], ob = o[digit_rev<16>(k + 8)];
        if (k != 0) {
            const c32 w{C[k], -S[k]};   // exp(-2 pi i k / 32); the inverse conjugates inside the product
            oa = mul_tw<DIR>(oa, w);
            ob = mul_tw<DIR>(ob, w);
        }
        x[k] = ea + oa;
        x[k + 16] = ea - oa;
        // eb + (DIR i) ob, eb - (DIR i) ob
        x[k + 8] = DIR > 0 ? add_ib(eb, ob) : sub_ib(eb, ob);
        x[k + 24] = DIR > 0 ? sub_ib(eb, ob) : add_ib(eb, ob);
    }
}

// The two halves of a 1024-point transform of one row, as one lane j (of 32) sees them; `row` is the row's exchange buffer.
//   first half:  A[j][k1] = sum_t x[j + 32 t] W32^(t k1)                                  -> row[34 j + k1]
//   second half: X[k1 + 32 k2] = sum_j (A[j][k1] W1024^(j k1)) W32^(j k2), lane = k1       -> register k2: canonical layout again
// (written apart so that tests/csrc can drive them lane by lane on the CPU)
template <int DIR>
P3D_HD void p32_half1(c32 (&v)[32], c32* row, int j)
{
    dft32<DIR>(v);
    c32* const p = row + 34 * j;
#pragma unroll
    for (int m = 0; m < 32; ++m) p[m] = v[m];
}
template <int DIR>
P3D_HD void p32_half2(c32 (&v)[32], const c32* row, const c32* tw, int j)
{
    const c32* const q = row + j;
    v[0] = q[0];
#pragma unroll
    for (int t = 1; t < 32; ++t) {
        const c32 w = tw[(t - 1) * 32 + j];
        v[t] = DIR > 0 ? mul_conj(q[34 * t], w) : q[34 * t] * w;
    }
    dft32<DIR>(v);
}

}  // namespace p3d
