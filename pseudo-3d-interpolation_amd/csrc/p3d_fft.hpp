// p3d_fft.hpp -- in-register radix butterflies and the LDS Stockham line-FFT engine (gfx950).
//
// Replaces, on the device, what the reference gets from numpy.fft.fft2 / ifft2 (pocketfft) at
// pseudo_3D_interpolation/cube_POCS_interpolation_3D.py:255-257, called from
// pseudo_3D_interpolation/functions/POCS.py:535, 592, 613.
//
// One "line" (a row or a column of an (iline, xline) slice) of N = 2^m points is transformed by
// TPL = N/PPT threads, each holding PPT = min(16, N) points in registers.  Canonical register
// layout, valid before the first and after the last pass of every transform:
//
//        register q of thread tl  <->  element  tl + TPL*q            (q = 0..PPT-1)
//
// so global loads/stores of a contiguous line are coalesced over tl, and two transforms can be
// chained through registers (forward -> threshold -> inverse, or inverse -> re-insertion ->
// forward) without touching LDS in between.  A pass of radix R (Ns = product of the radices
// before it) follows the Stockham autosort recipe
//        v[t]  = in[j + t*N/R] * w^(t*(j mod Ns)),  w = exp(-+2*pi*i/(Ns*R))
//        out[(j div Ns)*Ns*R + (j mod Ns) + k*Ns] = DFT_R(v)[k]
// and exchanges data through LDS only between passes.  Forward transforms use the radix
// sequence 16,16,..,rem and inverse transforms the reversed sequence rem,..,16,16: the last
// forward pass and the first inverse pass then touch the same elements per thread.
//
// Twiddles: ONE table per line length, tw[k] = exp(-2*pi*i*k/N), k = 0..N-1, evaluated in double
// and rounded once; every pass of both directions indexes it (the inverse conjugates).  In LDS it
// is padded by one slot per 32 entries so that the power-of-two strides of the passes spread over
// the banks.
#pragma once

#if defined(__HIPCC__)
#define P3D_HD __host__ __device__ __forceinline__
#define P3D_D __device__ __forceinline__
#else
#define P3D_HD inline
#define P3D_D inline
#endif

namespace p3d {

struct __attribute__((aligned(8))) c32 {
    float x, y;
};

P3D_HD c32 operator+(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
P3D_HD c32 operator-(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
// Complex product with the fused operations spelled out.  The library is compiled with -ffp-contract=off so
// that the SAME source expression yields the SAME bits in every kernel it is inlined into (the compiler would
// otherwise fuse multiply-adds differently from one kernel to the next, and a hard threshold turns a
// last-bit difference into a kept-or-zeroed coefficient).
//
// On the device the three primitives that need a lane swap -- complex product, a + i*b, a - i*b -- are written
// as packed-f32 instructions with op_sel / neg modifiers (CDNA3+ VOP3P: op_sel picks the 32-bit half of each
// 64-bit source for the low result, op_sel_hi for the high result).  hipcc emits v_pk_* for the plain sums but
// builds the swaps with v_mov pairs (13-16 % of all instructions of both passes before this).  Same arithmetic,
// same rounding as the portable expressions below.
#ifndef P3D_PK_ASM
#define P3D_PK_ASM 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && P3D_PK_ASM
typedef float p3d_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ p3d_f2 as_f2(c32 a) { return p3d_f2{a.x, a.y}; }
__device__ __forceinline__ c32 as_c32(p3d_f2 a) { return c32{a.x, a.y}; }
__device__ __forceinline__ c32 operator*(c32 a, c32 b)
{
    p3d_f2 t, r;
    // t = (a.y*b.y, a.y*b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(as_f2(a)), "v"(as_f2(b)));
    // r = (a.x*b.x - t.x, a.x*b.y + t.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=v"(r) : "v"(as_f2(a)), "v"(as_f2(b)), "v"(t));
    return as_c32(r);
}
// a * conj(b): the same two instructions, the sign of b.y carried by the operand modifiers (bit-identical to conjugating first)
__device__ __forceinline__ c32 mul_conj(c32 a, c32 b)
{
    p3d_f2 t, r;
    // t = (a.y*b.y, a.y*b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(as_f2(a)), "v"(as_f2(b)));
    // r = (a.x*b.x + t.x, -a.x*b.y + t.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(r) : "v"(as_f2(a)), "v"(as_f2(b)), "v"(t));
    return as_c32(r);
}
// a + i*b = (a.x - b.y, a.y + b.x);  a - i*b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ c32 add_ib(c32 a, c32 b)
{
    p3d_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(as_f2(a)), "v"(as_f2(b)));
    return as_c32(r);
}
__device__ __forceinline__ c32 sub_ib(c32 a, c32 b)
{
    p3d_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(as_f2(a)), "v"(as_f2(b)));
    return as_c32(r);
}
#else
P3D_HD c32 operator*(c32 a, c32 b)
{
    return {__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x)};
}
P3D_HD c32 mul_conj(c32 a, c32 b) { return a * c32{b.x, -b.y}; }
P3D_HD c32 add_ib(c32 a, c32 b) { return {a.x - b.y, a.y + b.x}; }
P3D_HD c32 sub_ib(c32 a, c32 b) { return {a.x + b.y, a.y - b.x}; }
#endif
// a*s + b*t with real s, t
P3D_HD c32 axpby(c32 a, float s, c32 b, float t) { return {__builtin_fmaf(a.x, s, b.x * t), __builtin_fmaf(a.y, s, b.y * t)}; }
P3D_HD c32 operator*(c32 a, float s) { return {a.x * s, a.y * s}; }

constexpr int FWD = -1;  // exp(-i...)  (numpy fft2)
constexpr int INV = +1;  // exp(+i...)  (numpy ifft2, unnormalised here; 1/N applied by the caller)

// multiply by DIR*i
template <int DIR>
P3D_HD c32 mul_i(c32 a)
{
    return DIR < 0 ? c32{a.y, -a.x} : c32{-a.y, a.x};
}

// ---- natural-order radix-2 / radix-4 -------------------------------------------------------
P3D_HD void dft2(c32& a, c32& b)
{
    c32 s = a + b;
    b = a - b;
    a = s;
}

template <int DIR>
P3D_HD void dft4(c32& v0, c32& v1, c32& v2, c32& v3)
{
    const c32 s0 = v0 + v2, d0 = v0 - v2;
    const c32 s1 = v1 + v3, t = v1 - v3;
    v0 = s0 + s1;
    v2 = s0 - s1;
    // v1 = d0 + DIR*i*t, v3 = d0 - DIR*i*t
    v1 = DIR > 0 ? add_ib(d0, t) : sub_ib(d0, t);
    v3 = DIR > 0 ? sub_ib(d0, t) : add_ib(d0, t);
}

// ---- in-place radix-R DFT; result k ends up at position digit_rev<R>(k) --------------------
template <int R>
constexpr int digit_rev(int k)
{
    return R == 16 ? (k >> 2) + 4 * (k & 3) : R == 8 ? (k >> 2) + 2 * (k & 3) : k;
}

template <int R, int DIR>
struct Dft;

template <int DIR>
struct Dft<1, DIR> {
    static P3D_HD void run(c32*) {}
};
template <int DIR>
struct Dft<2, DIR> {
    static P3D_HD void run(c32* a) { dft2(a[0], a[1]); }
};
template <int DIR>
struct Dft<4, DIR> {
    static P3D_HD void run(c32* a) { dft4<DIR>(a[0], a[1], a[2], a[3]); }
};
template <int DIR>
struct Dft<8, DIR> {
    static P3D_HD void run(c32* a)
    {
        constexpr float H = 0.70710678118654752440f;
        dft4<DIR>(a[0], a[2], a[4], a[6]);
        dft4<DIR>(a[1], a[3], a[5], a[7]);
        a[3] = a[3] * c32{H, DIR * H};   // W8^1
        a[5] = mul_i<DIR>(a[5]);         // W8^2
        a[7] = a[7] * c32{-H, DIR * H};  // W8^3
        dft2(a[0], a[1]);
        dft2(a[2], a[3]);
        dft2(a[4], a[5]);
        dft2(a[6], a[7]);
    }
};
template <int DIR>
struct Dft<16, DIR> {
    static P3D_HD void run(c32* a)
    {
        constexpr float H = 0.70710678118654752440f;
        constexpr float C = 0.92387953251128675613f;  // cos(pi/8)
        constexpr float S = 0.38268343236508977173f;  // sin(pi/8)
        dft4<DIR>(a[0], a[4], a[8], a[12]);
        dft4<DIR>(a[1], a[5], a[9], a[13]);
        dft4<DIR>(a[2], a[6], a[10], a[14]);
        dft4<DIR>(a[3], a[7], a[11], a[15]);
        // a[t1 + 4*k2] *= W16^(t1*k2)
        a[5] = a[5] * c32{C, DIR * S};     // 1
        a[9] = a[9] * c32{H, DIR * H};     // 2
        a[13] = a[13] * c32{S, DIR * C};   // 3
        a[6] = a[6] * c32{H, DIR * H};     // 2
        a[10] = mul_i<DIR>(a[10]);         // 4
        a[14] = a[14] * c32{-H, DIR * H};  // 6
        a[7] = a[7] * c32{S, DIR * C};     // 3
        a[11] = a[11] * c32{-H, DIR * H};  // 6
        a[15] = a[15] * c32{-C, -DIR * S}; // 9
        dft4<DIR>(a[0], a[1], a[2], a[3]);
        dft4<DIR>(a[4], a[5], a[6], a[7]);
        dft4<DIR>(a[8], a[9], a[10], a[11]);
        dft4<DIR>(a[12], a[13], a[14], a[15]);
    }
};

// ---- compile-time plan of a length-N line ----------------------------------------------------
constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

template <int N>
struct Plan {
    static_assert(N >= 2 && (N & (N - 1)) == 0, "power-of-two line length expected");
    static constexpr int LOG2 = ilog2(N);
    static constexpr int PPT = N >= 16 ? 16 : N;   // points per thread
    static constexpr int TPL = N / PPT;             // threads per line
    static constexpr int NFULL = LOG2 <= 4 ? 0 : LOG2 / 4;
    static constexpr int REM = LOG2 <= 4 ? N : (1 << (LOG2 % 4));
    static constexpr int NPASS = LOG2 <= 4 ? 1 : NFULL + (REM > 1 ? 1 : 0);

    // radix of pass p for direction DIR
    static constexpr int radix(int dir, int p)
    {
        if (LOG2 <= 4) return N;
        int pf = dir == FWD ? p : NPASS - 1 - p;  // position in the forward ordering
        return pf < NFULL ? 16 : REM;
    }
    // product of the radices before pass p
    static constexpr int ns(int dir, int p)
    {
        int r = 1;
        for (int q = 0; q < p; ++q) r *= radix(dir, q);
        return r;
    }
};

// ---- twiddle tables ----------------------------------------------------------------------------
// (b) per-pass ordered tables (row pass): for every pass p >= 1 of a direction, (R-1) rows of Ns entries, row t-1
//     holding exp(dir*2*pi*i*t*jm/(Ns*R)), jm = 0..Ns-1 -- the order in which neighbouring threads read
//     them, at compile-time offsets from one per-thread base (no address arithmetic, no extra registers).
//     Forward table first, inverse table at offset N.  Twice the LDS: used by the row pass, which is
//     limited by registers, not by LDS.
// P3D_EXP_HALFWG (experiment builds only, WRONG results): the inverse table aliases the forward one -- what a shared table would
// cost in LDS -- to time two 8-row workgroups per CU in row_pipe64_kernel (profiles/r02_rowpass_two_workgroups.txt)
#ifndef P3D_EXP_HALFWG
#define P3D_EXP_HALFWG 0
#endif
template <int N>
struct PassTables {
    static constexpr int off(int dir, int p)
    {
        int o = dir == FWD ? 0 : (P3D_EXP_HALFWG ? 0 : N);
        for (int q = 1; q < p; ++q) o += (Plan<N>::radix(dir, q) - 1) * Plan<N>::ns(dir, q);
        return o;
    }
    static constexpr int slots() { return P3D_EXP_HALFWG ? N : 2 * N; }
    static void build(c32* out)
    {
        using PL = Plan<N>;
        for (int i = 0; i < slots(); ++i) out[i] = c32{0.f, 0.f};
        for (int dir = -1; dir <= 1; dir += 2)
            for (int p = 1; p < PL::NPASS; ++p) {
                const int R = PL::radix(dir, p), NS = PL::ns(dir, p), o = off(dir, p);
                for (int t = 1; t < R; ++t)
                    for (int jm = 0; jm < NS; ++jm) {
                        const double ang = dir * 6.283185307179586476925286766559 * double(t) * double(jm) / (double(NS) * R);
                        out[o + (t - 1) * NS + jm] = c32{float(__builtin_cos(ang)), float(__builtin_sin(ang))};
                    }
            }
    }
};

// (c) column-pass tables.  The LAST pass of a direction has entry k = t*jm of the plain table exp(-2*pi*i*k/N) (conjugated
//     inside the multiply for the inverse): the lanes of a wave hold consecutive jm, so the strides t*8 B spread over the LDS
//     banks without padding and the address is ONE multiply.  The MIDDLE pass of a three-pass plan (k = t*STEP*jm, STEP >= 2)
//     would pile onto one bank there: it gets ordered rows like (b), (R-1)*NS entries per direction, with the very same
//     values (built from the same expression).  ~10 KiB at N = 1024, so two column tiles still share a CU's 160 KiB.
template <int N>
struct ColTables {
    using PL = Plan<N>;
    static constexpr int master() { return N >= 16 ? N - N / 16 + 1 : N; }   // (R-1)*(N/R - 1) < N - N/16 for every radix
    static constexpr bool has_mid() { return PL::NPASS == 3; }
    static constexpr int mid_len(int dir) { return has_mid() ? (PL::radix(dir, 1) - 1) * PL::ns(dir, 1) : 0; }
    static constexpr int mid_off(int dir) { return master() + (dir == FWD ? 0 : mid_len(FWD)); }
    static constexpr int slots() { return master() + mid_len(FWD) + mid_len(INV); }
    static c32 entry(int k)
    {
        const double ang = -6.283185307179586476925286766559 * double(k) / double(N);
        return c32{float(__builtin_cos(ang)), float(__builtin_sin(ang))};
    }
    static void build(c32* out)
    {
        for (int k = 0; k < master(); ++k) out[k] = entry(k);
        if constexpr (PL::NPASS == 3) {
            for (int dir = -1; dir <= 1; dir += 2) {
                const int R = PL::radix(dir, 1), NS = PL::ns(dir, 1), STEP = N / (NS * R);
                for (int t = 1; t < R; ++t)
                    for (int jm = 0; jm < NS; ++jm) {
                        c32 w = entry(t * STEP * jm);
                        if (dir > 0) w.y = -w.y;
                        out[mid_off(dir) + (t - 1) * NS + jm] = w;
                    }
            }
        }
    }
};

// twiddle policies for pass_compute: mul<N, DIR, P, T>(a, jm) = a * (twiddle t = T of pass P, position jm)
struct TwCol {
    const c32* tw;
    template <int N, int DIR, int P, int T>
    P3D_HD c32 mul(c32 a, int jm) const
    {
        using PL = Plan<N>;
        if constexpr (P + 1 == PL::NPASS) {
            const c32 w = tw[T * jm];
            return DIR > 0 ? mul_conj(a, w) : a * w;
        } else {
            return a * (tw + (ColTables<N>::mid_off(DIR) + (T - 1) * PL::ns(DIR, P)))[jm];
        }
    }
};
struct TwOrdered {
    const c32* tw;
    template <int N, int DIR, int P, int T>
    P3D_HD c32 mul(c32 a, int jm) const
    {
        return a * (tw + (PassTables<N>::off(DIR, P) + (T - 1) * Plan<N>::ns(DIR, P)))[jm];
    }
};

// ---- LDS views ---------------------------------------------------------------------------------
// One padding slot after every 16 points keeps the strided Stockham scatter (stride R*8 B) off a single
// bank.  pad(base + c) == pad(base) + c + (c >> 4) whenever (base mod 16) + (c mod 16) < 16; the engine only
// asks for such (base, c) pairs, so a thread needs ONE computed LDS address per scatter / gather and the
// 16 accesses use instruction-immediate offsets.
// Row view: one line is contiguous.
struct LdsRow {
    c32* base;
    static constexpr int stride(int n) { return n + (n >> 4); }
    P3D_HD c32& at(int pos) const { return base[pos + (pos >> 4)]; }
    P3D_HD c32* ptr(int pos) const { return base + pos + (pos >> 4); }
    static constexpr int rel(int c) { return c + (c >> 4); }
};
// Column-block view: W (<= 8) columns of one 64-byte column block interleaved, element-major
// ([pos][W]); `base` already points at this thread's column.
template <int W>
struct LdsColW {
    c32* base;
    static constexpr int stride(int n) { return (n + (n >> 4)) * W; }  // elements per column block
    P3D_HD c32& at(int pos) const { return base[(pos + (pos >> 4)) * W]; }
    P3D_HD c32* ptr(int pos) const { return base + (pos + (pos >> 4)) * W; }
    static constexpr int rel(int c) { return (c + (c >> 4)) * W; }
};
using LdsColBlk = LdsColW<8>;

// One pass, all in registers: twiddle, radix-R DFTs, re-order to "output k of butterfly s at
// register s + NB*k".
template <int N, int DIR, int P, int T, class TW>
struct TwApply {  // a[t] = v[s + NB*t] * w_t for t = T..R-1 (compile-time t: table offsets fold into the instruction)
    template <int R, int NB>
    static P3D_HD void run(c32* a, const c32* v, int s, TW tw, int jm)
    {
        if constexpr (T < R) {
            a[T] = tw.template mul<N, DIR, P, T>(v[s + NB * T], jm);
            TwApply<N, DIR, P, T + 1, TW>::template run<R, NB>(a, v, s, tw, jm);
        }
    }
};

template <int N, int DIR, int P, class TW>
P3D_HD void pass_compute(c32 (&v)[Plan<N>::PPT], TW tw, int tl)
{
    using PL = Plan<N>;
    constexpr int R = PL::radix(DIR, P);
    constexpr int NS = PL::ns(DIR, P);
    constexpr int NB = PL::PPT / R;
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        c32 a[R];
        const int jm = (tl + s * PL::TPL) & (NS - 1);
        a[0] = v[s];
        if constexpr (P == 0) {
#pragma unroll
            for (int t = 1; t < R; ++t) a[t] = v[s + NB * t];
        } else {
            TwApply<N, DIR, P, 1, TW>::template run<R, NB>(a, v, s, tw, jm);
        }
        Dft<R, DIR>::run(a);
#pragma unroll
        for (int k = 0; k < R; ++k) v[s + NB * k] = a[digit_rev<R>(k)];
    }
}

template <int N, int DIR, int P, class LDS>
P3D_HD void pass_scatter(const c32 (&v)[Plan<N>::PPT], LDS lds, int tl)
{
    using PL = Plan<N>;
    constexpr int R = PL::radix(DIR, P);
    constexpr int NS = PL::ns(DIR, P);
    constexpr int NB = PL::PPT / R;
    // j0 = hi*NS*R + lo with lo < NS: (j0 mod 16) + (k*NS mod 16) <= 15 whenever NS*R is a multiple of 16
    constexpr bool FOLD = (NS * R) % 16 == 0 && (NS >= 16 || 16 % NS == 0);
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        const int jb = tl + s * PL::TPL;
        const int j0 = (jb & ~(NS - 1)) * R + (jb & (NS - 1));
        if constexpr (FOLD) {
            c32* p = lds.ptr(j0);
#pragma unroll
            for (int k = 0; k < R; ++k) p[LDS::rel(k * NS)] = v[s + NB * k];
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) lds.at(j0 + k * NS) = v[s + NB * k];
        }
    }
}

template <int N, class LDS>
P3D_HD void canonical_gather(c32 (&v)[Plan<N>::PPT], LDS lds, int tl)
{
    constexpr int TPL = Plan<N>::TPL;
    if constexpr (TPL % 16 == 0) {
        const c32* p = lds.ptr(tl);
#pragma unroll
        for (int q = 0; q < Plan<N>::PPT; ++q) v[q] = p[LDS::rel(TPL * q)];
    } else {
#pragma unroll
        for (int q = 0; q < Plan<N>::PPT; ++q) v[q] = lds.at(tl + TPL * q);
    }
}

#if defined(__HIPCC__)
// Exchange synchronisation.  WAVE = true: every line lives inside one wavefront (TPL <= 64), LDS
// operations of one wave execute in program order, so only the compiler has to be kept from
// reordering them; false: workgroup barrier.
template <bool WAVE>
P3D_D void exchange_sync()
{
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int N, int DIR, int P, bool WAVE, class LDS, class TW>
struct PassLoop {
    static P3D_D void run(c32 (&v)[Plan<N>::PPT], LDS lds, TW tw, int tl)
    {
        if constexpr (P > 0) canonical_gather<N>(v, lds, tl);
        pass_compute<N, DIR, P>(v, tw, tl);
        if constexpr (P + 1 < Plan<N>::NPASS) {
            exchange_sync<WAVE>();  // everybody is done reading the previous contents
            pass_scatter<N, DIR, P>(v, lds, tl);
            exchange_sync<WAVE>();
            PassLoop<N, DIR, P + 1, WAVE, LDS, TW>::run(v, lds, tw, tl);
        }
    }
};

// Transform one line held in canonical register layout; result is canonical again.
// Every thread of the workgroup (WAVE = false) / wavefront (WAVE = true) must call this.
template <int N, int DIR, bool WAVE, class LDS, class TW>
P3D_D void line_fft(c32 (&v)[Plan<N>::PPT], LDS lds, TW tw, int tl)
{
    PassLoop<N, DIR, 0, WAVE, LDS, TW>::run(v, lds, tw, tl);
}
#endif  // __HIPCC__

}  // namespace p3d
