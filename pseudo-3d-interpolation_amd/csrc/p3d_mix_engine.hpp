// p3d_mix_engine.hpp -- the mixed-radix register-resident line-FFT engine, generic over the element type (c32 / c64d).  Design notes: the head of
// p3d_mix.hpp (the float32 passes built on it); the double-precision passes are in p3d_mix64.hip.
#pragma once

#include "p3d_kernels_common.hpp"

namespace p3d {
namespace mix {

// double-precision complex numbers: the engine below is generic over the element type (c32: the float32 passes of p3d_mix.hpp; c64d: the loop in
// the reference's double precision, p3d_mix64.hip)
struct __attribute__((aligned(16))) c64d {
    double x, y;
};
P3D_HD c64d operator+(c64d a, c64d b) { return {a.x + b.x, a.y + b.y}; }
P3D_HD c64d operator-(c64d a, c64d b) { return {a.x - b.x, a.y - b.y}; }
P3D_HD c64d operator*(c64d a, c64d b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
P3D_HD c64d operator*(c64d a, double s) { return {a.x * s, a.y * s}; }
P3D_HD c64d mul_conj(c64d a, c64d b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a * conj(b)
P3D_HD c64d add_ib(c64d a, c64d b) { return {a.x - b.y, a.y + b.x}; }                              // a + i b
P3D_HD c64d sub_ib(c64d a, c64d b) { return {a.x + b.y, a.y - b.x}; }                              // a - i b
template <int DIR>
P3D_HD c64d mul_i(c64d a) { return DIR < 0 ? c64d{a.y, -a.x} : c64d{-a.y, a.x}; }

template <int R>
struct Roots;    // cos / sin of 2 pi q / R as float ...
template <int R>
struct RootsD;   // ... and as double
#include "p3d_mix_roots.inc"

// W^q, W = exp(DIR 2 pi i / R), in the precision of C
template <class C, int R, int DIR>
__device__ __forceinline__ C root(int q)
{
    if constexpr (sizeof(C) == sizeof(c32)) return C{Roots<R>::c[q], DIR > 0 ? Roots<R>::s[q] : -Roots<R>::s[q]};
    else return C{RootsD<R>::c[q], DIR > 0 ? RootsD<R>::s[q] : -RootsD<R>::s[q]};
}

// ---- small DFTs on registers, natural order in and out: X[k] = sum_t x[t] W^(t k), W = exp(DIR 2 pi i / R) ------------------------------
constexpr int first_factor(int r) { return r % 4 == 0 ? 4 : r % 2 == 0 ? 2 : r % 3 == 0 ? 3 : r % 5 == 0 ? 5 : r % 7 == 0 ? 7 : r % 11 == 0 ? 11 : r % 13 == 0 ? 13 : r; }

template <class C, int R, int DIR>
struct SmallDft {
    static __device__ __forceinline__ void run(C (&x)[R])
    {
        constexpr int R1 = first_factor(R);
        if constexpr (R == 1) {
        } else if constexpr (R == 2) {
            const C a = x[0] + x[1], b = x[0] - x[1];
            x[0] = a;
            x[1] = b;
        } else if constexpr (R == 4) {
            const C a = x[0] + x[2], b = x[0] - x[2], s = x[1] + x[3], t = x[1] - x[3];
            x[0] = a + s;
            x[2] = a - s;
            x[1] = DIR > 0 ? add_ib(b, t) : sub_ib(b, t);
            x[3] = DIR > 0 ? sub_ib(b, t) : add_ib(b, t);
        } else if constexpr (R1 == R) {
            // odd prime: pair x[q] with x[R-q]:  X[k], X[R-k] = A_k +- i B_k,  A_k = x0 + sum_q cos(2 pi q k / R) (x[q] + x[R-q]),
            // B_k = sum_q (DIR sin(2 pi q k / R)) (x[q] - x[R-q]) -- real coefficients only
            constexpr int H = (R - 1) / 2;
            C sp[H], dm[H];
            C x0 = x[0];
#pragma unroll
            for (int q = 1; q <= H; ++q) {
                sp[q - 1] = x[q] + x[R - q];
                dm[q - 1] = x[q] - x[R - q];
                x0 = x0 + sp[q - 1];
            }
            const C xin = x[0];
            x[0] = x0;
#pragma unroll
            for (int k = 1; k <= H; ++k) {
                C A = xin, B{0, 0};
#pragma unroll
                for (int q = 1; q <= H; ++q) {
                    const C wq = root<C, R, DIR>((q * k) % R);
                    A = A + sp[q - 1] * wq.x;
                    B = B + dm[q - 1] * wq.y;
                }
                x[k] = add_ib(A, B);
                x[R - k] = sub_ib(A, B);
            }
        } else {
            // R = R1 R2, t = R2 t1 + t2, k = k1 + R1 k2:  X[k] = sum_t2 [ (sum_t1 x[t] W_R1^(t1 k1)) W_R^(t2 k1) ] W_R2^(t2 k2)
            constexpr int R2 = R / R1;
            C u[R];   // u[k1 R2 + t2]
#pragma unroll
            for (int t2 = 0; t2 < R2; ++t2) {
                C a[R1];
#pragma unroll
                for (int t1 = 0; t1 < R1; ++t1) a[t1] = x[R2 * t1 + t2];
                SmallDft<C, R1, DIR>::run(a);
#pragma unroll
                for (int k1 = 0; k1 < R1; ++k1) {
                    const int e = (k1 * t2) % R;
                    if (e == 0) u[k1 * R2 + t2] = a[k1];
                    else if (4 * e == R) u[k1 * R2 + t2] = mul_i<DIR>(a[k1]);            // W^(R/4) = DIR i
                    else if (2 * e == R) u[k1 * R2 + t2] = C{-a[k1].x, -a[k1].y};        // W^(R/2) = -1
                    else if (4 * e == 3 * R) u[k1 * R2 + t2] = mul_i<-DIR>(a[k1]);        // W^(3R/4) = -DIR i
                    else u[k1 * R2 + t2] = a[k1] * root<C, R, DIR>(e);
                }
            }
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) {
                C b[R2];
#pragma unroll
                for (int t2 = 0; t2 < R2; ++t2) b[t2] = u[k1 * R2 + t2];
                SmallDft<C, R2, DIR>::run(b);
#pragma unroll
                for (int k2 = 0; k2 < R2; ++k2) x[k1 + R1 * k2] = b[k2];
            }
        }
    }
};

// ---- compile-time plan ---------------------------------------------------------------------------------------------------------------
// Forward passes p = 0 ... NPASS-1 have radix R_p and B_p butterflies per thread, i.e. PPT_p = R_p B_p points per thread and TPL_p = N / PPT_p
// active threads per line IN THAT PASS: the data goes round through LDS between two passes anyway, so every pass picks the split that keeps
// ~16-24 points in registers whatever its radix (960 = 15 x 8 x 8 runs as 64 x 15, 60 x 16, 60 x 16 points; one common PPT would have to be
// 120).  The inverse transform runs the passes in REVERSED order: its first pass has the layout of the forward transform's last one, so
// forward -> threshold -> inverse (column pass) and inverse -> re-insertion -> forward (row pass) chain through registers.  Two layouts
// therefore meet global memory:  A = (PPT_0, TPL_0), the first forward pass -- cubes, masks, compact samples;  B = (PPT_last, TPL_last) --
// the work buffer between the two passes of an iteration.
template <int N_, int COLT_, int ROWLB_, int NPASS_, int R0, int B0, int R1, int B1, int R2, int B2, int R3, int B3>
struct MixPlan {
    static constexpr int N = N_, NPASS = NPASS_;
    static constexpr int COLT = COLT_;     // columns per workgroup of the column pass
    static constexpr int ROWLB = ROWLB_;   // rows per workgroup of the row pass
    static constexpr int fr(int p) { return p == 0 ? R0 : p == 1 ? R1 : p == 2 ? R2 : R3; }
    static constexpr int fb(int p) { return p == 0 ? B0 : p == 1 ? B1 : p == 2 ? B2 : B3; }
    static constexpr int fwd_index(int dir, int p) { return dir == FWD ? p : NPASS_ - 1 - p; }
    static constexpr int radix(int dir, int p) { return fr(fwd_index(dir, p)); }
    static constexpr int nb(int dir, int p) { return fb(fwd_index(dir, p)); }
    static constexpr int ppt(int dir, int p) { return radix(dir, p) * nb(dir, p); }
    static constexpr int tpl(int dir, int p) { return N_ / ppt(dir, p); }
    static constexpr int ns(int dir, int p)
    {
        int r = 1;
        for (int q = 0; q < p; ++q) r *= radix(dir, q);
        return r;
    }
    static_assert(ns(FWD, NPASS_) == N_, "the radices must multiply to N");
    static_assert(N_ % (R0 * B0) == 0 && N_ % (R1 * B1) == 0 && N_ % (R2 * B2) == 0 && N_ % (R3 * B3) == 0, "every pass splits the line evenly");
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int VMAX = cmax(cmax(R0 * B0, R1 * B1), cmax(NPASS_ > 2 ? R2 * B2 : 1, NPASS_ > 3 ? R3 * B3 : 1));   // registers (points) per thread
    static constexpr int tmax()
    {
        int m = R0 * B0;
        for (int p = 1; p < NPASS_; ++p) m = (fr(p) * fb(p) < m) ? fr(p) * fb(p) : m;
        return N_ / m;
    }
    static constexpr int TMAX = tmax();   // threads per line
    // a radix with a factor 11 or 13 (paired-form butterflies: register hungry).  Such plans are used for COLUMNS only: their row passes measured
    // slower than the in-place LDS-image rows of p3d_flex.hip (1430-point rows 2.83 vs 1.76 ms, 2002: 3.20 vs 2.86, 990: 1.62 vs 1.38 per 128 x 1024
    // rows; 770 and 1573 would gain 10-20 %), their column passes 7-40 % faster (1573 points: 0.89 vs 1.44 ms)
    static constexpr bool big_prime()
    {
        for (int p = 0; p < NPASS_; ++p)
            if (fr(p) % 11 == 0 || fr(p) % 13 == 0) return true;
        return false;
    }
    static constexpr bool BIG_PRIME = big_prime();
    static constexpr int PPT_A = ppt(FWD, 0), TPL_A = tpl(FWD, 0), PPT_B = ppt(FWD, NPASS_ - 1), TPL_B = tpl(FWD, NPASS_ - 1);
    // Padding of a line's LDS image, per direction: one slot per PADQ positions where the first radix of the direction is even (its scatter
    // then walks the banks with the odd stride R + 1; an odd radix does so by itself).  Offsets fold into the instructions wherever PADQ
    // divides the constant part of a position.
    static constexpr int padq(int dir) { return (radix(dir, 0) % 2 == 0 && radix(dir, 0) > 2) ? radix(dir, 0) : 0; }
    static constexpr int pad(int dir, int pos) { return padq(dir) ? pos + pos / padq(dir) : pos; }
    static constexpr int LINE = cmax(pad(FWD, N_), pad(INV, N_));   // slots of one line's LDS image
    // Twiddles.  The LAST pass of either direction (Ns = N / R) multiplies by exp(-+2 pi i t jm / N): both directions read ONE master table
    // exp(-2 pi i k / N) at k = t jm (conjugated inside the multiply for the inverse transform), padded by one slot per 32 entries so that the
    // strides t spread over the banks.  The MIDDLE passes (1 <= p < NPASS - 1) have (R - 1) rows of Ns entries exp(-2 pi i t jm / (Ns R)) each,
    // in the order neighbouring threads read them, per direction.  (Ordered rows for the last passes too would take ~N entries per direction:
    // with a 64-KiB column tile that is the difference between two workgroups per CU and one.)
    static constexpr int master_idx(int k) { return k + (k >> 5); }
    static constexpr int MASTER = NPASS_ > 1 ? master_idx(N_) + 1 : 0;
    static constexpr int mid_len(int dir)
    {
        int o = 0;
        for (int q = 1; q + 1 < NPASS_; ++q) o += (radix(dir, q) - 1) * ns(dir, q);
        return o;
    }
    static constexpr int tw_off(int dir, int p)   // rows of middle pass p
    {
        int o = MASTER + (dir == FWD ? 0 : mid_len(FWD));
        for (int q = 1; q < p; ++q) o += (radix(dir, q) - 1) * ns(dir, q);
        return o;
    }
    static constexpr int TW_SLOTS = MASTER + mid_len(FWD) + mid_len(INV);
    template <class C>
    static void build_tw(C* out)
    {
        using S = decltype(C{}.x);
        for (int i = 0; i < TW_SLOTS; ++i) out[i] = C{0, 0};
        if (NPASS > 1)
            for (int k = 0; k < N; ++k) {
                const double ang = -6.283185307179586476925286766559 * double(k) / double(N);
                out[master_idx(k)] = C{S(__builtin_cos(ang)), S(__builtin_sin(ang))};
            }
        for (int dir = -1; dir <= 1; dir += 2)
            for (int p = 1; p + 1 < NPASS; ++p) {
                const int R = radix(dir, p), NS = ns(dir, p), o = tw_off(dir, p);
                for (int t = 1; t < R; ++t)
                    for (int jm = 0; jm < NS; ++jm) {
                        const double ang = -6.283185307179586476925286766559 * double(t) * double(jm) / (double(NS) * R);
                        out[o + (t - 1) * NS + jm] = C{S(__builtin_cos(ang)), S(__builtin_sin(ang))};
                    }
            }
    }
};

// LDS view of one line for the transforms of one direction: W lines interleaved element-major ([pos][W]); `base` points at this thread's line
template <class PL, int DIR, int W, class C = c32>
struct Lds {
    static constexpr int WIDTH = W, PADQ = PL::padq(DIR);
    C* base;
    __device__ __forceinline__ C* ptr(int pos) const { return base + (PADQ ? pos + pos / (PADQ ? PADQ : 1) : pos) * W; }
    static constexpr int rel(int c) { return PL::pad(DIR, c) * W; }   // pad(p + c) == pad(p) + pad(c) when PADQ | c
    static constexpr bool folds(int c) { return PADQ == 0 || c % (PADQ ? PADQ : 1) == 0; }
};

template <class PL, int DIR, int P, int T, class C>
struct TwApply {   // a[t] = v[s + NB t] * w_t, t = T ... R-1 (compile-time t: the row offsets fold into the instruction)
    template <int R, int NB>
    static __device__ __forceinline__ void run(C* a, const C* v, int s, const C* tw, int jm)
    {
        if constexpr (T < R) {
            C w;
            if constexpr (P + 1 == PL::NPASS) {
                const int k = T * jm;
                w = tw[k + (k >> 5)];                                           // the master table (last pass)
            } else {
                w = (tw + (PL::tw_off(DIR, P) + (T - 1) * PL::ns(DIR, P)))[jm];   // ordered rows (middle pass)
            }
            a[T] = DIR > 0 ? mul_conj(v[s + NB * T], w) : v[s + NB * T] * w;
            TwApply<PL, DIR, P, T + 1, C>::template run<R, NB>(a, v, s, tw, jm);
        }
    }
};

// one pass in registers: twiddles, NB radix-R DFTs; output k of butterfly s ends up in register s + NB k.  Threads tl >= TPL_p sit the pass out.
template <class PL, int DIR, int P, class C>
__device__ __forceinline__ void pass_compute(C (&v)[PL::VMAX], const C* tw, int tl)
{
    constexpr int R = PL::radix(DIR, P), NS = PL::ns(DIR, P), NB = PL::nb(DIR, P), TPL = PL::tpl(DIR, P);
    if (TPL < PL::TMAX && tl >= TPL) return;
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        C a[R];
        a[0] = v[s];
        if constexpr (P == 0) {
#pragma unroll
            for (int t = 1; t < R; ++t) a[t] = v[s + NB * t];
        } else {
            const int jm = (tl + s * TPL) % NS;
            TwApply<PL, DIR, P, 1, C>::template run<R, NB>(a, v, s, tw, jm);
        }
        SmallDft<C, R, DIR>::run(a);
#pragma unroll
        for (int k = 0; k < R; ++k) v[s + NB * k] = a[k];
    }
}

template <class PL, int DIR, int P, class LDS, class C>
__device__ __forceinline__ void pass_scatter(const C (&v)[PL::VMAX], LDS lds, int tl)
{
    constexpr int R = PL::radix(DIR, P), NS = PL::ns(DIR, P), NB = PL::nb(DIR, P), TPL = PL::tpl(DIR, P);
    if (TPL < PL::TMAX && tl >= TPL) return;
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        const int jb = tl + s * TPL;
        const int j0 = (jb / NS) * (NS * R) + jb % NS;
        if constexpr (LDS::folds(NS)) {
            C* p = lds.ptr(j0);
#pragma unroll
            for (int k = 0; k < R; ++k) p[LDS::rel(k * NS)] = v[s + NB * k];
        } else if constexpr (P == 0 && R == LDS::PADQ) {
            C* p = lds.ptr(j0);   // j0 = jb R0: a multiple of PADQ, and k < PADQ adds no padding slot
#pragma unroll
            for (int k = 0; k < R; ++k) p[k * LDS::WIDTH] = v[s + NB * k];
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) *lds.ptr(j0 + k * NS) = v[s + NB * k];
        }
    }
}

// the inputs of pass P from the LDS image: register q <- element tl + TPL_P q
template <class PL, int DIR, int P, class LDS, class C>
__device__ __forceinline__ void pass_gather(C (&v)[PL::VMAX], LDS lds, int tl)
{
    constexpr int PPT = PL::ppt(DIR, P), TPL = PL::tpl(DIR, P);
    if (TPL < PL::TMAX && tl >= TPL) return;
    if constexpr (LDS::folds(TPL)) {
        const C* p = lds.ptr(tl);
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = p[LDS::rel(TPL * q)];
    } else {
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = *lds.ptr(tl + TPL * q);
    }
}

template <class PL, int DIR, int P, class LDS, class C>
struct PassLoop {
    static __device__ __forceinline__ void run(C (&v)[PL::VMAX], LDS lds, const C* tw, int tl)
    {
        if constexpr (P > 0) pass_gather<PL, DIR, P>(v, lds, tl);
        pass_compute<PL, DIR, P>(v, tw, tl);
        if constexpr (P + 1 < PL::NPASS) {
            __syncthreads();   // everybody is done reading the previous contents
            pass_scatter<PL, DIR, P>(v, lds, tl);
            __syncthreads();
            PassLoop<PL, DIR, P + 1, LDS, C>::run(v, lds, tw, tl);
        }
    }
};

// Transform one line: in = layout of the direction's first pass (FWD: A, INV: B), out = layout of its last pass (FWD: B, INV: A).
// Every thread of the workgroup must call this.  `image` points at this thread's line of the LDS tile (W lines interleaved).
template <class PL, int DIR, int W, class C>
__device__ __forceinline__ void line_fft(C (&v)[PL::VMAX], C* image, const C* tw, int tl)
{
    PassLoop<PL, DIR, 0, Lds<PL, DIR, W, C>, C>::run(v, Lds<PL, DIR, W, C>{image}, tw, tl);
}

}  // namespace mix
}  // namespace p3d
