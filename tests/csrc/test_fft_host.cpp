// Host-side emulation of the device line-FFT engine (tests/test_host_fft.py builds and runs it):
// the SAME templates the kernels use (butterflies, pass schedule, Stockham scatter, twiddle table
// layout, LDS views) are driven thread-by-thread and phase-by-phase on the CPU and compared with a
// naive double-precision DFT.  Barriers are emulated by finishing a phase for all threads first.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "p3d_fft.hpp"
#include "p3d_fft32.hpp"

using namespace p3d;
using cd = std::complex<double>;

template <int N, int DIR, int P, class MakeLds, class TW>
struct Emu {
    static void run(std::vector<std::vector<c32>>& regs, MakeLds mk, TW tab)
    {
        using PL = Plan<N>;
        for (int tl = 0; tl < PL::TPL; ++tl) {
            c32(&v)[PL::PPT] = *reinterpret_cast<c32(*)[PL::PPT]>(regs[tl].data());
            if constexpr (P > 0) canonical_gather<N>(v, mk(), tl);
            pass_compute<N, DIR, P>(v, tab, tl);
        }
        if constexpr (P + 1 < PL::NPASS) {
            for (int tl = 0; tl < PL::TPL; ++tl) {
                c32(&v)[PL::PPT] = *reinterpret_cast<c32(*)[PL::PPT]>(regs[tl].data());
                pass_scatter<N, DIR, P>(v, mk(), tl);
            }
            Emu<N, DIR, P + 1, MakeLds, TW>::run(regs, mk, tab);
        }
    }
};

template <int N, int DIR, bool COLVIEW, bool ORDERED = false>
double check()
{
    using PL = Plan<N>;
    std::vector<cd> x(N);
    for (int i = 0; i < N; ++i) x[i] = cd(std::sin(0.37 * i * i + 0.1) + 0.25, std::cos(1.3 * i) - 0.5 * (i % 3));
    std::vector<c32> tab(ORDERED ? PassTables<N>::slots() : ColTables<N>::slots());
    if (ORDERED) PassTables<N>::build(tab.data());
    else ColTables<N>::build(tab.data());
    using TW = std::conditional_t<ORDERED, TwOrdered, TwCol>;
    const TW tw{tab.data()};
    std::vector<std::vector<c32>> regs(PL::TPL, std::vector<c32>(PL::PPT));
    for (int tl = 0; tl < PL::TPL; ++tl)
        for (int q = 0; q < PL::PPT; ++q) regs[tl][q] = c32{float(x[tl + PL::TPL * q].real()), float(x[tl + PL::TPL * q].imag())};
    std::vector<c32> lds(COLVIEW ? LdsColBlk::stride(N) : LdsRow::stride(N) + 1);
    if constexpr (COLVIEW) {
        auto mk = [&]() { return LdsColBlk{lds.data() + 3}; };
        Emu<N, DIR, 0, decltype(mk), TW>::run(regs, mk, tw);
    } else {
        auto mk = [&]() { return LdsRow{lds.data()}; };
        Emu<N, DIR, 0, decltype(mk), TW>::run(regs, mk, tw);
    }
    double err = 0, nrm = 0;
    for (int k = 0; k < N; ++k) {
        cd acc = 0;
        for (int n = 0; n < N; ++n) acc += x[n] * std::polar(1.0, DIR * 2 * M_PI * double((long long)n * k % N) / N);
        const c32 g = regs[k % PL::TPL][k / PL::TPL];
        err += std::norm(acc - cd(g.x, g.y));
        nrm += std::norm(acc);
    }
    return std::sqrt(err / nrm);
}

template <int R, int DIR>
double check_dft()
{
    c32 a[R];
    cd x[R];
    for (int i = 0; i < R; ++i) {
        x[i] = cd(0.3 * i - 1.0 + 0.01 * i * i, 0.7 - 0.11 * i);
        a[i] = c32{float(x[i].real()), float(x[i].imag())};
    }
    Dft<R, DIR>::run(a);
    double err = 0;
    for (int k = 0; k < R; ++k) {
        cd acc = 0;
        for (int n = 0; n < R; ++n) acc += x[n] * std::polar(1.0, DIR * 2 * M_PI * n * k / R);
        const c32 g = a[digit_rev<R>(k)];
        err = std::fmax(err, std::abs(acc - cd(g.x, g.y)));
    }
    return err;
}

// the one-exchange 1024-point transform of row_pipe32_kernel (p3d_fft32.hpp): 32 lanes, 32 points each, first half -> exchange buffer -> second half
template <int DIR>
double check_32x32()
{
    constexpr int N = P32::N;
    std::vector<cd> x(N);
    for (int i = 0; i < N; ++i) x[i] = cd(std::sin(0.37 * i * i + 0.1) + 0.25, std::cos(1.3 * i) - 0.5 * (i % 3));
    std::vector<c32> tab(P32::TW), row(P32::LSTR);
    P32::build_tw(tab.data());
    std::vector<std::vector<c32>> regs(32, std::vector<c32>(32));
    for (int j = 0; j < 32; ++j)
        for (int k = 0; k < 32; ++k) regs[j][k] = c32{float(x[j + 32 * k].real()), float(x[j + 32 * k].imag())};
    for (int j = 0; j < 32; ++j) p32_half1<DIR>(*reinterpret_cast<c32(*)[32]>(regs[j].data()), row.data(), j);
    for (int j = 0; j < 32; ++j) p32_half2<DIR>(*reinterpret_cast<c32(*)[32]>(regs[j].data()), row.data(), tab.data(), j);
    double err = 0, nrm = 0;
    for (int k = 0; k < N; ++k) {
        cd acc = 0;
        for (int n = 0; n < N; ++n) acc += x[n] * std::polar(1.0, DIR * 2 * M_PI * double((long long)n * k % N) / N);
        const c32 g = regs[k % 32][k / 32];   // canonical layout again: register k2 of lane k1 = element k1 + 32 k2
        err += std::norm(acc - cd(g.x, g.y));
        nrm += std::norm(acc);
    }
    return std::sqrt(err / nrm);
}
template <int DIR>
double check_dft32()
{
    c32 a[32];
    cd x[32];
    for (int i = 0; i < 32; ++i) {
        x[i] = cd(0.3 * i - 1.0 + 0.01 * i * i, 0.7 - 0.11 * i);
        a[i] = c32{float(x[i].real()), float(x[i].imag())};
    }
    dft32<DIR>(a);
    double err = 0;
    for (int k = 0; k < 32; ++k) {
        cd acc = 0;
        for (int n = 0; n < 32; ++n) acc += x[n] * std::polar(1.0, DIR * 2 * M_PI * n * k / 32);
        err = std::fmax(err, std::abs(acc - cd(a[k].x, a[k].y)));   // natural order in and out
    }
    return err;
}

int fails = 0;
void report(const char* what, int n, int dir, double e, double tol)
{
    std::printf("%-10s N=%5d dir=%+d err=%.3e %s\n", what, n, dir, e, e <= tol ? "ok" : "FAIL");
    if (!(e <= tol)) ++fails;
}

template <int N>
void sweep()
{
    report("row", N, FWD, check<N, FWD, false>(), 1e-6);
    report("row", N, INV, check<N, INV, false>(), 1e-6);
    report("col", N, FWD, check<N, FWD, true>(), 1e-6);
    report("col", N, INV, check<N, INV, true>(), 1e-6);
    report("row/ord", N, FWD, check<N, FWD, false, true>(), 1e-6);
    report("row/ord", N, INV, check<N, INV, false, true>(), 1e-6);
}

int main()
{
    report("dft2", 2, FWD, check_dft<2, FWD>(), 1e-5);
    report("dft4", 4, FWD, check_dft<4, FWD>(), 1e-5);
    report("dft4", 4, INV, check_dft<4, INV>(), 1e-5);
    report("dft8", 8, FWD, check_dft<8, FWD>(), 1e-5);
    report("dft8", 8, INV, check_dft<8, INV>(), 1e-5);
    report("dft16", 16, FWD, check_dft<16, FWD>(), 1e-5);
    report("dft16", 16, INV, check_dft<16, INV>(), 1e-5);
    sweep<2>(); sweep<4>(); sweep<8>(); sweep<16>(); sweep<32>(); sweep<64>(); sweep<128>();
    sweep<256>(); sweep<512>(); sweep<1024>(); sweep<2048>(); sweep<4096>();
    report("dft32", 32, FWD, check_dft32<FWD>(), 1e-4);
    report("dft32", 32, INV, check_dft32<INV>(), 1e-4);
    report("32x32", 1024, FWD, check_32x32<FWD>(), 1e-6);
    report("32x32", 1024, INV, check_32x32<INV>(), 1e-6);
    std::printf(fails ? "FAILED %d\n" : "ALL OK\n", fails);
    return fails ? 1 : 0;
}
