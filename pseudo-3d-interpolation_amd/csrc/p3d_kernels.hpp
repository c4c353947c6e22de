// p3d_kernels.hpp -- the two fused passes of one POCS iteration on a batch of slices.
//
// One iteration of the reference loop (pseudo_3D_interpolation/functions/POCS.py:560-632)
//     X = fft2(x_old); X = threshold(X, tau_k); x = ifft2(X); x *= 1 - alpha*mask; x += alpha*x_obs
// is regrouped so that every slice is read and written exactly twice per iteration:
//
//   spectrum (column) pass  col_kernel<N1,T,COL_ITER>:
//        load T columns of the row-transformed slice -> forward column FFT
//        -> threshold (threshold_operator.py:9-112)  -> inverse column FFT -> store
//   space (row) pass        row_kernel<N2,ROW_MID>:
//        load rows -> inverse row FFT, 1/(N1*N2) -> re-insertion of the observed traces
//        (POCS.py:616-619) -> sum|x| for the cost (POCS.py:622) -> [APOCS input mix, POCS.py:574-575]
//        -> forward row FFT of the NEXT iteration -> store
//
// ROW_FIRST starts the chain (x_obs -> forward row FFT), ROW_LAST ends it (stores x instead of
// transforming again).
//
// Work-buffer layout ("column blocked"): the intermediate between the two passes is private to this
// library, so it is stored as  W[slice][cb = col/8][row][col%8]  (complex64).  One column block
// (8 columns = 64 bytes per row) of a slice is then ONE contiguous N1*64-byte run: the column pass
// streams it with perfectly linear addresses and needs only N1*64 B of LDS per tile (2 workgroups
// per CU at N1 = 1024, so loads of one tile overlap the transforms of another), while the row pass,
// whose workgroups own adjacent rows, still touches 64-byte pieces that are neighbours in memory.
// Measured on MI355X (tools/micro/membench.hip): 4.9-5.2 TB/s for the blocked column tiles vs 4.7
// (16-column tiles, 1 WG/CU) and 3.0 TB/s (8-column tiles) on the row-major layout.
#pragma once

#include "p3d_kernels_common.hpp"
#include "p3d_row_kernels.hpp"
#include "p3d_row_pipe64.hpp"
#include "p3d_row_pipe32.hpp"
#include "p3d_row_real.hpp"
#include "p3d_col_kernels.hpp"
#include "p3d_col_shear.hpp"

// The kernels live in one header per pass (p3d_row_kernels.hpp: one-launch and per-lane persistent row pass; p3d_row_pipe64.hpp: the
// wave-uniform persistent row pass; p3d_row_real.hpp: float32 cubes; p3d_col_kernels.hpp: the column pass); this file keeps the launch
// helpers and the per-length dispatch table (LineOps).

namespace p3d {

// ---- launch helpers, one instantiation set per line length --------------------------------------
// columns per workgroup of the column pass: keep 512..1024 threads and <= ~80 KiB of LDS
template <int N>
constexpr int col_tile()
{
    return N >= 4096 ? 2 : (N >= 1024 ? 8 : (N == 512 ? 16 : (N == 256 ? 32 : 64)));  // N = 2048: 4-column tiles (2 WG/CU) measured slower (3.4 vs 2.6 ms)
}

template <int N>
constexpr size_t row_lds_bytes()
{
    return sizeof(c32) * (PassTables<N>::slots() + (ROW_THREADS / Plan<N>::TPL) * LdsRow::stride(N)) + 8 * sizeof(double);
}
template <int N, int MODE>
constexpr size_t row_lds_bytes_mode()
{
    return sizeof(c32) * (PassTables<N>::slots() + (row_threads<N, MODE>() / Plan<N>::TPL) * LdsRow::stride(N)) + 16 * sizeof(double);
}
template <int N>
constexpr size_t col_lds_bytes()
{
    constexpr int T = col_tile<N>();
    constexpr int CW = T < 8 ? T : 8;
    return sizeof(c32) * (ColTables<N>::slots() + (size_t)(T / CW) * LdsColW<CW>::stride(N));
}

template <class K>
inline hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int N, int MODE, bool BITS>
hipError_t launch_row_one(const RowArgs& a, hipStream_t st)
{
    constexpr int LB = row_threads<N, MODE>() / Plan<N>::TPL;
    unsigned gx = (unsigned)((a.n1 + LB - 1) / LB);
    if constexpr (MODE == ROW_SPREAD_INV || MODE == ROW_GATHER_FWD) {   // rows 0 ... n1/2 only for Hermitian work slices; whole groups of 8 workgroups (XCD placement)
        gx = (unsigned)((shear_rows(a.sh, a.n1) + LB - 1) / LB);
        if (a.sh.half) gx = (gx + 7u) & ~7u;
    }
    const dim3 grid(gx, a.nslices);
    constexpr size_t lds = row_lds_bytes_mode<N, MODE>();
    hipError_t e = allow_lds(row_kernel<N, MODE, BITS>, lds);
    if (e != hipSuccess) return e;
    row_kernel<N, MODE, BITS><<<grid, row_threads<N, MODE>(), lds, st>>>(a);
    return hipGetLastError();
}

// one launch of the wave-uniform persistent row pass (row_pipe64_kernel) in mode `pm` on a device with `cus` compute units;
// hipErrorNotSupported where the kernel does not apply (the caller falls back to the generic kernels)
template <int N>
hipError_t launch_row_pipe64(int pm, const RowArgs& a, int cus, hipStream_t st)
{
    if constexpr (Plan<N>::TPL >= 8 && Plan<N>::TPL <= 256 && Plan<N>::PPT == 16) {
        constexpr int RPW64 = Plan<N>::TPL >= 64 ? 1 : 64 / Plan<N>::TPL;
        if ((double)a.nslices * (double)wk_slice_stride(a.n1, N) >= 4294967296.0 || a.n1 % RPW64 != 0 || cus < 1) return hipErrorNotSupported;
        if (a.only_done || a.plain) return hipErrorNotSupported;                  // finalize, fft2 hook
        if (a.write_out && !(a.adaptive && pm == PIPE_MID)) return hipErrorNotSupported;   // the per-iteration store exists for APOCS' steady state only
        if (a.adaptive && pm == PIPE_FIRST) return hipErrorNotSupported;          // APOCS: the input mix of the first pass stays with row_kernel (once per job)
        const bool tables = a.bits64 != nullptr && a.cbase != nullptr;
        if (pm == PIPE_FIRST) {
            if (a.x == nullptr || (a.xc != nullptr && !tables)) return hipErrorNotSupported;   // (no tables: the transform only)
        } else {
            if (a.bits == nullptr || a.xc == nullptr || !tables) return hipErrorNotSupported;
        }
        constexpr int LB64 = pipe64_rows<N>();
        constexpr size_t lds64 = pipe64_lds_bytes<N>();
        constexpr int UPB64 = LB64 / RPW64;
        const long groups64 = ((long)a.nslices * (a.n1 / RPW64) + UPB64 - 1) / UPB64;
        int per_cu64 = (int)((160 * 1024) / lds64);            // workgroups a CU holds: LDS ...
        const int by_waves = 16 / (pipe64_threads<N>() / 64);  // ... and 4 waves per SIMD
        if (per_cu64 > by_waves) per_cu64 = by_waves;
        if (per_cu64 < 1) per_cu64 = 1;
        const long wgs64 = (long)cus * per_cu64;
        const dim3 grid64((unsigned)(groups64 < wgs64 ? groups64 : wgs64));
        hipError_t e = hipSuccess;
#define P3D_PIPE64(DT, SP, PM)                                                                                  \
    do {                                                                                                        \
        if ((e = allow_lds(row_pipe64_kernel<N, DT, SP, PM>, lds64)) != hipSuccess) return e;                   \
        row_pipe64_kernel<N, DT, SP, PM><<<grid64, pipe64_threads<N>(), lds64, st>>>(a);                        \
    } while (0)
        const bool sp = a.nzl != nullptr;
        if (pm == PIPE_FIRST) {
            if (a.dtype == 0) P3D_PIPE64(0, false, PIPE_FIRST); else P3D_PIPE64(1, false, PIPE_FIRST);
        } else if (pm == PIPE_LAST) {
            if (a.dtype == 0) { if (sp) P3D_PIPE64(0, true, PIPE_LAST); else P3D_PIPE64(0, false, PIPE_LAST); }
            else { if (sp) P3D_PIPE64(1, true, PIPE_LAST); else P3D_PIPE64(1, false, PIPE_LAST); }
        } else if (pm == PIPE_MID && a.adaptive) {   // APOCS (plain stores: the transposed store's two barriers bought nothing measurable there)
#define P3D_PIPE64_AD(DT, SP)                                                                                          \
    do {                                                                                                               \
        if ((e = allow_lds(row_pipe64_kernel<N, DT, SP, PIPE_MID, false, true>, lds64)) != hipSuccess) return e;       \
        row_pipe64_kernel<N, DT, SP, PIPE_MID, false, true><<<grid64, pipe64_threads<N>(), lds64, st>>>(a);            \
    } while (0)
            if (a.dtype == 0) { if (sp) P3D_PIPE64_AD(0, true); else P3D_PIPE64_AD(0, false); }
            else { if (sp) P3D_PIPE64_AD(1, true); else P3D_PIPE64_AD(1, false); }
#undef P3D_PIPE64_AD
        } else if (pm == PIPE_MID) {
            bool ts = false;
            if constexpr (pipe64_can_tstore<N>() && LB64 == 16) {
                ts = a.tstore && a.n1 % 16 == 0;
                if (ts) {
#define P3D_PIPE64_TS(DT, SP)                                                                                   \
    do {                                                                                                        \
        if ((e = allow_lds(row_pipe64_kernel<N, DT, SP, PIPE_MID, true>, lds64)) != hipSuccess) return e;       \
        row_pipe64_kernel<N, DT, SP, PIPE_MID, true><<<grid64, pipe64_threads<N>(), lds64, st>>>(a);            \
    } while (0)
                    if (a.dtype == 0) { if (sp) P3D_PIPE64_TS(0, true); else P3D_PIPE64_TS(0, false); }
                    else { if (sp) P3D_PIPE64_TS(1, true); else P3D_PIPE64_TS(1, false); }
#undef P3D_PIPE64_TS
                }
            }
            if (!ts) {
                if (a.dtype == 0) { if (sp) P3D_PIPE64(0, true, PIPE_MID); else P3D_PIPE64(0, false, PIPE_MID); }
                else { if (sp) P3D_PIPE64(1, true, PIPE_MID); else P3D_PIPE64(1, false, PIPE_MID); }
            }
        } else {
            return hipErrorInvalidValue;
        }
#undef P3D_PIPE64
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

// one launch of the one-exchange persistent row pass of 1024-sample rows (row_pipe32_kernel) in mode `pm`; hipErrorNotSupported where
// the kernel does not apply.  What it needs is decided per PLAN (p3d_api.hip: use32) except the tables a mode reads, which the caller
// checks before it commits a job to this kernel family (the compact samples are ordered differently from the other kernels').
template <int N>
hipError_t launch_row_pipe32(int pm, const RowArgs& a, int cus, hipStream_t st)
{
    if constexpr (N == P32::N) {
        if ((double)a.nslices * (double)wk_slice_stride(a.n1, N) >= 4294967296.0 || a.n1 % 2 != 0 || cus < 1 || a.tw32 == nullptr) return hipErrorNotSupported;
        if (a.only_done || a.plain) return hipErrorNotSupported;                  // finalize, fft2 hook
        if (a.write_out && !a.adaptive) return hipErrorNotSupported;   // the per-iteration store exists for APOCS only (the last pass stores anyway)
        const bool tables = a.bits32 != nullptr && a.rowbase != nullptr;
        if (pm == PIPE_FIRST) {
            if (a.x == nullptr || (a.xc != nullptr && !tables) || (a.adaptive && !tables)) return hipErrorNotSupported;
        } else {
            if (a.xc == nullptr || !tables) return hipErrorNotSupported;
        }
        const long groups = ((long)a.nslices * (a.n1 / 2) + P32::UPB - 1) / P32::UPB;
        const dim3 grid((unsigned)(groups < cus ? groups : cus));   // one 147-KiB workgroup per CU
        constexpr size_t lds = P32::lds_bytes();
        hipError_t e = hipSuccess;
#define P3D_PIPE32(DT, SP, PM, AD)                                                                              \
    do {                                                                                                        \
        if ((e = allow_lds(row_pipe32_kernel<DT, SP, PM, AD>, lds)) != hipSuccess) return e;                    \
        row_pipe32_kernel<DT, SP, PM, AD><<<grid, P32::THREADS, lds, st>>>(a);                                  \
    } while (0)
        const bool sp = a.nzl != nullptr, c64 = a.dtype == 0;
        if (pm == PIPE_FIRST) {
            if (a.adaptive) { if (c64) P3D_PIPE32(0, false, PIPE_FIRST, true); else P3D_PIPE32(1, false, PIPE_FIRST, true); }
            else { if (c64) P3D_PIPE32(0, false, PIPE_FIRST, false); else P3D_PIPE32(1, false, PIPE_FIRST, false); }
        } else if (pm == PIPE_LAST) {
            if (c64) { if (sp) P3D_PIPE32(0, true, PIPE_LAST, false); else P3D_PIPE32(0, false, PIPE_LAST, false); }
            else { if (sp) P3D_PIPE32(1, true, PIPE_LAST, false); else P3D_PIPE32(1, false, PIPE_LAST, false); }
        } else if (pm == PIPE_MID && a.adaptive) {
            if (c64) { if (sp) P3D_PIPE32(0, true, PIPE_MID, true); else P3D_PIPE32(0, false, PIPE_MID, true); }
            else { if (sp) P3D_PIPE32(1, true, PIPE_MID, true); else P3D_PIPE32(1, false, PIPE_MID, true); }
        } else if (pm == PIPE_MID) {
            if (c64) { if (sp) P3D_PIPE32(0, true, PIPE_MID, false); else P3D_PIPE32(0, false, PIPE_MID, false); }
            else { if (sp) P3D_PIPE32(1, true, PIPE_MID, false); else P3D_PIPE32(1, false, PIPE_MID, false); }
        } else {
            return hipErrorInvalidValue;
        }
#undef P3D_PIPE32
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

// persistent steady-state row pass on a device with `cus` compute units
template <int N>
hipError_t launch_row_pipe(const RowArgs& a, int cus, hipStream_t st)
{
    if ((double)a.nslices * (double)wk_slice_stride(a.n1, N) >= 4294967296.0) return hipErrorNotSupported;  // 32-bit offsets
    const bool bits = a.bits != nullptr;
    const bool extra = a.adaptive || a.write_out || a.done != nullptr;
    const bool compact = bits && a.xc != nullptr;
    hipError_t e = launch_row_pipe64<N>(PIPE_MID, a, cus, st);   // the wave-uniform variant (it also honours the per-slice `done` flags)
    if (e != hipErrorNotSupported) return e;
    e = hipSuccess;
    if constexpr (Plan<N>::TPL > 64) {
        return hipErrorNotSupported;
    } else {
        constexpr int LB = ROW_THREADS / Plan<N>::TPL;
        // resident workgroups per CU: LDS (160 KiB per CU) and the register budget of the variant
        const int by_lds = (int)((160 * 1024) / row_lds_bytes<N>());
        const int wpe = (compact && P3D_COMPACT_LATE) ? P3D_PIPE_WAVES_PER_EU_COMPACT : P3D_PIPE_WAVES_PER_EU;
        const int by_regs = (wpe * 4 * 64) / ROW_THREADS;
        const int per_cu = by_lds < by_regs ? by_lds : by_regs;
        if (per_cu < 1) return hipErrorNotSupported;
        const long wgs = (long)cus * per_cu;
        const long groups = ((long)a.nslices * a.n1 + LB - 1) / LB;
        const dim3 grid((unsigned)(groups < wgs ? groups : wgs));
        constexpr size_t lds = row_lds_bytes<N>();
#define P3D_PIPE(BITS, DT, EXTRA, COMPACT)                                                                  \
    do {                                                                                                    \
        if ((e = allow_lds(row_pipe_kernel<N, BITS, DT, EXTRA, COMPACT>, lds)) != hipSuccess) return e;     \
        row_pipe_kernel<N, BITS, DT, EXTRA, COMPACT><<<grid, ROW_THREADS, lds, st>>>(a);                    \
    } while (0)
        if (a.dtype == 0) {
            if (compact) { if (extra) P3D_PIPE(true, 0, true, true); else P3D_PIPE(true, 0, false, true); }
            else if (bits) { if (extra) P3D_PIPE(true, 0, true, false); else P3D_PIPE(true, 0, false, false); }
            else { if (extra) P3D_PIPE(false, 0, true, false); else P3D_PIPE(false, 0, false, false); }
        } else {
            if (compact) { if (extra) P3D_PIPE(true, 1, true, true); else P3D_PIPE(true, 1, false, true); }
            else if (bits) { if (extra) P3D_PIPE(true, 1, true, false); else P3D_PIPE(true, 1, false, false); }
            else { if (extra) P3D_PIPE(false, 1, true, false); else P3D_PIPE(false, 1, false, false); }
        }
#undef P3D_PIPE
        return hipGetLastError();
    }
}

// real (float32) cubes, rows of 128 ... 1024 samples: the row-pair passes over the half-spectrum work buffer
template <int N>
hipError_t launch_row_real(int mode, const RowArgs& a, int cus, hipStream_t st)
{
    if constexpr ((64 % Plan<N>::TPL == 0 || Plan<N>::TPL % 64 == 0) && Plan<N>::TPL >= 8 && Plan<N>::TPL <= 256 && Plan<N>::PPT == 16) {
        constexpr int PW = Plan<N>::TPL >= 64 ? 1 : 64 / Plan<N>::TPL, WPL = Plan<N>::TPL >= 64 ? Plan<N>::TPL / 64 : 1;
        // Rows of 2048 samples (two wavefronts per pair: workgroup barriers around the split and the sums on top of those of the transforms, 8 waves
        // per CU) used to lose what the half spectrum gains (round 2: 71.3 ms against 68.4 on the complex path for 1024 x 2048 x 256 float32, 20
        // iterations) -- at a register budget meant for 16 waves per CU, with 27 registers spilled.  With the budget its LDS-bound occupancy allows
        // (round 3) the pair form wins there too: 16.8 against 17.7 ms on 1024 x 2048 x 64; 4096 samples: 17.7 against 19.5.  P3D_NO_REAL_2048=1: off.
        if (WPL == 2 && !a.real_2048) return hipErrorNotSupported;
        if (a.n1 % (2 * PW) != 0 || a.bits64 == nullptr || a.cbase == nullptr || a.dtype != 1) return hipErrorNotSupported;
        if ((double)a.nslices * (double)wk_slice_stride(a.n1, N / 2 + 1) >= 4294967296.0) return hipErrorNotSupported;
        constexpr size_t lds = pipe64_lds_bytes<N>();
        constexpr int UPB = pipe64_threads<N>() / 64 / WPL;
        const long groups = ((long)a.nslices * (a.n1 / (2 * PW)) + UPB - 1) / UPB;
        int per_cu = (int)((160 * 1024) / lds);               // workgroups a CU holds: LDS and 4 waves per SIMD
        if (per_cu > 16 / (pipe64_threads<N>() / 64)) per_cu = 16 / (pipe64_threads<N>() / 64);
        if (per_cu < 1) per_cu = 1;
        const long wgs = (long)cus * per_cu;
        const dim3 grid((unsigned)(groups < wgs ? groups : wgs));
        hipError_t e = hipSuccess;
#define P3D_REAL(MODE, SP)                                                                      \
    do {                                                                                        \
        if ((e = allow_lds(row_real_kernel<N, MODE, SP>, lds)) != hipSuccess) return e;         \
        row_real_kernel<N, MODE, SP><<<grid, pipe64_threads<N>(), lds, st>>>(a);                \
    } while (0)
        const bool sp = a.nzl != nullptr;
        switch (mode) {
            case REAL_FIRST: P3D_REAL(REAL_FIRST, false); break;
            case REAL_MID: if (sp) P3D_REAL(REAL_MID, true); else P3D_REAL(REAL_MID, false); break;
            case REAL_LAST: if (sp) P3D_REAL(REAL_LAST, true); else P3D_REAL(REAL_LAST, false); break;
            default: return hipErrorInvalidValue;
        }
#undef P3D_REAL
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int N>
hipError_t launch_row(int mode, const RowArgs& a, hipStream_t st)
{
    const bool bits = a.bits != nullptr;
    switch (mode) {
        case ROW_FIRST: return bits ? launch_row_one<N, ROW_FIRST, true>(a, st) : launch_row_one<N, ROW_FIRST, false>(a, st);
        case ROW_MID: return bits ? launch_row_one<N, ROW_MID, true>(a, st) : launch_row_one<N, ROW_MID, false>(a, st);
        case ROW_LAST: return bits ? launch_row_one<N, ROW_LAST, true>(a, st) : launch_row_one<N, ROW_LAST, false>(a, st);
        case ROW_SPREAD_INV: return launch_row_one<N, ROW_SPREAD_INV, false>(a, st);
        case ROW_GATHER_FWD: return launch_row_one<N, ROW_GATHER_FWD, false>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int N, int MODE>
hipError_t launch_col_one(const ColArgs& a, hipStream_t st)
{
    constexpr int T = col_tile<N>();
    constexpr int THREADS = T * Plan<N>::TPL;
    const dim3 grid((a.n2 + T - 1) / T, a.nslices);
    constexpr size_t lds = col_lds_bytes<N>();
    hipError_t e = allow_lds(col_kernel<N, T, MODE>, lds);
    if (e != hipSuccess) return e;
    col_kernel<N, T, MODE><<<grid, THREADS, lds, st>>>(a);
    return hipGetLastError();
}

// persistent column pass of the iteration (column-blocked buffer in and out), `wgs` resident workgroups
template <int N>
hipError_t launch_col_pipe(const ColArgs& a, int cus, hipStream_t st)
{
    constexpr int T = col_tile<N>();
    if constexpr (T % 8 == 0 && Plan<N>::PPT == 16) {
        if (a.in_std || a.out_std || a.in != a.out || cus < 1) return hipErrorNotSupported;
        if ((double)wk_slice_stride(N, a.n2) * 8.0 >= 2147483648.0) return hipErrorNotSupported;
        constexpr size_t lds = col_lds_bytes<N>();
        int per_cu = (int)((160 * 1024) / lds);
        const int by_waves = (P3D_COLPIPE_WAVES_PER_EU * 4) / (T * Plan<N>::TPL / 64);
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        const long total = (long)a.nslices * ((a.n2 + T - 1) / T);
        const long wgs = (long)cus * per_cu;
        const dim3 grid((unsigned)(total < wgs ? total : wgs));
        hipError_t e = hipSuccess;
#define P3D_COLPIPE(OP)                                                                       \
    do {                                                                                      \
        if ((e = allow_lds(col_pipe_kernel<N, T, OP>, lds)) != hipSuccess) return e;          \
        col_pipe_kernel<N, T, OP><<<grid, T * Plan<N>::TPL, lds, st>>>(a);                    \
    } while (0)
        if (a.sh.tau != nullptr) {   // the column pass of a SHEARLET iteration
            if ((e = allow_lds(col_pipe_kernel<N, T, 0, true>, lds)) != hipSuccess) return e;
            col_pipe_kernel<N, T, 0, true><<<grid, T * Plan<N>::TPL, lds, st>>>(a);
        } else if (a.op == 1) P3D_COLPIPE(1); else if (a.op == 2) P3D_COLPIPE(2); else P3D_COLPIPE(0);
#undef P3D_COLPIPE
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int N>
hipError_t launch_col(int mode, const ColArgs& a, hipStream_t st)
{
    switch (mode) {
        case COL_ITER:
            if (a.op == 1) return launch_col_one<N, COL_ITER_SOFT>(a, st);
            if (a.op == 2) return launch_col_one<N, COL_ITER_GARROTE>(a, st);
            return launch_col_one<N, COL_ITER>(a, st);
        case COL_STATS: return launch_col_one<N, COL_STATS>(a, st);
        case COL_FWD: return launch_col_one<N, COL_FWD>(a, st);
        case COL_INV: return launch_col_one<N, COL_INV>(a, st);
        case COL_SHRINK: return launch_col_one<N, COL_SHRINK>(a, st);
        default: return hipErrorInvalidValue;
    }
}

// what the API layer sees of one line length
struct LineOps {
    int n;
    int col_tile;
    int tpl;  // threads per line (layout of the packed mask words)
    int ppt;
    hipError_t (*row)(int mode, const RowArgs&, hipStream_t);
    hipError_t (*col)(int mode, const ColArgs&, hipStream_t);
    hipError_t (*row_pipe)(const RowArgs&, int cus, hipStream_t);  // hipErrorNotSupported when a line spans waves
    size_t row_lds;
    int row_tw_slots;                    // length of the row pass's twiddle tables ...
    void (*build_row_tw)(c32* out);      // ... and their builder
    int col_tw_slots;                    // the same for the column pass (ColTables)
    void (*build_col_tw)(c32* out);
    hipError_t (*row_real)(int mode, const RowArgs&, int cus, hipStream_t);   // REAL_* passes (hipErrorNotSupported where absent)
    hipError_t (*row_pipe64)(int pm, const RowArgs&, int cus, hipStream_t);   // PIPE_FIRST / PIPE_MID / PIPE_LAST (hipErrorNotSupported where absent)
    hipError_t (*col_pipe)(const ColArgs&, int cus, hipStream_t);             // persistent COL_ITER (hipErrorNotSupported where absent)
    hipError_t (*col_shear_pair)(const ColArgs&, hipStream_t);                // SHEARLET column pass of float32 cubes, two columns per transform
    hipError_t (*row_pipe32)(int pm, const RowArgs&, int cus, hipStream_t);   // rows of 1024 samples: the one-exchange persistent passes (nullptr elsewhere)
    int row_tw32_slots;                                                       // ... their twiddle table
    void (*build_row_tw32)(c32* out);
};

}  // namespace p3d
