// p3d_mix64.hpp -- the double-precision passes on the mixed-radix register engine (p3d_mix64.hip), seen from p3d_f64.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace p3d {
namespace mix {
struct c64d;
}
namespace mix64 {

// column pass of a batch: work [nslices][N][n2] complex128, row-major; modes COL_ITER / COL_STATS / COL_FWD (p3d_kernels_common.hpp)
struct ColArgs64 {
    mix::c64d* work;
    const mix::c64d* tab;      // MixPlan::build_tw<c64d> (device)
    int n2, nslices;
    const mix::c64d* tau;      // [nslices][niter]
    int niter, iter, op;
    double* partial;           // COL_STATS: [nslices][tiles][8]
    const int* done;
    unsigned char* nzflag;     // COL_ITER: [nslices][tiles] 1 = the tile kept a coefficient; tiles the threshold emptied are neither transformed back nor stored (NULL: always)
};
// row pass: modes ROW_FIRST / ROW_MID / ROW_LAST; partial [nslices][row groups] sums of |x|
struct RowArgs64 {
    mix::c64d* work;
    const mix::c64d* tab;
    int n1, nslices;
    const void* x;
    int dtype;                 // P3D_C128 / P3D_F64 / P3D_C64 / P3D_F32 (of x and out)
    const double* mask;
    void* out;
    double* partial;
    int adaptive, write_out;
    double alpha;
    const int* done;
    int zero_fill;
    const unsigned char* nzflag;   // ROW_MID / ROW_LAST: the column pass's tile flags [nslices][nz_tiles], a tile = nz_col_t columns: emptied tiles read as zeros (NULL: dense)
    int nz_tiles, nz_col_t;
};
// mode numbers (= ColMode / RowMode of p3d_kernels_common.hpp, = the C64_* / R64_* of p3d_f64.hip)
enum { M64_COL_ITER = 0, M64_COL_STATS = 1, M64_COL_FWD = 2, M64_ROW_FIRST = 0, M64_ROW_MID = 1, M64_ROW_LAST = 2 };

// ---- the fused passes of the double-precision SHEARLET loop (p3d_shearlet64.hip): coefficient buffer U [nb * nsh][N][n2] complex128, entry
// b * nsh + s = shearlet s of slice b; F [nb][n1][n2] spectra; psi [nsh][n1][n2] doubles (real spectra, FFT order) -------------------------------
// column pass of U (or of F with nsh = 1): inverse transform, x scale, real part if real_only, then
//   mode 0: threshold with tau[b][iter][s], forward transform, stored as a spectrum along the columns again;   mode 1: stored as it is (samples)
struct ShearCol64 {
    mix::c64d* U;
    const mix::c64d* tab;
    int n2, nslices, nsh;      // nslices = nb * nsh entries
    const mix::c64d* tau;      // [nb][niter][nsh]
    int niter, iter, op, real_only, mode;
    double scale;
    const int* done;           // [nb]
    const unsigned char* sup;  // [nsh][sup_groups] or NULL: 0 = the spectrum of shearlet s vanishes on rows g * sup_rows ... (+ sup_rows - 1): those rows of U are
    int sup_groups, sup_rows;  //   never written by the spread pass, read as zeros here and not stored (the gather pass skips them)
    int pair;                  // 1: REAL cubes on symmetric spectra -- the coefficients are real, so U is Hermitian along its columns: only rows 0 ... N/2 exist,
                               //   and two adjacent columns go through one complex transform (Z = Va + i Vb; n2 even, N even)
};
// rows of U[b * nsh + s] = inverse row transform (unscaled) of psi_s x F[b]
struct SpreadRow64 {
    const mix::c64d* F;
    const double* psi;
    mix::c64d* U;
    const mix::c64d* tab;
    int n1, nb, nsh;
    const int* done;
    const unsigned char* sup;  // [nsh][sup_groups] or NULL (a group = the rows of one workgroup, Entry::row_lines)
    int sup_groups;
    int rows;                  // rows 0 ... rows - 1 are worked on (n1, or n1 / 2 + 1 for Hermitian coefficient slices: ShearCol64::pair)
};
// rows of F[b] = sum_s psi_s x forward row transform of U[b * nsh + s]
struct GatherRow64 {
    const mix::c64d* U;
    const double* psi;
    mix::c64d* F;
    const mix::c64d* tab;
    int n1, nb, nsh;
    const int* done;
    const unsigned char* sup;
    int sup_groups;
    int rows;                  // as SpreadRow64::rows; the rows beyond are not written (the caller mirrors them)
};

struct Entry {
    int n, col_tile, row_lines, tw_slots;
    void (*build_tw)(mix::c64d* out);
    hipError_t (*col)(int mode, const ColArgs64& a, hipStream_t st);
    hipError_t (*row)(int mode, const RowArgs64& a, hipStream_t st);
    hipError_t (*shear_col)(const ShearCol64& a, hipStream_t st);   // (ShearCol64::pair selects the two-columns-per-transform kernel)
    hipError_t (*spread_row)(const SpreadRow64& a, hipStream_t st);
    hipError_t (*gather_row)(const GatherRow64& a, hipStream_t st);
};
const Entry* find(int n);   // nullptr: no plan (p3d_f64.hip's LDS-image passes)
const Entry* part_0();

}  // namespace mix64
}  // namespace p3d
