// p3d_chirp.hpp -- the POCS passes for line lengths with a large prime factor (p3d_chirp.hip), seen from p3d_flex.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "p3d_kernels_common.hpp"

namespace p3d {

// device tables of one chirp-z line length n on M = the power of two >= 2n - 1 (all built by flex_build_table)
struct ChirpTabs {
    const c32* rowtab;   // PassTables<M>: the ordered per-pass twiddles of the row pass
    const c32* coltab;   // ColTables<M>: the column pass's tables
    const c32* chirp;    // c_k = exp(-i pi k^2 / n), k < n
    const c32* bhat;     // FFT_M of conj c, wrapped
    int n, m;
};

bool chirp_supported(int m);            // m: a power of two the register-resident engine is instantiated for here (64 ... 4096)
int chirp_col_tile(int m);              // columns per workgroup of the column pass
size_t chirp_table_slots(int m);        // entries of rowtab + coltab
void chirp_build_tables(int m, c32* rowtab_then_coltab);
size_t chirp_rowtab_slots(int m);

// mode: RowMode (ROW_FIRST / ROW_MID / ROW_LAST) and ColMode (COL_ITER* / COL_STATS / COL_FWD / COL_INV) of p3d_kernels_common.hpp
hipError_t chirp_row(int mode, const RowArgs& a, const ChirpTabs& t, hipStream_t st);
hipError_t chirp_row_real(int mode, const RowArgs& a, const ChirpTabs& t, hipStream_t st);   // float32 cubes, hard operator: row pairs
hipError_t chirp_col(int mode, const ColArgs& a, const ChirpTabs& t, hipStream_t st);

}  // namespace p3d
