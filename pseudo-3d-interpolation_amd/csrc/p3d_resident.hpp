// p3d_resident.hpp -- interface of the slice-resident single-kernel POCS job (p3d_resident.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "p3d_fft.hpp"

namespace p3d {

struct ResidentArgs {
    const void* x;          // observed cube [nslices][nil][nxl], complex64 or float32
    void* out;              // result cube, same type
    const uint16_t* bits;   // packed binary trace mask [nil][nxl / 16]: bit q of entry (row, tl) = mask[row][tl + (nxl / 16) q]
    const c32* tau;         // [nslices][niter]
    int* done;              // in: 0 = run, < 0 = all-zero slice; out: iteration at which the slice converged (0: ran all)
    double* sums;           // [niter + 1][nslices] sums of |x| (zeroed by the caller)
    const c32* tw_row;      // PassTables<nxl>
    const c32* tw_col;      // ColTables<nil>
    int nslices, niter, op, dtype;
    float alpha, scale;     // scale = 1 / (nil nxl)
    double eps;
};

bool resident_supported(int nil, int nxl);   // 32, 64 or 128 points per axis
hipError_t resident_launch(int nil, int nxl, const ResidentArgs& a, hipStream_t st);

}  // namespace p3d
