// p3d_select.hpp -- lexicographic order statistics of complex spectra (p3d_select.hip), for the 'data-driven' threshold model
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "p3d_fft.hpp"

namespace p3d {

// spectrum [nslices][per] complex64 is REPLACED by its 64-bit order keys; `sorted` [nslices][per] x 8 bytes receives them in
// descending order per slice; peaks_dev [nslices][2] the lexicographic maximum of every slice.  Synchronises the stream.
hipError_t lex_sort_desc(c32* spectrum, void* sorted, size_t per, int nslices, float* peaks_dev, hipStream_t st);

// bounds_dev [nslices][4] = tau_min (re, im), tau_max (re, im) -> count_dev [nslices] = Nv = #{tau_min < X < tau_max},
// tau_dev [nslices][niter][2] = the picks of POCS.py:359-362 (untouched where Nv = 0)
hipError_t data_driven_pick(const void* sorted, size_t per, int nslices, int niter, const float* bounds_dev, float* tau_dev, long long* count_dev,
                            hipStream_t st);

}  // namespace p3d
