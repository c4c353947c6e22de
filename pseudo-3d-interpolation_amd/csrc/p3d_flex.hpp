// p3d_flex.hpp -- the POCS passes for line lengths that are not powers of two (p3d_flex.hip), seen from the API layer.
#pragma once
#include <vector>

#include "p3d_generic.hpp"

namespace p3d {

struct LineOps;
bool flex_supported(int n);        // the lines of length n fit LDS in both passes
int flex_col_tile(int n);          // columns per workgroup of the column pass for lines of length n
void flex_build_table(int n, std::vector<c32>& out);   // twiddles (and chirp tables) of a line length, for the device
const LineOps* get_flex_ops();     // one record for every length: the launchers read the length from RowArgs::len / ColArgs::len

}  // namespace p3d
