// p3d_inst.hip -- compiled once per supported line length (-DP3D_N=<power of two>); exports the
// launchers of that length to the API layer (p3d_api.hip) through one LineOps record.
#include "p3d_kernels.hpp"

#ifndef P3D_N
#error "compile with -DP3D_N=<line length>"
#endif
#define P3D_CAT2(a, b) a##b
#define P3D_CAT(a, b) P3D_CAT2(a, b)

namespace p3d {
const LineOps* P3D_CAT(get_line_ops_, P3D_N)()
{
    static const LineOps ops = {P3D_N,           col_tile<P3D_N>(), Plan<P3D_N>::TPL, Plan<P3D_N>::PPT,
                                &launch_row<P3D_N>, &launch_col<P3D_N>, &launch_row_pipe<P3D_N>, row_lds_bytes<P3D_N>(),
                                PassTables<P3D_N>::slots(), &PassTables<P3D_N>::build,
                                ColTables<P3D_N>::slots(),  &ColTables<P3D_N>::build,
                                &launch_row_real<P3D_N>, &launch_row_pipe64<P3D_N>, &launch_col_pipe<P3D_N>, &launch_col_shear_pair<P3D_N>,
                                P3D_N == P32::N ? &launch_row_pipe32<P3D_N> : nullptr, P3D_N == P32::N ? P32::TW : 0, P3D_N == P32::N ? &P32::build_tw : nullptr};
    return &ops;
}
}  // namespace p3d

#if P3D_STAMPS && P3D_N == 1024
// diagnostic build only: hand the stamps of the last row_pipe64_kernel<1024> launch to the host (tools/rowpass_stamps.py)
extern "C" int p3d_debug_read_stamps(unsigned* out, int n)
{
    const size_t bytes = sizeof(unsigned) * (size_t)n;
    if (bytes > sizeof(p3d::p3d_stamp_buf)) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3d::p3d_stamp_buf), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif
