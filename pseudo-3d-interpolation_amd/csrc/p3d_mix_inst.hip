// p3d_mix_inst.hip -- instantiations of the mixed-radix register engine (p3d_mix.hpp), one translation unit per P3D_MIX_PART (the plan list
// p3d_mix_plans.inc deals the lengths round; P3D_MIX_PART is set by the Makefile).
#include <hip/hip_runtime.h>

#include "p3d_mix.hpp"

#ifndef P3D_MIX_PART
#error "compile with -DP3D_MIX_PART=<k>"
#endif

namespace p3d {
namespace mix {

namespace {

constexpr size_t LDS_LIMIT = 160 * 1024;

template <class PL>
hipError_t launch_col(int mode, const ColArgs& a, const c32* tab, hipStream_t st)
{
    constexpr size_t lds = sizeof(c32) * ((size_t)PL::TW_SLOTS + (size_t)PL::LINE * PL::COLT);
    static_assert(lds + 1024 <= LDS_LIMIT, "column tile does not fit LDS");
    static_assert(PL::COLT * PL::TMAX <= 1024, "column tile needs more than 1024 threads");
    if (mode == COL_SHRINK) return hipErrorNotSupported;
    static bool attr = false;   // (idempotent: a race sets it twice)
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mix_col_kernel<PL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    // tiles narrower than a 64-byte column block: the grid is rounded up so that the kernel can deal the 8 / COLT tiles of one block to
    // workgroups that share an XCD (see mix_col_kernel); the surplus workgroups leave at once
    constexpr int G = 8 / PL::COLT;
    int tiles = (a.n2 + PL::COLT - 1) / PL::COLT;
    const int ntiles = tiles;
    if (G > 1) tiles = (tiles + 8 * G - 1) / (8 * G) * (8 * G);
    const dim3 grid(tiles, a.nslices);
    mix_col_kernel<PL><<<grid, PL::COLT * PL::TMAX, lds, st>>>(a, tab, mode, ntiles);
    return hipGetLastError();
}

template <class PL>
hipError_t launch_row(int mode, const RowArgs& a, const c32* tab, hipStream_t st)
{
    constexpr size_t lds = sizeof(c32) * ((size_t)PL::TW_SLOTS + (size_t)PL::LINE * PL::ROWLB);
    static_assert(lds + 1024 <= LDS_LIMIT, "row group does not fit LDS");
    static_assert(PL::ROWLB * PL::TMAX <= 1024, "row group needs more than 1024 threads");
    if (mode != ROW_FIRST && mode != ROW_MID && mode != ROW_LAST) return hipErrorNotSupported;
    static bool attr = false;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mix_row_kernel<PL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const dim3 grid((a.n1 + PL::ROWLB - 1) / PL::ROWLB, a.nslices);
    mix_row_kernel<PL><<<grid, PL::ROWLB * PL::TMAX, lds, st>>>(a, tab, mode);
    return hipGetLastError();
}

// plans with a radix-11 / 13 pass serve columns only (MixPlan::BIG_PRIME): no row kernel is instantiated for them
template <class PL>
constexpr auto row_launcher() -> hipError_t (*)(int, const RowArgs&, const c32*, hipStream_t)
{
    if constexpr (PL::BIG_PRIME) return nullptr;
    else return &launch_row<PL>;
}

#define P3D_MIX_PLAN(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3) MixPlan<N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3>
#define X(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)                                                                              \
    {N, COLT, P3D_MIX_PLAN(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)::TW_SLOTS, P3D_MIX_PLAN(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)::TPL_A, \
     P3D_MIX_PLAN(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)::PPT_A, &P3D_MIX_PLAN(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)::template build_tw<c32>,         \
     row_launcher<P3D_MIX_PLAN(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)>(), &launch_col<P3D_MIX_PLAN(N, COLT, LB, NP, R0, B0, R1, B1, R2, B2, R3, B3)>},
const Entry entries[] = {
#include "p3d_mix_plans.inc"
    {0, 0, 0, 0, 0, nullptr, nullptr, nullptr}};
#undef X

}  // namespace

#define P3D_MIX_CAT2(a, b) a##b
#define P3D_MIX_CAT(a, b) P3D_MIX_CAT2(a, b)
const Entry* P3D_MIX_CAT(part_, P3D_MIX_PART)() { return entries; }

}  // namespace mix
}  // namespace p3d
