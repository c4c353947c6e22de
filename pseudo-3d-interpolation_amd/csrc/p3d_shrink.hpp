// p3d_shrink.hpp -- the threshold operators of threshold_operator.py on one coefficient (shared by the FFT, WAVELET and SHEARLET
// kernels).
#pragma once
#include <hip/hip_runtime.h>

#include "p3d_fft.hpp"

namespace p3d {

// ---- thresholding of one coefficient ---------------------------------------------------------
// tau is complex because the reference scales its schedule with numpy's lexicographic complex
// max (POCS.py:288); comparisons and clipping against it are lexicographic as well.
__device__ __forceinline__ c32 shrink(c32 X, c32 tau, int op)
{
    const float p = X.x * X.x + X.y * X.y;
    if (op == 0) {  // hard: where(|X| < tau, 0, X)          threshold_operator.py:110-112
        // |X| < Re tau  <=>  |X|^2 < (Re tau)^2 for Re tau > 0 (a negative Re tau keeps everything): no square root, whose
        // IEEE fix-up sequence is a dozen instructions per coefficient of a pass that is bound by instruction issue.  Both forms
        // place a coefficient within an ulp of the threshold arbitrarily; the tests treat that band as ties.
        // Lexicographic "<": |X| == Re tau counts as below when Im tau > 0, i.e. p <= t2, i.e. p < next_float(t2).  The limit
        // depends on tau alone (a handful of instructions per thread, not per coefficient).
        const float t2 = tau.x * tau.x;
        float lim = 0.0f;                                          // Re tau < 0: nothing is below
        if (tau.x >= 0.0f) lim = (tau.y > 0.0f && t2 < __builtin_inff()) ? __uint_as_float(__float_as_uint(t2) + 1u) : t2;
        return p < lim ? c32{0.f, 0.f} : X;
    }
    const float m = sqrtf(p);
    if (m == 0.0f) return c32{0.f, 0.f};  // 1 - tau/0 = -inf -> clipped to 0
    float gr, gi;
    if (op == 1) {  // soft: X * clip(1 - tau/|X|, 0)           threshold_operator.py:36-39
        const float r = 1.0f / m;
        gr = 1.0f - tau.x * r;
        gi = -tau.y * r;
    } else {        // garrote: X * clip(1 - tau^2/|X|^2, 0)    threshold_operator.py:75-78
        const float r = 1.0f / (m * m);
        gr = 1.0f - (tau.x * tau.x - tau.y * tau.y) * r;
        gi = -(2.0f * tau.x * tau.y) * r;
    }
    const bool keep = (gr > 0.0f) || (gr == 0.0f && gi >= 0.0f);  // lexicographic max(g, 0)
    return keep ? X * c32{gr, gi} : c32{0.f, 0.f};
}

// real data with a real threshold (float32 cubes through the WAVELET / SHEARLET transforms): the same operators on |x|
__device__ __forceinline__ float shrink(float x, c32 tau, int op)
{
    const float m = fabsf(x);
    if (op == 0) return m < tau.x ? 0.f : x;
    if (m == 0.0f) return 0.f;
    const float g = op == 1 ? 1.0f - tau.x / m : 1.0f - (tau.x * tau.x) / (m * m);
    return g > 0.0f ? x * g : 0.f;
}

}  // namespace p3d
