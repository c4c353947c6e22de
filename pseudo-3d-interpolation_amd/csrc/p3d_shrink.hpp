// p3d_shrink.hpp -- the threshold operators of threshold_operator.py on one coefficient (shared by the FFT, WAVELET and SHEARLET
// kernels).
#pragma once
#include <hip/hip_runtime.h>

#include "p3d_fft.hpp"

namespace p3d {

// ---- thresholding of one coefficient ---------------------------------------------------------
// tau is complex because the reference scales its schedule with numpy's lexicographic complex
// max (POCS.py:288); comparisons and clipping against it are lexicographic as well.
// The hard operator without a square root per coefficient (whose IEEE fix-up sequence is a dozen instructions in passes that
// are bound by instruction issue): sqrtf is monotone, so "sqrtf(p) < Re tau" holds exactly for the p below ONE limit, and that
// limit is a function of tau alone.  sqrtf(p) = RN(sqrt(p)) < t  <=>  sqrt(p) < the midpoint of t and its lower neighbour
// <=>  p < midpoint^2, which is exact in double (25 x 2 bits) and is then rounded UP to a float.  With the lexicographic tie
// rule (|X| == Re tau counts as below when Im tau > 0) the upper neighbour takes the place of the lower one.  This keeps the
// decisions of "sqrtf(p) < tau" bit for bit -- in particular a coefficient whose own modulus IS the threshold (the first
// threshold of the inverse-proportional model, percentile thresholds) is kept, as in the reference.  A few instructions per
// thread (hoisted out of the loop over its coefficients), not per coefficient.
__device__ __forceinline__ float hard_limit(c32 tau)
{
    const bool none = !(tau.x >= 0.0f);                      // negative real part: nothing is below
    const float t = none ? 0.0f : tau.x;
    const unsigned b = __float_as_uint(t);
    const float up = b >= 0x7f800000u ? t : __uint_as_float(b + 1u);
    const float dn = b == 0u ? 0.0f : __uint_as_float(b - 1u);
    const float nb = tau.y > 0.0f ? up : dn;
    const double mid = 0.5 * ((double)t + (double)nb);
    const double m2 = mid * mid;
    const float lo = (float)m2;
    const float lim = (double)lo < m2 ? __uint_as_float(__float_as_uint(lo) + 1u) : lo;   // round up
    return none ? 0.0f : lim;
}

// One threshold applied to many coefficients: what depends on tau alone is computed once.
struct Shrink {
    c32 tau;
    int op;
    float lim;
    __device__ __forceinline__ Shrink(c32 t, int o) : tau(t), op(o), lim(o == 0 ? hard_limit(t) : 0.0f) {}
    __device__ __forceinline__ c32 operator()(c32 X) const;
};

__device__ __forceinline__ c32 Shrink::operator()(c32 X) const
{
    const float p = X.x * X.x + X.y * X.y;
    if (op == 0) {  // hard: where(|X| < tau, 0, X)          threshold_operator.py:110-112
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

__device__ __forceinline__ c32 shrink(c32 X, c32 tau, int op) { return Shrink(tau, op)(X); }

// real data with a real threshold (float32 cubes through the WAVELET / SHEARLET transforms): the same operators on |x|
__device__ __forceinline__ float shrink(float x, c32 tau, int op)
{
    const float m = fabsf(x);
    if (op == 0) return m < tau.x ? 0.f : x;
    if (m == 0.0f) return 0.f;
    const float g = op == 1 ? 1.0f - tau.x / m : 1.0f - (tau.x * tau.x) / (m * m);
    return g > 0.0f ? x * g : 0.f;
}

// Shrink for either coefficient type (the WAVELET kernels are templated on float / c32)
template <typename T>
struct ShrinkOf;
template <>
struct ShrinkOf<c32> : Shrink {
    __device__ __forceinline__ ShrinkOf(c32 t, int o) : Shrink(t, o) {}
};
template <>
struct ShrinkOf<float> {
    c32 tau;
    int op;
    __device__ __forceinline__ ShrinkOf(c32 t, int o) : tau(t), op(o) {}
    __device__ __forceinline__ float operator()(float x) const { return shrink(x, tau, op); }
};

}  // namespace p3d
