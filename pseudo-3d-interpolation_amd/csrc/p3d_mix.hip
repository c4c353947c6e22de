// p3d_mix.hip -- registry of the mixed-radix register-engine plans (p3d_mix.hpp; instantiated in p3d_mix_inst.hip, parts 0 ... 7).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>

#include "p3d_mix_entry.hpp"

namespace p3d {
namespace mix {

const Entry* part_0();
const Entry* part_1();
const Entry* part_2();
const Entry* part_3();
const Entry* part_4();
const Entry* part_5();
const Entry* part_6();
const Entry* part_7();

namespace {

// one thread per word (row, tl)
__global__ void pack_words_kernel(const float* __restrict__ mask, unsigned long long* bits, unsigned* cnt, int* nonbinary, int n1, int n, int tpl, int ppt)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n1 * tpl) return;
    const int row = idx / tpl, tl = idx - row * tpl;
    unsigned long long w = 0;
    bool odd = false;
    for (int q = 0; q < ppt; ++q) {
        const float m = mask[(size_t)row * n + tl + tpl * q];
        w |= (unsigned long long)(m == 1.0f) << q;
        odd = odd || (m != 0.0f && m != 1.0f);
    }
    bits[idx] = w;
    cnt[idx] = (unsigned)__popcll(w);
    if (odd) atomicOr(nonbinary, 1);
}

// exclusive prefix sum of cnt[0 .. total) in place, cnt[total] = sum; one workgroup of 1024 threads, contiguous chunks
__global__ __launch_bounds__(1024) void scan_kernel(unsigned* cnt, int total)
{
    __shared__ unsigned part[1024];
    const int t = threadIdx.x, chunk = (total + 1023) / 1024, lo = t * chunk, hi = min(lo + chunk, total);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned add = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    unsigned run = t ? part[t - 1] : 0u;
    for (int i = lo; i < hi; ++i) {
        const unsigned c = cnt[i];
        cnt[i] = run;
        run += c;
    }
    if (t == 1023) cnt[total] = part[1023];
}

}  // namespace

hipError_t pack_mask(const Entry* e, const float* mask, int n1, unsigned long long* bits, unsigned* base, int* nonbinary, hipStream_t st)
{
    const int words = n1 * e->tpl;
    pack_words_kernel<<<(words + 255) / 256, 256, 0, st>>>(mask, bits, base, nonbinary, n1, e->n, e->tpl, e->ppt);
    scan_kernel<<<1, 1024, 0, st>>>(base, words);
    return hipGetLastError();
}

const Entry* find(int n)
{
    // experiment switch: every non-power-of-two length on the LDS-image passes of p3d_flex.hip.  Read at every call (a static table that looked at the
    // environment once per process made the switch a no-op for every plan after the first); callers release their plans before they flip it
    if (getenv("P3D_NO_MIX")) return nullptr;
    static const std::map<int, const Entry*> table = [] {
        std::map<int, const Entry*> t;
        for (const Entry* (*part)() : {&part_0, &part_1, &part_2, &part_3, &part_4, &part_5, &part_6, &part_7})
            for (const Entry* e = part(); e->n != 0; ++e) t[e->n] = e;
        return t;
    }();
    const auto it = table.find(n);
    return it == table.end() ? nullptr : it->second;
}

}  // namespace mix
}  // namespace p3d
