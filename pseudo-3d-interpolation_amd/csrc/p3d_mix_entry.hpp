// p3d_mix_entry.hpp -- the mixed-radix register engine (p3d_mix.hpp) seen from the launchers of p3d_flex.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "p3d_fft.hpp"

namespace p3d {
struct RowArgs;
struct ColArgs;
namespace mix {

// what the launchers of p3d_flex.hip need to know about a plan
struct Entry {
    int n, col_tile, tw_slots, tpl, ppt;
    void (*build_tw)(c32* out);
    hipError_t (*row)(int mode, const RowArgs& a, const c32* tab, hipStream_t st);   // nullptr: the plan serves columns only (radix-11 / 13 passes)
    hipError_t (*col)(int mode, const ColArgs& a, const c32* tab, hipStream_t st);
};
const Entry* find(int n);   // nullptr: no plan for this length (p3d_flex.hip runs it as an LDS image)
// Packed form of a trace mask for the row pass of plan `e` (rows of e->n samples): bits[row * tpl + tl] bit q = (mask[row][tl + tpl q] == 1),
// base[row * tpl + tl] = number of set bits in the words before it (base[n1 * tpl] = observed positions per slice); *nonbinary is raised
// when an entry of the mask is neither 0 nor 1 (the float weights are used then).  Runs on `st`, does not wait.
hipError_t pack_mask(const Entry* e, const float* mask, int n1, unsigned long long* bits, unsigned* base, int* nonbinary, hipStream_t st);

}  // namespace mix
}  // namespace p3d
