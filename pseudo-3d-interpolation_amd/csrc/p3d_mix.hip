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

const Entry* find(int n)
{
    static const std::map<int, const Entry*> table = [] {
        std::map<int, const Entry*> t;
        if (getenv("P3D_NO_MIX")) return t;   // experiment switch: every non-power-of-two length on the LDS-image passes of p3d_flex.hip
        for (const Entry* (*part)() : {&part_0, &part_1, &part_2, &part_3, &part_4, &part_5, &part_6, &part_7})
            for (const Entry* e = part(); e->n != 0; ++e) t[e->n] = e;
        return t;
    }();
    const auto it = table.find(n);
    return it == table.end() ? nullptr : it->second;
}

}  // namespace mix
}  // namespace p3d
