// p3d_internal.hpp -- what the translation units of libp3d_hip.so share beyond the public C ABI (include/p3d.h).
#pragma once
#include <hip/hip_runtime.h>

#include "p3d.h"
#include "p3d_fft.hpp"

namespace p3d {

// thread-local message returned by p3d_last_error() (p3d_api.hip)
void set_last_error(const char* msg);

// the stream all launches of a plan go to
hipStream_t plan_stream(p3d_plan* plan);

// p3d_fft2_c64_dev without the trailing stream synchronisation: batched 2-D FFT of complex64 slices on device buffers,
// numpy.fft conventions, in == out allowed; enqueued on plan_stream(plan)
int fft2_async(p3d_plan* plan, const c32* in, c32* out, int nslices, int inverse);

}  // namespace p3d
