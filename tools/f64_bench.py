#!/usr/bin/env python3
"""The double-precision loop (precision='reference', p3d_f64.hip) on the GPU box: device time of the loop for a sample of BASELINE configs[2]'s
cube and parity against the float64 oracle.   NIL / NXL / NS / K / OP / DTYPE (complex64 | complex128 | float32 | float64) / NCHECK in the environment;
P3D_F64_UNFUSED=1, P3D_F64_COL_TILE, P3D_F64_ROW_TILE, P3D_F64_COL_THREADS, P3D_F64_ROW_THREADS select the kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats

nil = int(os.environ.get("NIL", 1024)); nxl = int(os.environ.get("NXL", 1024)); ns = int(os.environ.get("NS", 32)); K = int(os.environ.get("K", 20))
op = os.environ.get("OP", "hard"); dtype = np.dtype(os.environ.get("DTYPE", "complex64")); ncheck = int(os.environ.get("NCHECK", 1))
real = dtype.kind == "f"
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s, real=real) for s in range(4)]) * mask
cube = np.ascontiguousarray(np.tile(base, ((ns + 3) // 4, 1, 1))[:ns]).astype(dtype)
with _ffi.Plan64(nil, nxl, ns) as plan:
    st = plan.stats(cube)
    tau = _schedule_from_stats(st, nil * nxl, "exponential", K, 0.99, 1e-3, "values")
    best = None
    for rep in range(3):
        out, done, sums, ms = plan.run(cube, mask, tau, K, thresh_op=op, eps=0.0)
        best = ms if best is None else min(best, ms)
err = []
for s in range(ncheck):
    want = orc.pocs_slice(cube[s].astype(np.float64 if real else np.complex128), mask, niter=K, thresh_op=op, thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
    err.append(float(np.linalg.norm(out[s] - want) / np.linalg.norm(want)))
print(json.dumps({"workload": f"{nil}x{nxl}x{ns} {dtype} cube, {op}, {K} iterations, double-precision loop", "ms_per_iteration": best / K,
                  "slice_iterations_per_s": ns * K / (best * 1e-3), "Gpt_per_s": ns * nil * nxl * K / (best * 1e-3) / 1e9, "rel_l2_vs_oracle": err}))
