"""Torch-free run of the FFT POCS loop on an arbitrary slice shape (generic path for non-power-of-two extents), for profiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats
nil = int(os.environ.get("NIL", 1000)); nxl = int(os.environ.get("NXL", 1000)); ns = int(os.environ.get("NS", 64)); K = int(os.environ.get("K", 10))
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(4)]) * mask
cube = np.ascontiguousarray(np.tile(base, (ns // 4, 1, 1))).astype(np.complex64)
plan = _ffi.Plan(nil, nxl, ns)
stats = plan.stats(cube)
tau = _schedule_from_stats(stats, nil * nxl, "exponential", K, 0.99, 1e-3, "values")
for rep in range(2):
    out, done, sums, ms = plan.run(cube, mask.astype(np.float32), tau, K, thresh_op="hard", eps=0.0)
print(f"{nil}x{nxl}x{ns}: {ms / K:.3f} ms per iteration, {ns * nil * nxl * K / ms / 1e6:.2f} Gpt/s")
