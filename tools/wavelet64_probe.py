#!/usr/bin/env python3
"""Device time per iteration of the double-precision WAVELET loop (p3d_wavelet64.hip) for batches of 16 ... 256 slices of configs[3]'s shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P

n, K = 512, 20
mask = orc.synthetic_mask(n, n, 0.7)
base = np.stack([orc.synthetic_slice(n, n, s, real=True) for s in range(4)]) * mask
for ns in (16, 64, 256):
    cube = np.ascontiguousarray(np.tile(base, (ns // 4, 1, 1))).astype(np.float32)
    with _ffi.WaveletPlan64(n, n, ns, wavelet="db4") as plan:
        tau = P._wavelet_schedule_from_stats(plan.stats(cube), "exponential", K, 0.99, 1e-3, "values")
        ms = min(plan.run(cube, mask, tau, K, thresh_op="soft")[3] for _ in range(3))
    print(f"{ns} slices: {ms / K:.3f} ms per iteration, {ns * K / (ms * 1e-3):.0f} slice-iterations/s, cube-equivalent {ns * K / (ms * 1e-3) / 256:.0f} it/s")
