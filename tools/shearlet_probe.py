#!/usr/bin/env python3
"""Device time per slice-iteration of the SHEARLET loop, float32 kernels vs the double-precision loop, for a few slice shapes (NIL NXL pairs in argv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P, shearlets

K = 6
args = [int(v) for v in sys.argv[1:]] or [1000, 1000]
for nil, nxl in zip(args[0::2], args[1::2]):
    psi = shearlets.scalesShearsAndSpectra((nil, nxl))
    mask = orc.synthetic_mask(nil, nxl, 0.8)
    x = (orc.synthetic_slice(nil, nxl, 3, real=True) * mask).astype(np.float32)[None]
    out = {}
    for name, cls in (("float32", _ffi.ShearletPlan), ("double", _ffi.ShearletPlan64)):
        with cls(psi, max_slices=1) as plan:
            tau = P._shearlet_schedule_from_stats(plan.stats(x), (nil, nxl), "exponential", K, 0.99, 1e-3, "values")
            ms = min(plan.run(x, mask, tau, K, thresh_op="hard")[3] for _ in range(3))
            out[name] = (ms / K, getattr(plan, "fused", None), getattr(plan, "row_group_fraction", None))
    print(f"{nil} x {nxl} x {psi.shape[2]} shearlets: float32 {out['float32'][0]:.2f} ms per slice-iteration (rows touched {out['float32'][2]}), "
          f"double {out['double'][0]:.2f} ms (fused {out['double'][1]}, rows touched {out['double'][2]:.2f})")
