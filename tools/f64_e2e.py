#!/usr/bin/env python3
"""Wall time of pocs_cube on a complex128 cube (64 slices of 1024 x 1024, K = 20): the double-precision loop through the host-buffer entry point (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
nil = nxl = 1024; ns = 64
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(4)]) * mask
cube = np.ascontiguousarray(np.tile(base, (ns // 4, 1, 1))).astype(np.complex128)
for rep in range(3):
    t0 = time.perf_counter(); out = P.pocs_cube(cube, mask, niter=20, thresh_op="hard", eps=0, p_min=1e-3); t1 = time.perf_counter()
    print(f"pocs_cube complex128 64 x 1024^2, K=20: {1e3*(t1-t0):7.1f} ms", flush=True)
want = orc.pocs_slice(cube[1], mask, niter=20, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
print("rel l2 vs oracle", np.linalg.norm(out[1]-want)/np.linalg.norm(want))
