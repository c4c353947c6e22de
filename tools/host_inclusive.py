#!/usr/bin/env python3
"""PCIe-inclusive rate of the POCS job (host NumPy arrays in -> host arrays out through p3d_pocs_run): NOT the bench value,
recorded for DESIGN.md section 5."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P

nil = nxl = 1024
nslices, niter = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 100
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(8)]) * mask
cube = np.ascontiguousarray(np.tile(base, (nslices // 8, 1, 1)))
params = dict(niter=niter, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
P.pocs_cube(cube[:8], mask, **dict(params, niter=3))          # plan + library warm-up
t0 = time.perf_counter()
out = P.pocs_cube(cube, mask, **params)
dt = time.perf_counter() - t0
print(f"host-inclusive: {nslices} slices x {niter} iterations of {nil}x{nxl} complex64 in {dt:.3f} s "
      f"= {niter / dt * nslices / 512:.1f} cube-equivalent iterations/s (cube = 512 slices); "
      f"{cube.nbytes / 2**30:.2f} GiB in + out over PCIe, pageable host memory")
