#!/usr/bin/env python3
"""Repeated host-inclusive calls (GPU box): does the rate depend on the result array being fresh memory?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
nil = nxl = 1024; nslices = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(8)]) * mask
cube = np.ascontiguousarray(np.tile(base, (nslices // 8, 1, 1)))
params = dict(niter=100, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
P.pocs_cube(cube[:8], mask, **dict(params, niter=3))
outs = []
for rep in range(5):
    t0 = time.perf_counter(); out = P.pocs_cube(cube, mask, **params); dt = time.perf_counter() - t0
    keep = rep < 2
    print(f"call {rep}: {dt:.3f} s ({'previous result kept alive' if keep else 'previous result freed before the call'})", flush=True)
    if keep: outs.append(out)
    else: outs.clear()
    del out
