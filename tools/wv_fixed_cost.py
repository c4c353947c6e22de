#!/usr/bin/env python3
"""Where the fixed cost of a WAVELET job goes (configs[3] shape): statistics pass, host schedule, the loop's own set-up."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_3d_interpolation_amd import _ffi
import pseudo_3d_interpolation_amd.functions.POCS as P

nil = nxl = 512; ns = 256; K = 50
rng = np.random.default_rng(0)
mask = (rng.random((nil, nxl)) >= 0.7).astype(np.float32)
x = torch.from_numpy((rng.standard_normal((ns, nil, nxl)).astype(np.float32)) * mask).cuda()
m = torch.from_numpy(mask).cuda(); out = torch.empty_like(x)
plan = _ffi.WaveletPlan(nil, nxl, ns, wavelet="db4", device=0)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    sync(); t0 = time.perf_counter()
    stats = plan.stats_dev(x.data_ptr(), _ffi.P3D_F32, ns); t1 = time.perf_counter()
    tau = P._wavelet_schedule_from_stats(stats, "exponential", K, 0.99, 1e-3, "values"); t2 = time.perf_counter()
    r = plan.run_dev(x.data_ptr(), _ffi.P3D_F32, m.data_ptr(), tau, K, out.data_ptr(), ns, thresh_op="soft"); sync(); t3 = time.perf_counter()
    print(f"stats {1e3*(t1-t0):.3f} ms  schedule {1e3*(t2-t1):.3f} ms  run {1e3*(t3-t2):.3f} ms (device loop {r[-1]:.3f} ms)")
