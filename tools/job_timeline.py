#!/usr/bin/env python3
"""Fixed cost of one resident FFT job (GPU box): the phases of bench.py's job() -- statistics pass (prime_dev), host schedule, loop
(run_dev) -- as host wall time and device time, for the whole headline cube and for the 1/8 block a rank works on at N = 8.

    python tools/job_timeline.py [niter=20]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nil = nxl = 1024
dev = torch.device("cuda", 0)
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(8)]) * mask
for n in (512, 64):
    x = torch.from_numpy(np.ascontiguousarray(np.tile(base, (n // 8, 1, 1)))).to(dev)
    out = torch.empty_like(x)
    m = torch.from_numpy(mask.astype(np.float32)).to(dev)
    torch.cuda.synchronize()
    plan = _ffi.Plan(nil, nxl, n, device=0)
    rows = []
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = plan.prime_dev(x.data_ptr(), _ffi.P3D_C64, m.data_ptr(), n)
        t1 = time.perf_counter()
        active = stats[:, 2] > 0
        stats[~active] = 1.0
        tau = P._schedule_from_stats(stats, nil * nxl, "exponential", K, 0.99, 1e-3, "values")
        t2 = time.perf_counter()
        done, _, ms = plan.run_dev(x.data_ptr(), _ffi.P3D_C64, m.data_ptr(), tau, K, out.data_ptr(), n, thresh_op="hard", eps=0.0, alpha=1.0,
                                   active=active, want_sums=False, primed=True)
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, ms * 1e-3))
    r = np.median(np.array(rows[2:]), axis=0) * 1e3
    print(f"{n:4d} slices, K = {K}: job {r[:3].sum():7.3f} ms = prime_dev {r[0]:6.3f} + host schedule {r[1]:6.3f} + run_dev {r[2]:7.3f} (device time of the loop "
          f"incl. last pass {r[3]:7.3f}: host-side overhead of run_dev {r[2]-r[3]:6.3f})", flush=True)
    plan.close()
    del x, out
    torch.cuda.empty_cache()
