#!/usr/bin/env python3
"""Does a small resident job gain from running as TWO half-jobs side by side (two plans, two streams, two host threads)?  (GPU box)
The tails of one half's kernels and its host round trip (statistics -> schedule -> thresholds) then overlap the other half's kernels.
    python tools/job_split.py [niter=20] [nslices=64]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nil = nxl = 1024
dev = torch.device("cuda", 0)
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(8)]) * mask
x = torch.from_numpy(np.ascontiguousarray(np.tile(base, (n // 8, 1, 1)))).to(dev)
out = torch.empty_like(x)
m = torch.from_numpy(mask.astype(np.float32)).to(dev)
torch.cuda.synchronize()
per = nil * nxl * 8

def job(plan, lo, cnt):
    xp, op = x.data_ptr() + lo * per, out.data_ptr() + lo * per
    stats = plan.prime_dev(xp, _ffi.P3D_C64, m.data_ptr(), cnt)
    active = stats[:, 2] > 0
    stats[~active] = 1.0
    tau = P._schedule_from_stats(stats, nil * nxl, "exponential", K, 0.99, 1e-3, "values")
    plan.run_dev(xp, _ffi.P3D_C64, m.data_ptr(), tau, K, op, cnt, thresh_op="hard", eps=0.0, alpha=1.0, active=active, want_sums=False, primed=True)

for parts in (1, 2, 4):
    plans = [_ffi.Plan(nil, nxl, n // parts, device=0) for _ in range(parts)]
    def run_all():
        if parts == 1:
            job(plans[0], 0, n)
        else:
            th = [threading.Thread(target=job, args=(plans[i], i * (n // parts), n // parts)) for i in range(parts)]
            [t.start() for t in th]; [t.join() for t in th]
    run_all(); run_all()
    ts = []
    for rep in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run_all(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ref = out.clone() if parts == 1 else ref
    same = bool(torch.equal(out, ref))
    print(f"{n} slices, K = {K}, {parts} part(s) side by side: median {np.median(ts)*1e3:7.3f} ms  min {np.min(ts)*1e3:7.3f} ms  same bits as one part: {same}", flush=True)
    for p_ in plans: p_.close()
