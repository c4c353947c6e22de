#!/usr/bin/env python3
"""Where the wall time of the host-buffer entry point goes (GPU box): functions.POCS.pocs_cube on BASELINE configs[2]'s cube
(pageable NumPy in, NumPy out), phases of every chunk of every worker -- H2D, statistics pass, schedule, loop, D2H -- as totals per
phase, busy time per worker and the idle gaps, plus what lies outside the workers (allocation of the result, thread pool).

    python tools/e2e_timeline.py [niter=20] [nslices=512] [prefault: 0|1]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P

niter = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nslices = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nil = nxl = 1024
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(8)]) * mask
cube = np.ascontiguousarray(np.tile(base, (nslices // 8, 1, 1)))
params = dict(niter=niter, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
P.pocs_cube(cube[:64], mask, **dict(params, niter=2))
for rep in range(3):
    P._timeline = []
    t0 = time.perf_counter()
    out = P.pocs_cube(cube, mask, **params)
    wall = time.perf_counter() - t0
    tl, P._timeline = P._timeline, None
    tl = [(w, m) for w, m in tl if w != 'setup']
    first = min(m[0][1] for _, m in tl)
    last = max(m[-1][1] for _, m in tl)
    tot = {}
    per_worker = {}
    for wid, marks in tl:
        for (n0, a), (n1, b) in zip(marks[:-1], marks[1:]):
            tot[n1] = tot.get(n1, 0.0) + (b - a)
        per_worker.setdefault(wid, []).append((marks[0][1], marks[-1][1]))
    print(f"call {rep}: wall {wall*1e3:7.1f} ms = {niter/wall:6.1f} it/s; before the first chunk starts {1e3*(first-t0):6.1f} ms, after the last chunk ends "
          f"{1e3*(t0+wall-last):6.1f} ms; {len(tl)} chunks on {len(per_worker)} workers", flush=True)
    print("   phase totals over all chunks (ms; divide by the workers for the wall share): " + "  ".join(f"{k} {v*1e3:7.1f}" for k, v in tot.items()))
    for i, (wid, spans) in enumerate(sorted(per_worker.items())):
        busy = sum(b - a for a, b in spans)
        print(f"   worker {i}: {len(spans)} chunks, busy {busy*1e3:7.1f} ms, first start +{1e3*(spans[0][0]-t0):6.1f} ms, last end +{1e3*(spans[-1][1]-t0):7.1f} ms")
    del out
