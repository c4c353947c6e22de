"""Temporal blocking experiment: run all K iterations on groups of G slices (working set G x 8 MiB) instead of sweeping the
whole cube once per iteration, so that the work buffer of a group could stay in the 256 MiB Infinity Cache between iterations.
GROUPS=512,64,... K=50 NS=512 python tools/group_sweep.py      (torch-free: ctypes + NumPy)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats

nil = nxl = 1024
ns = int(os.environ.get("NS", 512))
K = int(os.environ.get("K", 100))
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(8)]) * mask
cube = np.ascontiguousarray(np.tile(base, (ns // 8, 1, 1))).astype(np.complex64)
maskf = mask.astype(np.float32)
per = nil * nxl * 8
holder = _ffi.Plan(nil, nxl, 1)
x, out, m = holder.alloc(cube.nbytes).upload(cube), holder.alloc(cube.nbytes), holder.alloc(maskf.nbytes).upload(maskf)
for G in [int(g) for g in os.environ.get("GROUPS", "512,64,32,24,16,12,8,4").split(",")]:
    plan = _ffi.Plan(nil, nxl, G)
    stats = np.concatenate([plan.prime_dev(x.ptr + g * per, _ffi.P3D_C64, m.ptr, min(G, ns - g)) for g in range(0, ns, G)])
    tau = _schedule_from_stats(stats, nil * nxl, "exponential", K, 0.99, 1e-3, "values")
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for g in range(0, ns, G):
            n = min(G, ns - g)
            plan.run_dev(x.ptr + g * per, _ffi.P3D_C64, m.ptr, tau[g:g + n], K, out.ptr + g * per, n, thresh_op="hard", eps=0.0, want_sums=False)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"G={G:4d}  {K / best:8.1f} it/s   {best * 1e3 / K:7.3f} ms/iteration of the {ns} slices (wall, statistics pass excluded)", flush=True)
    plan.close()
