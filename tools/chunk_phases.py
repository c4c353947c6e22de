#!/usr/bin/env python3
"""Phase timings of one 256-MiB chunk through the pocs_cube chunk worker (GPU box): host copy in, H2D, statistics, loop, D2H,
host copy out -- and the same transfers from pageable memory for comparison."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P

nil = nxl = 1024; n = 32
rng = np.random.default_rng(0)
chunk = (rng.standard_normal((n, nil, nxl)) + 1j * rng.standard_normal((n, nil, nxl))).astype(np.complex64)
dst = np.empty_like(chunk)
mask = (rng.random((nil, nxl)) > 0.8).astype(np.float32)
w = P._FFTWorker.get(nil, nxl, n, 0, 0, mask)
gb = chunk.nbytes / 1e9
def T(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); f(); best = min(best, time.perf_counter() - t)
    return best
xin = w.hx.view(chunk.shape, np.complex64); xout = w.ho.view(chunk.shape, np.complex64)
print("threads", P._COPY_THREADS, "cpus", os.cpu_count())
t = T(lambda: P._slab_copy(xin, chunk)); print(f"copy in  (pageable -> pinned, slabs): {t*1e3:6.1f} ms {gb/t:5.1f} GB/s")
t = T(lambda: np.copyto(xin, chunk)); print(f"copy in  (single np.copyto):          {t*1e3:6.1f} ms {gb/t:5.1f} GB/s")
t = T(lambda: w.x.upload(xin)); print(f"H2D from pinned:                      {t*1e3:6.1f} ms {gb/t:5.1f} GB/s")
t = T(lambda: w.x.upload(chunk)); print(f"H2D from pageable:                    {t*1e3:6.1f} ms {gb/t:5.1f} GB/s")
t = T(lambda: w.plan.stats_dev(w.x.ptr, _ffi.P3D_C64, n)); print(f"statistics:                           {t*1e3:6.1f} ms")
t = T(lambda: w.o.download_into(xout)); print(f"D2H to pinned:                        {t*1e3:6.1f} ms {gb/t:5.1f} GB/s")
t = T(lambda: w.o.download_into(dst)); print(f"D2H to pageable:                      {t*1e3:6.1f} ms {gb/t:5.1f} GB/s")
t = T(lambda: P._slab_copy(dst, xout)); print(f"copy out (pinned -> pageable, slabs): {t*1e3:6.1f} ms {gb/t:5.1f} GB/s")
