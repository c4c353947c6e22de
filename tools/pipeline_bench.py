"""Steps 12 -> 13 -> 14 on a time-domain cube: three host-level calls (one PCIe round trip each) vs the device-resident pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi, pipeline
from pseudo_3d_interpolation_amd.functions import POCS as P
nt, nil, nxl = int(os.environ.get("NT", 512)), int(os.environ.get("NIL", 512)), int(os.environ.get("NXL", 512))
rng = np.random.default_rng(0)
mask = orc.synthetic_mask(nil, nxl, 0.7)
t = np.arange(nt)[:, None, None]
x = np.zeros((nt, nil, nxl), np.float32)
for _ in range(4):
    f, k1, k2 = rng.uniform(0.02, 0.2), rng.integers(-6, 7), rng.integers(-6, 7)
    x += (np.cos(2 * np.pi * (f * t + k1 * np.arange(nil)[None, :, None] / nil + k2 * np.arange(nxl)[None, None, :] / nxl))).astype(np.float32)
x *= mask
kw = dict(niter=50, thresh_op="hard", thresh_model="exponential", eps=1e-16, alpha=1.0, p_max=0.99, p_min=1e-3)
dt, t0 = 0.004, 0.0
pipeline.interpolate_time_cube(x[:16], mask, dt, t0, **dict(kw, niter=3))        # warm-up
a0 = time.perf_counter()
F = _ffi.time2freq(x, dt, t0, real_only=True)
a1 = time.perf_counter()
G = P.pocs_cube(F, mask, **kw)
a2 = time.perf_counter()
y3 = _ffi.freq2time(G, dt, t0, nfft=nt, real_only=True)
a3 = time.perf_counter()
y1 = pipeline.interpolate_time_cube(x, mask, dt, t0, real_only=True, **kw)
a4 = time.perf_counter()
err = float(np.linalg.norm(y1 - y3) / np.linalg.norm(y3))
print(f"{nt}x{nil}x{nxl} float32, 50 iterations: three steps {a3 - a0:.3f} s (12: {a1 - a0:.3f}, 13: {a2 - a1:.3f}, 14: {a3 - a2:.3f}); "
      f"device-resident pipeline {a4 - a3:.3f} s; rel-L2 between them {err:.1e}")
