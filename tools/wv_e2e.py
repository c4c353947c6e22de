#!/usr/bin/env python3
"""Where the wall time of a WAVELET job through the host-buffer entry point goes (BASELINE configs[3]'s cube): pocs_cube as it is against its pieces
(statistics call, loop call, the copy into the result) and against the same job with ONE upload, device-resident statistics + loop, and a download
straight into the result array."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as po
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P

nil = nxl = 512; ns = 256; K = 50
mask = po.synthetic_mask(nil, nxl, 0.7)
base = np.stack([po.synthetic_slice(nil, nxl, s, real=True) for s in range(8)])
cube = (np.concatenate([np.roll(base, 7 * r, axis=2) for r in range(ns // 8)]) * mask).astype(np.float32)
kw = dict(transform_kind="WAVELET", wavelet="db4", thresh_op="soft", thresh_model="exponential", niter=K, p_max=0.99, p_min=1e-2, eps=0.0)
for rep in range(3):
    t0 = time.perf_counter(); a = P.pocs_cube(cube, mask, **kw); t1 = time.perf_counter()
    print(f"pocs_cube: {1e3*(t1-t0):7.1f} ms")
plan = P._get_wavelet_plan(nil, nxl, ns, "db4", 0)
maskf = mask.astype(np.float32)
for rep in range(2):
    t0 = time.perf_counter(); st = plan.stats(cube); t1 = time.perf_counter()
    tau = P._wavelet_schedule_from_stats(st, "exponential", K, 0.99, 1e-2, "values"); t2 = time.perf_counter()
    res, done, sums, ms = plan.run(cube, maskf, tau, K, thresh_op="soft", eps=0.0); t3 = time.perf_counter()
    out = np.empty_like(cube); out[:] = res; t4 = time.perf_counter()
    print(f"pieces: stats {1e3*(t1-t0):6.1f}  schedule {1e3*(t2-t1):6.1f}  run {1e3*(t3-t2):6.1f} (device loop {ms:5.1f})  copy into out {1e3*(t4-t3):6.1f}  total {1e3*(t4-t0):6.1f} ms")
holder = _ffi.Plan(4, 4, 1)
for rep in range(3):
    t0 = time.perf_counter()
    xd, od, md = holder.alloc(cube.nbytes), holder.alloc(cube.nbytes), holder.alloc(maskf.nbytes)
    ta = time.perf_counter()
    xd.upload(cube); md.upload(maskf); t1 = time.perf_counter()
    st = plan.stats_dev(xd.ptr, _ffi.P3D_F32, ns); t2 = time.perf_counter()
    tau = P._wavelet_schedule_from_stats(st, "exponential", K, 0.99, 1e-2, "values"); t3 = time.perf_counter()
    done, sums, ms = plan.run_dev(xd.ptr, _ffi.P3D_F32, md.ptr, tau, K, od.ptr, ns, thresh_op="soft", eps=0.0); t4 = time.perf_counter()
    out2 = np.empty_like(cube); od.download_into(out2); t5 = time.perf_counter()
    for b in (xd, od, md): b.free()
    t6 = time.perf_counter()
    print(f"resident: alloc {1e3*(ta-t0):5.1f}  upload {1e3*(t1-ta):5.1f}  stats {1e3*(t2-t1):5.1f}  schedule {1e3*(t3-t2):5.1f}  run {1e3*(t4-t3):5.1f} (loop {ms:5.1f})  download {1e3*(t5-t4):5.1f}  free {1e3*(t6-t5):5.1f}  total {1e3*(t6-t0):6.1f} ms; same result: {np.array_equal(out2, a)}")
