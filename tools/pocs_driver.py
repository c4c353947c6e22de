#!/usr/bin/env python3
"""Torch-free driver of the HIP path (ctypes + NumPy only) for rocprofv3 --pmc passes: counter
collection crashes inside torch's own RNG kernels on this image, and the counters of interest belong
to our two kernels anyway.  Same workload shape as bench.py (1024 x 1024 x 512, 80 % missing)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pseudo_3d_interpolation_amd import _ffi  # noqa: E402
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nil", type=int, default=1024)
ap.add_argument("--nxl", type=int, default=1024)
ap.add_argument("--nslices", type=int, default=512)
ap.add_argument("--niter", type=int, default=3)
ap.add_argument("--missing", type=float, default=0.8)
ap.add_argument("--thresh-op", default="hard")
ap.add_argument("--distinct", type=int, default=8)
ap.add_argument("--real", action="store_true", help="float32 (time-domain) cube instead of complex64")
ap.add_argument("--stamps", action="store_true", help="print the in-kernel phase stamps of a -DP3D_STAMPS=1 build (last steady-state row pass)")
a = ap.parse_args()

rng = np.random.default_rng(0)
mask = (np.random.default_rng(42).random((a.nil, a.nxl)) >= a.missing).astype(np.float32)
il = np.arange(a.nil)[:, None] / a.nil
xl = np.arange(a.nxl)[None, :] / a.nxl
base = []
for s in range(a.distinct):
    acc = np.zeros((a.nil, a.nxl), np.complex128)
    for _ in range(6):
        k1, k2 = rng.integers(-(a.nil // 8), max(a.nil // 8, 1)), rng.integers(-(a.nxl // 8), max(a.nxl // 8, 1))
        acc += (rng.standard_normal() + 1j * rng.standard_normal()) * np.exp(2j * np.pi * (k1 * il + k2 * xl))
    acc += 0.01 * (rng.standard_normal(acc.shape) + 1j * rng.standard_normal(acc.shape))
    base.append((acc.real * mask).astype(np.float32) if a.real else (acc * mask).astype(np.complex64))
base = np.stack(base)

plan = _ffi.Plan(a.nil, a.nxl, a.nslices)
slice_bytes = a.nil * a.nxl * (4 if a.real else 8)
DT = _ffi.P3D_F32 if a.real else _ffi.P3D_C64
x = plan.alloc(slice_bytes * a.nslices)
out = plan.alloc(slice_bytes * a.nslices)
m = plan.alloc(mask.nbytes).upload(mask)
for s in range(a.nslices):
    _ffi.check(_ffi.lib().p3d_memcpy_h2d(plan.handle, x.ptr + s * slice_bytes, base[s % a.distinct].ctypes.data, slice_bytes))
stats = plan.stats_dev(x.ptr, DT, a.nslices)
tau = _schedule_from_stats(stats, a.nil * a.nxl, "exponential", a.niter, 0.99, 1e-3, "values")
done, _, ms = plan.run_dev(x.ptr, DT, m.ptr, tau, a.niter, out.ptr, a.nslices, thresh_op=a.thresh_op,
                           profile=True, want_sums=False)
print("niter", a.niter, "device ms", ms, plan.last_profile())

if a.stamps:
    import ctypes
    NPH, names = 10, ["wait work loads", "inverse transform", "wait obs + re-insertion", "sum |x|", "request work (r+1)", "forward transform",
                      "request obs (r+1)", "lock-step barrier", "issue stores", "-"]
    buf = (ctypes.c_uint * (1024 * 16 * NPH))()
    rc = _ffi.lib().p3d_debug_read_stamps(buf, len(buf))
    assert rc == 0, rc
    st = np.frombuffer(buf, dtype=np.uint32).reshape(1024, 16, NPH).astype(np.float64)
    used = st.sum(axis=2) > 0
    rows_per_wave = a.nslices * a.nil / max(int(used.sum()), 1)
    per_row = st[used].mean(axis=0) / rows_per_wave
    print(f"stamps: {int(used.sum())} waves, {rows_per_wave:.1f} rows per wave; mean shader cycles per row and wave:")
    for n, c in zip(names, per_row):
        if n != "-":
            print(f"  {n:26s} {c:9.0f}  ({100 * c / per_row.sum():5.1f} %)")
    print(f"  {'total':26s} {per_row.sum():9.0f}   spread over waves (total): min {st[used].sum(axis=1).min() / rows_per_wave:.0f} max {st[used].sum(axis=1).max() / rows_per_wave:.0f}")
