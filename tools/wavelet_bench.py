"""BASELINE configs[3]: 512x512x256 cube, db4 wavelet, 50 iterations, soft threshold, one MI355X.
Prints the loop rate (device time of p3d_wavelet_run) and checks NCHECK slices against the wavelet oracle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as po, wavelet_oracle as wo
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P

nil = nxl = int(os.environ.get("N", 512))
ns = int(os.environ.get("NS", 256))
K = int(os.environ.get("K", 50))
ncheck = int(os.environ.get("NCHECK", 2))
wavelet = os.environ.get("WAVELET", "db4")
mask = po.synthetic_mask(nil, nxl, 0.7)
cplx = bool(int(os.environ.get("COMPLEX", 0)))   # COMPLEX=1: a complex64 cube (frequency slices) instead of configs[3]'s float32 one
base = np.stack([po.synthetic_slice(nil, nxl, s, real=not cplx) for s in range(8)])
cube = np.concatenate([np.roll(base, 7 * r, axis=2) for r in range((ns + 7) // 8)])[:ns] * mask
cube = cube.astype(np.complex64 if cplx else np.float32)
kw = dict(thresh_op="soft", thresh_model="exponential", niter=K, p_max=0.99, p_min=1e-2, eps=0.0)
plan = P._get_wavelet_plan(nil, nxl, ns, wavelet, 0)
stats = plan.stats(cube)
tau = P._wavelet_schedule_from_stats(stats, kw["thresh_model"], K, kw["p_max"], kw["p_min"], "values")
best = None
for rep in range(3):
    t0 = time.perf_counter()
    out, done, sums, ms = plan.run(cube, mask.astype(np.float32), tau, K, thresh_op="soft", eps=0.0)
    wall = time.perf_counter() - t0
    best = ms if best is None else min(best, ms)
err = []
for s in range(ncheck):
    want = wo.pocs_slice_wavelet(cube[s].astype(np.complex128 if cplx else np.float64), mask, wavelet=wavelet, **kw)
    err.append(float(np.linalg.norm(out[s] - want) / np.linalg.norm(want)))
pts = ns * nil * nxl
print(json.dumps({"workload": f"{nil}x{nxl}x{ns} {cube.dtype} cube, {wavelet}, soft, {K} iterations (BASELINE configs[3])", "nlev": plan.nlev,
                  "iterations_per_s": K / (best * 1e-3), "ms_per_iteration": best / K, "wall_s_incl_pcie": wall,
                  "rel_l2_vs_oracle": err, "points_per_s": pts * K / (best * 1e-3)}))
