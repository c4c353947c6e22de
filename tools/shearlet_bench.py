"""BASELINE configs[4] in one-GPU portions: 2048x1024 slices, shearlet frame (J=5: 125 shearlets), NB slices per batch, K iterations.
Prints slice-iterations/s (device time of p3d_shearlet_run) and checks Parseval reconstruction on the device."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as po
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P, shearlets

nil = int(os.environ.get("NIL", 2048)); nxl = int(os.environ.get("NXL", 1024))
nb = int(os.environ.get("NB", 4)); K = int(os.environ.get("K", 5))
t0 = time.perf_counter()
psi = shearlets.scalesShearsAndSpectra((nil, nxl), dtype=np.float32)
t_psi = time.perf_counter() - t0
mask = po.synthetic_mask(nil, nxl, 0.7)
cube = np.stack([po.synthetic_slice(nil, nxl, s, real=True) for s in range(nb)]) * mask
cube = cube.astype(np.float32)
plan = _ffi.ShearletPlan(psi, max_slices=nb)
stats = plan.stats(cube)
tau = P._shearlet_schedule_from_stats(stats, (nil, nxl), "exponential", K, 0.99, 1e-2, "values")
best = None
for rep in range(2):
    out, done, sums, ms = plan.run(cube, mask.astype(np.float32), tau, K, thresh_op="soft", eps=0.0)
    best = ms if best is None else min(best, ms)
x0 = cube[:1].astype(np.complex64)
rec = plan.inverse(plan.transform(x0))
print(json.dumps({"workload": f"{nil}x{nxl} float32 slices, {psi.shape[-1]} shearlets, soft, {nb} slices x {K} iterations (BASELINE configs[4] shape)",
                  "psi_setup_s": t_psi, "slice_iterations_per_s": nb * K / (best * 1e-3), "ms_per_slice_iteration": best / (nb * K),
                  "cube_1024_slices_iterations_per_s_per_gpu": nb * K / (best * 1e-3) / 1024,
                  "perfect_reconstruction_rel_l2": float(np.linalg.norm(rec - x0) / np.linalg.norm(x0))}))
