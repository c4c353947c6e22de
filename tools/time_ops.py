"""Run on the GPU box: device time of the FFT POCS loop per thresholding operator (the -percentile ones take the unfused path)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
nil = nxl = int(os.environ.get("N", 1024)); ns = int(os.environ.get("NS", 32)); K = int(os.environ.get("K", 10))
mask = orc.synthetic_mask(nil, nxl, 0.8)
cube = np.stack([orc.synthetic_slice(nil, nxl, s) * mask for s in range(4)]).astype(np.complex64)
cube = np.concatenate([cube] * (ns // 4))
for op, pmax, pmin in (("hard", 0.99, 1e-3), ("hard-percentile", 99.9, 90.0), ("soft-percentile", 99.9, 90.0)):
    for model in ("exponential",) + (("data-driven",) if op == "hard" else ()):
        res = []
        P.pocs_cube(cube[:4], mask, niter=2, thresh_op=op, thresh_model=model, eps=0.0, p_max=pmax, p_min=pmin)
        t0 = time.perf_counter()
        P.pocs_cube(cube, mask, niter=K, thresh_op=op, thresh_model=model, eps=0.0, p_max=pmax, p_min=pmin, results=res)
        dt = time.perf_counter() - t0
        print(f"{op:16s} {model:12s} {nil}x{nxl}x{ns}, {K} iterations: wall {dt * 1e3:8.1f} ms = {K / dt:7.1f} it/s")
