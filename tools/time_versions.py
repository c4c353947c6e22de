"""Run on the GPU box: device ms per iteration of the FFT loop for version = regular / fast / adaptive (POCS / FPOCS / APOCS)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats
from oracle import pocs_oracle as orc
nil = nxl = int(os.environ.get("N", 1024)); ns = int(os.environ.get("NS", 128)); K = int(os.environ.get("K", 20))
eps = float(os.environ.get("EPS", 0.0))   # > 0: with the convergence test (APOCS then stores every iterate)
mask = orc.synthetic_mask(nil, nxl, 0.8)
x = np.stack([orc.synthetic_slice(nil, nxl, s) * mask for s in range(8)]).astype(np.complex64)
x = np.concatenate([x] * (ns // 8))
with _ffi.Plan(nil, nxl, ns) as plan:
    stats = plan.stats(x)
    tau = _schedule_from_stats(stats, nil * nxl, "exponential", K, 0.99, 1e-3, "values")
    for version, alpha in (("regular", 1.0), ("fast", 1.0), ("adaptive", 1.0), ("adaptive", 0.8), ("regular", 0.8)):
        plan.run(x[:8], mask, tau[:8], K, thresh_op="hard", version=version, alpha=alpha, eps=eps)
        out, done, sums, ms = plan.run(x, mask, tau, K, thresh_op="hard", version=version, alpha=alpha, eps=eps)
        print(f"{version:9s} alpha {alpha}: {ms / K:7.3f} ms per iteration (device), {nil}x{nxl}x{ns}", plan.last_profile() if hasattr(plan, "last_profile") else "")
