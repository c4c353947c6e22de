"""Run on the GPU box: device ms per iteration of the -percentile operators (fused / unfused) against the plain operator."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_3d_interpolation_amd import _ffi
nil = nxl = int(os.environ.get("N", 1024)); ns = int(os.environ.get("NS", 128)); K = int(os.environ.get("K", 10))
rng = np.random.default_rng(0)
mask = (rng.random((nil, nxl)) >= 0.8).astype(np.float32)
x = ((rng.standard_normal((8, nil, nxl)) + 1j * rng.standard_normal((8, nil, nxl))) * mask).astype(np.complex64)
x = np.concatenate([x] * (ns // 8))
with _ffi.Plan(nil, nxl, ns) as plan:
    for op, tau in (("hard", np.full((ns, K), 50.0 + 0j)), ("hard-percentile", np.linspace(99.9, 90.0, K)[None, :].repeat(ns, 0) + 0j)):
        for env in ((None,) if op == "hard" else (None, "P3D_NO_PCT_FUSED")):
            if env: os.environ[env] = "1"
            plan.run(x[:8], mask, tau[:8], K, thresh_op=op)
            out, done, sums, ms = plan.run(x, mask, tau, K, thresh_op=op)
            if env: del os.environ[env]
            print(f"{op:16s} {'unfused' if env else 'fused  '} {nil}x{nxl}x{ns}: {ms / K:7.3f} ms per iteration (device)")
