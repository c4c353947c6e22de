"""Run on the GPU box.  float32 cubes + hard threshold + 'inverse_proportional' (first threshold = max |X|): the spectrum of a real
slice is Hermitian, so the maximum modulus belongs to a conjugate PAIR whose two float32 moduli agree only up to an ulp -- which of
the two survives `|X| < tau` differs between FFT implementations (NumPy included) and moves the result by 1e-2 ... 1e-1.  Every
path (row pairs, complex rows, flexible lengths, unfused) shows it on the same kind of slice: conditioning, not a defect
(decision-level parity)."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import pseudo_3d_interpolation_amd.functions.POCS as P
from oracle import pocs_oracle as orc
def run(nil, nxl, ns, kw, missing, seed0):
    mask = orc.synthetic_mask(nil, nxl, missing)
    cube = np.stack([orc.synthetic_slice(nil, nxl, seed0 + s) for s in range(ns)]) * mask
    cube = cube.real.astype(np.float32)
    want = orc.pocs_cube(cube.astype(np.float64), mask, **kw)
    for env in (None, "P3D_NO_REAL", "P3D_NO_FLEX"):
        P.release_plans()
        if env: os.environ[env] = "1"
        got = P.pocs_cube(cube, mask, **kw)
        if env: del os.environ[env]
        err = [float(np.linalg.norm(got[s] - want[s]) / np.linalg.norm(want[s])) for s in range(ns)]
        print((nil, nxl), env, ["%.1e" % e for e in err])
kw = dict(niter=2, thresh_op="hard", thresh_model="inverse_proportional", eps=1e-9, alpha=0.8, p_max=0.99, p_min=1e-2, version="regular")
for seed0 in (1, 2, 3):
    run(37, 100, 3, kw, 0.3, seed0)
    run(32, 60, 3, dict(kw, alpha=1.0, eps=0.0, niter=3), 0.3, seed0)
    run(64, 64, 3, dict(kw, alpha=1.0, eps=0.0, niter=3), 0.3, seed0)
