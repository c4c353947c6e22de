"""GPU diagnostic (not a test): where does a HIP-vs-oracle difference first appear?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
from pseudo_3d_interpolation_amd import _ffi

def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))

nil = nxl = 64
_, mask, obs = orc.synthetic_cube(nil, nxl, 8, 0.5)
base = dict(thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
with _ffi.Plan(nil, nxl, 8) as plan:
    X = plan.fft2(obs)
    Xr = np.fft.fft2(obs.astype(np.complex128))
    print("fft2 rel err per slice", [f"{rel(X[s], Xr[s]):.2e}" for s in range(8)])
    print("np c64 fft2 rel err  ", [f"{rel(np.fft.fft2(obs[s]), Xr[s]):.2e}" for s in range(8)])
for s in (5, 6):
    for niter in (20,):
        infos = []
        w = orc.pocs_cube(obs[s:s+1].astype(np.complex128), mask, niter=niter, infos=infos, **base)
        tau = infos[0]["tau"]
        # replay the oracle iteration by iteration and compare with a GPU run truncated at k iterations
        # (same schedule: pass tau explicitly through the plan)
        with _ffi.Plan(nil, nxl, 1) as plan:
            prev = obs[s].astype(np.complex128)
            for k in range(1, niter + 1):
                got, done, sums, ms = plan.run(obs[s:s+1], mask.astype(np.float32), tau[None, :k], k)
                spec = np.fft.fft2(prev)
                mags = np.abs(spec)
                margin = np.min(np.abs(mags - tau[k-1].real)) / tau[k-1].real
                spec = np.where(mags < tau[k-1].real, 0, spec)
                cur = np.fft.ifft2(spec) * (1 - mask) + obs[s]
                prev = cur
                print(f"slice {s} iter {k:2d} rel {rel(got[0], cur):.2e}  closest |X| to tau (rel) {margin:.2e} kept {np.count_nonzero(spec)}")
