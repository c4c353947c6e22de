"""Cost of the per-slice contract (POCS_algorithm called once per slice, as under xr.apply_ufunc) vs the batched pocs_cube."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
for n, niter in ((1024, 100), (256, 50), (64, 20)):
    mask = orc.synthetic_mask(n, n, 0.8)
    cube = (np.stack([orc.synthetic_slice(n, n, s) for s in range(16)]) * mask).astype(np.complex64)
    kw = dict(niter=niter, thresh_op="hard", thresh_model="exponential", eps=1e-16, p_max=0.99, p_min=1e-3)
    f, g = np.fft.fft2, np.fft.ifft2   # (checked by name: the kernels are the transform)
    P.POCS_algorithm(cube[0], mask, transform=f, itransform=g, transform_kind="FFT", **kw)
    t0 = time.perf_counter()
    for s in range(16):
        P.POCS_algorithm(cube[s], mask, transform=f, itransform=g, transform_kind="FFT", **kw)
    t1 = time.perf_counter()
    P.pocs_cube(cube, mask, **kw)
    t2 = time.perf_counter()
    P.pocs_cube(cube, mask, **kw)
    t3 = time.perf_counter()
    print(f"{n}x{n}, {niter} iterations: per-slice calls {1e3 * (t1 - t0) / 16:.2f} ms/slice, batched (16 slices) {1e3 * (t3 - t2) / 16:.2f} ms/slice")
