import sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
for n in (512, 1024, 1000):
    mask = orc.synthetic_mask(n, n, 0.8)
    x = (orc.synthetic_slice(n, n, 1) * mask).astype(np.complex64)
    kw = dict(transform=np.fft.fft2, itransform=np.fft.ifft2, transform_kind='FFT', niter=100, thresh_op='hard', thresh_model='exponential', eps=0, p_max=0.99, p_min=1e-3)
    P.POCS(x, mask, **kw)
    t0 = time.perf_counter()
    for _ in range(20):
        P.POCS(x, mask, **kw)
    print(n, 'per-slice call (100 iterations):', (time.perf_counter() - t0) / 20 * 1e3, 'ms')
