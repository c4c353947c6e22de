import sys, numpy as np
sys.path.insert(0, '/root/repo')
from pseudo_3d_interpolation_amd import _ffi
rng = np.random.default_rng(0)
for shape in [(3,5),(6,7),(1,8),(9,1),(2,3),(15,14),(31,17),(48,20),(20,48),(4,20),(20,4),(16,20),(20,16),(48,16),(16,48),(90,50),(12,12),(1000,960),(1536,6000),(6400,8),(10,6401),(999,64),(64,999),(1009,37),(37,1009),(1101,1451),(2039,16),(2053,8)]:
    x = (rng.standard_normal((2,)+shape) + 1j*rng.standard_normal((2,)+shape)).astype(np.complex64)
    with _ffi.Plan(shape[0], shape[1], 2) as p:
        F = p.fft2(x); want = np.fft.fft2(x.astype(np.complex128))
        e1 = np.linalg.norm(F-want)/np.linalg.norm(want)
        I = p.fft2(x, inverse=True); want = np.fft.ifft2(x.astype(np.complex128))
        e2 = np.linalg.norm(I-want)/np.linalg.norm(want)
    print(shape, "fwd %.2e inv %.2e" % (e1, e2))
