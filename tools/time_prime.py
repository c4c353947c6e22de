import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats
nil, nxl, ns, K = int(sys.argv[1]), int(sys.argv[2]), 128, 20
rng = np.random.default_rng(0)
mask = (rng.random((nil, nxl)) >= 0.8).astype(np.float32)
x = ((rng.standard_normal((8, nil, nxl)) + 1j * rng.standard_normal((8, nil, nxl))) * mask).astype(np.complex64)
plan = _ffi.Plan(nil, nxl, ns)
per = nil * nxl * 8
dx = plan.alloc(per * ns); out = plan.alloc(per * ns); m = plan.alloc(mask.nbytes).upload(mask)
for s in range(ns):
    _ffi.check(_ffi.lib().p3d_memcpy_h2d(plan.handle, dx.ptr + s * per, x[s % 8].ctypes.data, per))
for rep in range(3):
    t0 = time.perf_counter(); stats = plan.prime_dev(dx.ptr, _ffi.P3D_C64, m.ptr, ns); t1 = time.perf_counter()
    tau = _schedule_from_stats(stats, nil * nxl, "exponential", K, 0.99, 1e-3, "values"); t2 = time.perf_counter()
    r = plan.run_dev(dx.ptr, _ffi.P3D_C64, m.ptr, tau, K, out.ptr, ns, thresh_op="hard", want_sums=False, primed=True); t3 = time.perf_counter()
    print("prime %.2f ms, schedule %.2f ms, run %.2f ms (device %.2f)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, r[2]))
