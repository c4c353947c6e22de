"""ctypes binding of include/p3d.h (libp3d_hip.so).  No fallback: if the library is missing or no
GPU is visible, every compute entry point raises."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P3D_LIB_PATH") or os.path.join(_HERE, "libp3d_hip.so")  # env override: experiments only

P3D_OK = 0
P3D_ERR_INVALID, P3D_ERR_UNSUPPORTED, P3D_ERR_HIP = -1, -2, -3
P3D_C64, P3D_F32 = 0, 1
P3D_C128, P3D_F64 = 2, 3      # the double-precision entry points only (Plan64)
P3D_OP = {"hard": 0, "soft": 1, "garrote": 2, "garotte": 2}
P3D_OP_PERCENTILE = 16
P3D_OP.update({f"{k}-percentile": v | P3D_OP_PERCENTILE for k, v in list(P3D_OP.items())})
P3D_VER = {"regular": 0, "fast": 1, "adaptive": 2}
P3D_FLAG_PROFILE = 1
P3D_FLAG_PRIMED = 2
STATS_PER_SLICE = 6


class P3DError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libp3d_hip error {code}: {msg}")
        self.code = code


class UnsupportedError(P3DError, NotImplementedError):
    pass


class PocsParams(C.Structure):
    _fields_ = [("niter", C.c_int32), ("thresh_op", C.c_int32), ("version", C.c_int32), ("flags", C.c_int32),
                ("eps", C.c_double), ("alpha", C.c_double)]


_lib = None

# name -> (restype, argtypes); kept in one table so tests can check it against include/p3d.h
PROTOTYPES = {
    "p3d_abi_version": (C.c_int, []),
    "p3d_last_error": (C.c_char_p, []),
    "p3d_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "p3d_shape_supported": (C.c_int, [C.c_int, C.c_int]),
    "p3d_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "p3d_plan_destroy": (C.c_int, [C.c_void_p]),
    "p3d_malloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "p3d_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "p3d_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "p3d_host_free": (C.c_int, [C.c_void_p]),
    "p3d_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "p3d_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "p3d_fft2_c64_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "p3d_fft2_c64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "p3d_fft2_shrink_c64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "p3d_pocs_stats_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_runtime_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "p3d_plan64_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "p3d_plan64_destroy": (C.c_int, [C.c_void_p]),
    "p3d_pocs64_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_fft2_c128": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "p3d_pocs64_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PocsParams), C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "p3d_dev_malloc": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.c_size_t]),
    "p3d_dev_free": (C.c_int, [C.c_void_p]),
    "p3d_dev_memcpy": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "p3d_dev_memset": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_size_t]),
    "p3d_dev_synchronize": (C.c_int, [C.c_int]),
    "p3d_dev_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "p3d_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "p3d_host_unregister": (C.c_int, [C.c_void_p]),
    "p3d_pocs_prime_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "p3d_pocs_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_pocs_sorted_spectrum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "p3d_pocs_data_driven_pick": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "p3d_pocs_run_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(PocsParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_double)]),
    "p3d_pocs_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(PocsParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_double)]),
    "p3d_multi_stats": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_multi_run": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(PocsParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "p3d_time2freq": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p]),
    "p3d_freq2time": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int,
                                C.c_void_p]),
    "p3d_time2freq_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "p3d_freq2time_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.c_void_p]),
    "p3d_last_sparsity": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "p3d_smooth_gaussian": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "p3d_smooth_median": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "p3d_last_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int)]),
    "p3d_wavelet_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "p3d_wavelet_plan_destroy": (C.c_int, [C.c_void_p]),
    "p3d_wavelet_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_void_p]),
    "p3d_wavedec2_c64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "p3d_waverec2_c64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "p3d_wavelet_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_wavelet_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PocsParams),
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "p3d_wavelet64_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "p3d_wavelet64_plan_destroy": (C.c_int, [C.c_void_p]),
    "p3d_wavelet64_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "p3d_wavelet64_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_wavelet64_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PocsParams),
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "p3d_shearlet_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "p3d_shearlet64_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "p3d_shearlet64_plan_destroy": (C.c_int, [C.c_void_p]),
    "p3d_shearlet64_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "p3d_shearlet64_fused_shape": (C.c_int, [C.c_int, C.c_int]),
    "p3d_shearlet64_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_shearlet64_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PocsParams),
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "p3d_shearlet_plan_destroy": (C.c_int, [C.c_void_p]),
    "p3d_shearlet_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "p3d_shearlet_transform_c64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "p3d_shearlet_inverse_c64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "p3d_shearlet_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "p3d_shearlet_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PocsParams),
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
}


def _preload_torch_hip():
    """PyTorch-ROCm wheels bring their own HIP runtime (same SONAME as /opt/rocm's).  The library is built against the SYSTEM runtime and, by
    default, runs on it: nothing is preloaded.  A process that ALSO uses ``torch.cuda`` must let torch go first -- once this library's
    runtime has initialised, a later ``torch.cuda`` call finds no device ("No HIP GPUs are available", measured on the MI355X boxes), the
    other order works: ``import torch`` before the first call into this package (``sharding.py`` and ``bench.py --gpus N>1`` do), or set
    ``P3D_TORCH_HIP_PRELOAD=1`` to have torch's runtime library dlopen'ed ahead of ours without importing torch.  In both cases the
    library then runs on torch's runtime; `_check_runtime` records the pair of versions and warns when they differ."""
    if not os.environ.get("P3D_TORCH_HIP_PRELOAD") or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
        if path and os.path.isfile(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def lib():
    """Load libp3d_hip.so once.  Raises ``ImportError`` if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        _preload_torch_hip()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        _check_runtime(handle)
    return _lib


_runtime = None


def _check_runtime(handle):
    """Which HIP runtime did the process bind?  libp3d_hip.so is built against /opt/rocm; when a PyTorch wheel's copy of libamdhip64 (same
    SONAME) is mapped first -- `_preload_torch_hip` does that on purpose, see there -- the library runs on THAT runtime.  Record both
    versions and the file, and warn when major.minor differ (an unchecked ABI / code-object skew otherwise)."""
    global _runtime
    comp, run = C.c_int(0), C.c_int(0)
    path = None
    try:
        if handle.p3d_runtime_info(C.byref(comp), C.byref(run)) != P3D_OK:
            return
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass

    def split(v):
        return v // 10000000, (v // 100000) % 100, v % 100000
    _runtime = {"compiled_hip_version": "%d.%d.%d" % split(comp.value), "runtime_hip_version": "%d.%d.%d" % split(run.value), "runtime_library": path}
    if split(comp.value)[:2] != split(run.value)[:2]:
        import warnings
        warnings.warn(f"libp3d_hip.so was compiled against HIP {_runtime['compiled_hip_version']} but the process runs HIP "
                      f"{_runtime['runtime_hip_version']} ({path}): another copy of the runtime was mapped first (PyTorch imported before this package, or "
                      f"P3D_TORCH_HIP_PRELOAD=1); processes that do not need torch.cuda should load this package first", RuntimeWarning, stacklevel=3)


def runtime_info():
    """{'compiled_hip_version', 'runtime_hip_version', 'runtime_library'} of the loaded library (bench.py records it)."""
    lib()
    return dict(_runtime or {})


def check(code):
    if code != P3D_OK:
        msg = lib().p3d_last_error().decode("utf-8", "replace")
        raise (UnsupportedError if code == P3D_ERR_UNSUPPORTED else P3DError)(code, msg)


def device_count():
    n = C.c_int(0)
    check(lib().p3d_device_count(C.byref(n)))
    return n.value


def shape_supported(nil, nxl):
    return bool(lib().p3d_shape_supported(int(nil), int(nxl)))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A hipMalloc'ed buffer owned through a plan (for callers that keep cubes resident in HBM)."""

    def __init__(self, plan, nbytes):
        self.plan, self.nbytes = plan, int(nbytes)
        p = C.c_void_p()
        check(lib().p3d_malloc(plan.handle, C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().p3d_memcpy_h2d(self.plan.handle, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().p3d_memcpy_d2h(self.plan.handle, _ptr(out), self.ptr, out.nbytes))
        return out

    def download_into(self, arr):
        """Copy the first ``arr.nbytes`` bytes into a C-contiguous host array (e.g. a slab of the result cube)."""
        if not arr.flags.c_contiguous or arr.nbytes > self.nbytes:
            raise ValueError("destination must be C-contiguous and no larger than the buffer")
        check(lib().p3d_memcpy_d2h(self.plan.handle, _ptr(arr), self.ptr, arr.nbytes))
        return arr

    def free(self):
        if self.ptr:
            check(lib().p3d_free(self.plan.handle, self.ptr))
            self.ptr = None


class WaveletPlan64:
    """p3d_wplan64 wrapper: the WAVELET POCS loop in double precision (include/p3d.h) for complex128 / float64 cubes, and for complex64 /
    float32 cubes on request (``precision='reference'``)."""
    _DT = {np.dtype(np.complex128): P3D_C128, np.dtype(np.float64): P3D_F64, np.dtype(np.complex64): P3D_C64, np.dtype(np.float32): P3D_F32}

    def __init__(self, nil, nxl, max_slices, wavelet="coif5", level=None, device=0):
        self.nil, self.nxl, self.max_slices, self.device = int(nil), int(nxl), int(max_slices), int(device)
        self.wavelet = wavelet
        bank = wavelet if isinstance(wavelet, (tuple, list)) else wavelet_filters(wavelet)
        bank = [np.ascontiguousarray(b, dtype=np.float64) for b in bank]
        if len(bank) != 4 or len({b.size for b in bank}) != 1:
            raise ValueError("a filter bank is (dec_lo, dec_hi, rec_lo, rec_hi) of equal length")
        h = C.c_void_p()
        check(lib().p3d_wavelet64_plan_create(C.byref(h), self.device, self.nil, self.nxl, self.max_slices, *map(_ptr, bank),
                                              bank[0].size, -1 if level is None else int(level)))
        self.handle = h
        nlev, ncoef = C.c_int(0), C.c_int64(0)
        check(lib().p3d_wavelet64_info(self.handle, C.byref(nlev), C.byref(ncoef)))
        self.nlev, self.ncoef = nlev.value, ncoef.value

    def close(self):
        if getattr(self, "handle", None):
            lib().p3d_wavelet64_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if lib is not None:
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _cube(self, x):
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.nil, self.nxl) or x.shape[0] > self.max_slices:
            raise ValueError(f"expected (<= {self.max_slices}, {self.nil}, {self.nxl}), got {x.shape}")
        if x.dtype not in self._DT:
            x = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
        return np.ascontiguousarray(x), self._DT[x.dtype]

    @staticmethod
    def _tau(tau, n, niter, nlev):
        tau = np.broadcast_to(np.asarray(tau), (n, niter, nlev, 3))
        t = np.empty(tau.shape + (2,), np.float64)
        t[..., 0] = tau.real
        t[..., 1] = tau.imag if np.iscomplexobj(tau) else 0.0
        return t

    def stats(self, x):
        """(nslices, nlev, 3, 4): Re / Im of the lexicographic max, max |d|, min |d| per detail array (coarsest level first), in double."""
        xc, dt = self._cube(x)
        return self.stats_dev(xc.ctypes.data, dt, xc.shape[0])

    def stats_dev(self, x_ptr, dtype, n):
        st = np.empty((n, self.nlev, 3, 4), np.float64)
        check(lib().p3d_wavelet64_stats(self.handle, C.c_void_p(x_ptr), dtype, n, _ptr(st)))
        return st

    def run(self, x, mask, tau, niter, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """Host arrays in, host array out (dtype of ``x``).  tau: (nslices, niter, nlev, 3) real or complex.  Returns (out, niter_done, sums, ms)."""
        xc, dt = self._cube(x)
        m = np.ascontiguousarray(mask, dtype=np.float64)
        if m.shape != (self.nil, self.nxl):
            raise ValueError(f"mask shape {m.shape} != {(self.nil, self.nxl)}")
        out = np.empty_like(xc)
        done, sums, ms = self.run_dev(xc.ctypes.data, dt, m.ctypes.data, tau, niter, out.ctypes.data, xc.shape[0], thresh_op=thresh_op, version=version,
                                      eps=eps, alpha=alpha, active=active)
        return out, done, sums, ms

    def run_dev(self, x_ptr, dtype, mask_ptr, tau, niter, out_ptr, n, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """`run` on raw pointers (host or device; the mask is DOUBLE [nil][nxl]).  Returns (niter_done, sums, device ms of the loop)."""
        t = self._tau(tau, n, niter, self.nlev)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_wavelet64_run(self.handle, C.c_void_p(x_ptr), dtype, C.c_void_p(mask_ptr), _ptr(t), None if act is None else _ptr(act),
                                      C.byref(prm), C.c_void_p(out_ptr), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return done, sums, ms.value


class DeviceArray:
    """A typed block of device memory that belongs to no plan (``p3d_dev_malloc``): cubes that stay resident in HBM across several plans
    and jobs, without any other GPU runtime in the process.  ``ptr`` is what the ``*_dev`` entry points take."""

    def __init__(self, shape, dtype, device=0):
        self.shape, self.dtype, self.device = tuple(int(n) for n in shape), np.dtype(dtype), int(device)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().p3d_dev_malloc(self.device, C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    def _span(self, first, count):
        per = self.nbytes // self.shape[0] if self.shape[0] else 0
        count = self.shape[0] - first if count is None else count
        if first < 0 or count < 0 or first + count > self.shape[0]:
            raise ValueError(f'rows {first}:{first + count} outside an array of {self.shape[0]}')
        return self.ptr + first * per, count * per, count

    def upload(self, host, first=0):
        """rows ``first ...`` <- ``host`` (C-contiguous, this dtype, trailing shape of the array)."""
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.shape[1:] != self.shape[1:]:
            raise ValueError(f'trailing shape {host.shape[1:]} does not match {self.shape[1:]}')
        dst, nbytes, _ = self._span(first, host.shape[0])
        check(lib().p3d_dev_memcpy(self.device, dst, _ptr(host), nbytes, 0))
        return self

    def download(self, first=0, count=None, out=None):
        src, nbytes, count = self._span(first, count)
        if out is None:
            out = np.empty((count,) + self.shape[1:], self.dtype)
        elif not out.flags.c_contiguous or out.nbytes != nbytes or out.dtype != self.dtype:
            raise ValueError('out must be C-contiguous with the dtype and size of the rows asked for')
        check(lib().p3d_dev_memcpy(self.device, _ptr(out), src, nbytes, 1))
        return out

    def copy_from(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError('size mismatch')
        check(lib().p3d_dev_memcpy(self.device, self.ptr, other.ptr, self.nbytes, 2))
        return self

    def zero(self):
        check(lib().p3d_dev_memset(self.device, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self.ptr:
            check(lib().p3d_dev_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass


def device_synchronize(device=0):
    check(lib().p3d_dev_synchronize(int(device)))


def device_mem_info(device=0):
    free, total = C.c_size_t(0), C.c_size_t(0)
    check(lib().p3d_dev_mem_info(int(device), C.byref(free), C.byref(total)))
    return free.value, total.value


def host_register(arr):
    """Page-lock a C-contiguous NumPy array in place (p3d_host_register).  True when it is registered now (and must be handed to
    :func:`host_unregister` later), False when the runtime refused -- the array is usable either way."""
    if not arr.flags.c_contiguous or arr.nbytes == 0:
        return False
    return lib().p3d_host_register(_ptr(arr), arr.nbytes) == P3D_OK


def host_unregister(arr):
    check(lib().p3d_host_unregister(_ptr(arr)))


def host_register_range(addr, nbytes):
    """Page-lock ``nbytes`` of host memory from address ``addr`` (the caller keeps the memory alive); True when registered."""
    return nbytes > 0 and lib().p3d_host_register(C.c_void_p(addr), nbytes) == P3D_OK


def host_unregister_range(addr):
    check(lib().p3d_host_unregister(C.c_void_p(addr)))


class PinnedBuffer:
    """Page-locked host memory (p3d_host_alloc) viewed as NumPy arrays: the staging area of the chunk pipeline."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().p3d_host_alloc(C.byref(p), self.nbytes))
        self.ptr = p.value
        self._raw = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr))

    def view(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if n > self.nbytes:
            raise ValueError("view larger than the buffer")
        return self._raw[:n].view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self._raw = None
            check(lib().p3d_host_free(self.ptr))
            self.ptr = None


class Plan:
    """p3d_plan wrapper: one device, one (nil, nxl) slice shape, up to ``max_slices`` per call."""

    def __init__(self, nil, nxl, max_slices, device=0):
        self.nil, self.nxl, self.max_slices, self.device = int(nil), int(nxl), int(max_slices), int(device)
        h = C.c_void_p()
        check(lib().p3d_plan_create(C.byref(h), self.device, self.nil, self.nxl, self.max_slices))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            lib().p3d_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if lib is not None:   # (module globals are gone while the interpreter shuts down: the process's device memory goes with it)
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- helpers ---------------------------------------------------------------------------
    def _cube(self, x):
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.nil, self.nxl):
            raise ValueError(f"expected (nslices, {self.nil}, {self.nxl}), got {x.shape}")
        if x.shape[0] > self.max_slices:
            raise ValueError(f"{x.shape[0]} slices > max_slices {self.max_slices}")
        if np.iscomplexobj(x):
            return np.ascontiguousarray(x, dtype=np.complex64), P3D_C64
        return np.ascontiguousarray(x, dtype=np.float32), P3D_F32

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # ---- transforms ------------------------------------------------------------------------
    def fft2(self, x, inverse=False):
        x = np.asarray(x)
        squeeze = x.ndim == 2
        xc, _ = self._cube(x.astype(np.complex64, copy=False))
        out = np.empty_like(xc)
        check(lib().p3d_fft2_c64(self.handle, _ptr(xc), _ptr(out), xc.shape[0], int(bool(inverse))))
        return out[0] if squeeze else out

    def fft2_dev(self, in_ptr, out_ptr, nslices, inverse=False):
        """fft2 / ifft2 (numpy conventions) of complex64 slices resident on the device; `out_ptr` may equal `in_ptr`."""
        check(lib().p3d_fft2_c64_dev(self.handle, C.c_void_p(in_ptr), C.c_void_p(out_ptr), int(nslices), int(bool(inverse))))

    def fft2_shrink(self, x, tau, thresh_op="hard"):
        """threshold(fft2(x), tau, kind) on the device; ``tau`` scalar or one value per slice."""
        x = np.asarray(x)
        squeeze = x.ndim == 2
        xc, _ = self._cube(x.astype(np.complex64, copy=False))
        n = xc.shape[0]
        t = np.empty((n, 2), np.float64)
        tau = np.broadcast_to(np.asarray(tau, dtype=np.complex128), (n,))
        t[:, 0], t[:, 1] = tau.real, tau.imag
        out = np.empty_like(xc)
        check(lib().p3d_fft2_shrink_c64(self.handle, _ptr(xc), _ptr(t), P3D_OP[thresh_op], _ptr(out), n))
        return out[0] if squeeze else out

    # ---- POCS ------------------------------------------------------------------------------
    def stats(self, x):
        xc, dt = self._cube(x)
        st = np.empty((xc.shape[0], STATS_PER_SLICE), np.float64)
        check(lib().p3d_pocs_stats(self.handle, _ptr(xc), dt, xc.shape[0], _ptr(st)))
        return st

    def sorted_spectrum(self, x):
        """fft2 of every slice, sorted on the device in NumPy's complex order (kept in the plan); returns x_fwd.max() per slice
        (complex64) -- first half of the 'data-driven' schedule (POCS.py:356-362)."""
        xc = np.ascontiguousarray(x, dtype=np.complex64)
        if xc.ndim != 3 or xc.shape[1:] != (self.nil, self.nxl):
            raise ValueError(f"cube shape {xc.shape} != (n, {self.nil}, {self.nxl})")
        peaks = np.empty((xc.shape[0], 2), np.float32)
        check(lib().p3d_pocs_sorted_spectrum(self.handle, _ptr(xc), xc.shape[0], _ptr(peaks)))
        return peaks.view(np.complex64)[:, 0]

    def sorted_spectrum_dev(self, x_ptr, n):
        """`sorted_spectrum` of a complex64 batch resident on the device (raw pointer)."""
        peaks = np.empty((n, 2), np.float32)
        check(lib().p3d_pocs_sorted_spectrum(self.handle, C.c_void_p(x_ptr), n, _ptr(peaks)))
        return peaks.view(np.complex64)[:, 0]

    def data_driven_pick(self, tau_min, tau_max, niter):
        """Second half: per slice the number of coefficients strictly between the bounds (complex64, NumPy's order) and the
        niter thresholds picked from them (POCS.py:359-362); must follow sorted_spectrum directly."""
        lo = np.ascontiguousarray(tau_min, dtype=np.complex64)
        hi = np.ascontiguousarray(tau_max, dtype=np.complex64)
        n = lo.shape[0]
        bounds = np.empty((n, 4), np.float32)
        bounds[:, 0], bounds[:, 1], bounds[:, 2], bounds[:, 3] = lo.real, lo.imag, hi.real, hi.imag
        tau = np.empty((n, int(niter), 2), np.float32)
        count = np.empty((n,), np.int64)
        check(lib().p3d_pocs_data_driven_pick(self.handle, n, int(niter), _ptr(bounds), _ptr(tau), _ptr(count)))
        return tau.view(np.complex64)[..., 0], count

    def stats_dev(self, x_ptr, dtype, nslices):
        st = np.empty((nslices, STATS_PER_SLICE), np.float64)
        check(lib().p3d_pocs_stats_dev(self.handle, x_ptr, dtype, nslices, _ptr(st)))
        return st

    @staticmethod
    def _params(niter, thresh_op, version, eps, alpha, profile, primed=False):
        if thresh_op not in P3D_OP:
            raise UnsupportedError(P3D_ERR_UNSUPPORTED, f"thresh_op {thresh_op!r} is not implemented by the HIP kernels")
        return PocsParams(int(niter), P3D_OP[thresh_op], P3D_VER[version],
                          (P3D_FLAG_PROFILE if profile else 0) | (P3D_FLAG_PRIMED if primed else 0), float(eps), float(alpha))

    @staticmethod
    def _tau(tau, nslices, niter):
        tau = np.asarray(tau)
        t = np.empty((nslices, niter, 2), np.float64)
        t[..., 0] = np.broadcast_to(tau.real, (nslices, niter))
        t[..., 1] = np.broadcast_to(tau.imag, (nslices, niter)) if np.iscomplexobj(tau) else 0.0
        return t

    def run(self, x, mask, tau, niter, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None,
            profile=False):
        """Host arrays in, host arrays out.  Returns (out, niter_done, sums, elapsed_ms)."""
        xc, dt = self._cube(x)
        n = xc.shape[0]
        m = np.ascontiguousarray(mask, dtype=np.float32)
        if m.shape != (self.nil, self.nxl):
            raise ValueError(f"mask shape {m.shape} != {(self.nil, self.nxl)}")
        t = self._tau(tau, n, niter)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = self._params(niter, thresh_op, version, eps, alpha, profile)
        out = np.empty_like(xc)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_pocs_run(self.handle, _ptr(xc), dt, _ptr(m), _ptr(t), None if act is None else _ptr(act),
                                 C.byref(prm), _ptr(out), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return out, done, sums, ms.value

    def prime_dev(self, x_ptr, dtype, mask_ptr, nslices):
        """`stats_dev` that doubles as the first pass of the job: follow it with ``run_dev(..., primed=True)`` on the same
        pointers and batch (include/p3d.h, p3d_pocs_prime_dev).

        ``primed=True`` is a promise about the CONTENTS of the two device buffers: the plan can only check that pointers, dtype and
        batch are the ones it primed and that nothing else ran on it in between -- a caller that rewrites ``x`` or ``mask`` in
        place between the two calls must not pass the flag (the run would use the work buffer, compact samples and ``sum |x_obs|``
        of the old contents)."""
        st = np.empty((nslices, STATS_PER_SLICE), np.float64)
        check(lib().p3d_pocs_prime_dev(self.handle, x_ptr, dtype, mask_ptr, nslices, _ptr(st)))
        return st

    def run_dev(self, x_ptr, dtype, mask_ptr, tau, niter, out_ptr, nslices, thresh_op="hard", version="regular",
                eps=0.0, alpha=1.0, active=None, profile=False, want_sums=True, primed=False):
        """Device pointers in/out (cube stays resident in HBM).  Returns (niter_done, sums, elapsed_ms)."""
        t = self._tau(tau, nslices, niter)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = self._params(niter, thresh_op, version, eps, alpha, profile, primed)
        done = np.zeros(nslices, np.int32)
        sums = np.zeros((niter + 1, nslices), np.float64) if want_sums else None
        ms = C.c_double(0.0)
        check(lib().p3d_pocs_run_dev(self.handle, x_ptr, dtype, mask_ptr, _ptr(t), None if act is None else _ptr(act),
                                     C.byref(prm), out_ptr, nslices, _ptr(done), None if sums is None else _ptr(sums),
                                     C.byref(ms)))
        return done, sums, ms.value

    def last_sparsity(self):
        """Fraction of 8-column spectrum blocks that kept a coefficient in the last run (-1: dense path)."""
        v = C.c_double(-1.0)
        check(lib().p3d_last_sparsity(self.handle, C.byref(v)))
        return v.value

    def last_profile(self):
        cm, rm, cn, rn = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        check(lib().p3d_last_profile(self.handle, C.byref(cm), C.byref(cn), C.byref(rm), C.byref(rn)))
        return {"colpass_ms": cm.value, "colpass_launches": cn.value, "rowpass_ms": rm.value,
                "rowpass_launches": rn.value}


# ---- steps 12 / 14: time <-> frequency along the slice axis ------------------------------------------------

# ---- WAVELET variant -----------------------------------------------------------------------
_BANKS = None


def wavelet_filters(name):
    """(dec_lo, dec_hi, rec_lo, rec_hi) of a PyWavelets wavelet name (data file wavelets.json; pywt is not required)."""
    global _BANKS
    if _BANKS is None:
        import json
        with open(os.path.join(_HERE, "wavelets.json")) as f:
            _BANKS = json.load(f)["wavelets"]
    try:
        b = _BANKS[str(name)]
    except KeyError:
        raise ValueError(f"Unknown wavelet name {name!r}, check wavelets.json for the list of available builtin wavelets.") from None
    return tuple(np.ascontiguousarray(b[k], dtype=np.float64) for k in ("dec_lo", "dec_hi", "rec_lo", "rec_hi"))


class WaveletPlan:
    """p3d_wplan wrapper: multilevel 2-D DWT ('smooth' extension) of (nil, nxl) slices and the WAVELET POCS loop."""

    def __init__(self, nil, nxl, max_slices, wavelet="coif5", level=None, device=0):
        self.nil, self.nxl, self.max_slices, self.device = int(nil), int(nxl), int(max_slices), int(device)
        self.wavelet = wavelet
        bank = wavelet if isinstance(wavelet, (tuple, list)) else wavelet_filters(wavelet)
        bank = [np.ascontiguousarray(b, dtype=np.float64) for b in bank]
        if len(bank) != 4 or len({b.size for b in bank}) != 1:
            raise ValueError("a filter bank is (dec_lo, dec_hi, rec_lo, rec_hi) of equal length")
        h = C.c_void_p()
        check(lib().p3d_wavelet_plan_create(C.byref(h), self.device, self.nil, self.nxl, self.max_slices, *map(_ptr, bank),
                                            bank[0].size, -1 if level is None else int(level)))
        self.handle = h
        nlev, ncoef = C.c_int(0), C.c_int64(0)
        check(lib().p3d_wavelet_info(self.handle, C.byref(nlev), C.byref(ncoef), None))
        self.nlev, self.ncoef = nlev.value, ncoef.value
        shapes = np.zeros((self.nlev + 1, 2), np.int32)
        check(lib().p3d_wavelet_info(self.handle, None, None, _ptr(shapes)))
        self.shapes = [tuple(int(v) for v in r) for r in shapes]   # cA, then details coarsest -> finest

    def close(self):
        if getattr(self, "handle", None):
            lib().p3d_wavelet_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if lib is not None:   # (module globals are gone while the interpreter shuts down: the process's device memory goes with it)
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    _cube = Plan._cube

    def unpack(self, vec):
        """flat coefficient vector of ONE slice -> [cA, (cH, cV, cD), ...] like pywt.wavedec2."""
        r, c = self.shapes[0]
        out, off = [vec[:r * c].reshape(r, c)], r * c
        for r, c in self.shapes[1:]:
            det = []
            for _ in range(3):
                det.append(vec[off:off + r * c].reshape(r, c))
                off += r * c
            out.append(tuple(det))
        return out

    def pack(self, coeffs):
        return np.concatenate([np.ravel(coeffs[0])] + [np.ravel(d) for lvl in coeffs[1:] for d in lvl]).astype(np.complex64)

    def wavedec2(self, x):
        x = np.asarray(x)
        squeeze = x.ndim == 2
        xc, _ = self._cube(x.astype(np.complex64, copy=False))
        coef = np.empty((xc.shape[0], self.ncoef), np.complex64)
        check(lib().p3d_wavedec2_c64(self.handle, _ptr(xc), _ptr(coef), xc.shape[0]))
        return coef[0] if squeeze else coef

    def waverec2(self, coef):
        coef = np.ascontiguousarray(coef, dtype=np.complex64)
        squeeze = coef.ndim == 1
        coef = coef.reshape(-1, self.ncoef)
        if coef.shape[0] > self.max_slices:
            raise ValueError(f"{coef.shape[0]} slices > max_slices {self.max_slices}")
        out = np.empty((coef.shape[0], self.nil, self.nxl), np.complex64)
        check(lib().p3d_waverec2_c64(self.handle, _ptr(coef), _ptr(out), coef.shape[0]))
        return out[0] if squeeze else out

    def stats(self, x):
        """(nslices, nlev, 3, 4): Re/Im of the lexicographic max, max |d|, min |d| per detail array (coarsest level first)."""
        xc, dt = self._cube(x)
        st = np.empty((xc.shape[0], self.nlev, 3, 4), np.float64)
        check(lib().p3d_wavelet_stats(self.handle, _ptr(xc), dt, xc.shape[0], _ptr(st)))
        return st

    def run(self, x, mask, tau, niter, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """tau: (nslices, niter, nlev, 3) real or complex.  Returns (out, niter_done, sums, elapsed_ms)."""
        xc, dt = self._cube(x)
        n = xc.shape[0]
        m = np.ascontiguousarray(mask, dtype=np.float32)
        if m.shape != (self.nil, self.nxl):
            raise ValueError(f"mask shape {m.shape} != {(self.nil, self.nxl)}")
        tau = np.broadcast_to(np.asarray(tau), (n, niter, self.nlev, 3))
        t = np.empty(tau.shape + (2,), np.float64)
        t[..., 0] = tau.real
        t[..., 1] = tau.imag if np.iscomplexobj(tau) else 0.0
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        out = np.empty_like(xc)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_wavelet_run(self.handle, _ptr(xc), dt, _ptr(m), _ptr(t), None if act is None else _ptr(act),
                                    C.byref(prm), _ptr(out), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return out, done, sums, ms.value

    def stats_dev(self, x_ptr, dtype, n):
        """`stats` for a cube resident on the device (raw pointer, P3D_C64 / P3D_F32)."""
        st = np.empty((n, self.nlev, 3, 4), np.float64)
        check(lib().p3d_wavelet_stats(self.handle, C.c_void_p(x_ptr), dtype, n, _ptr(st)))
        return st

    def run_dev(self, x_ptr, dtype, mask_ptr, tau, niter, out_ptr, n, thresh_op="hard", version="regular", eps=0.0, alpha=1.0,
                active=None):
        """`run` on device-resident buffers (raw pointers; the library accepts host or device pointers for the cubes).  Returns
        (niter_done, sums, device ms of the loop)."""
        tau = np.broadcast_to(np.asarray(tau), (n, niter, self.nlev, 3))
        t = np.empty(tau.shape + (2,), np.float64)
        t[..., 0] = tau.real
        t[..., 1] = tau.imag if np.iscomplexobj(tau) else 0.0
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_wavelet_run(self.handle, C.c_void_p(x_ptr), dtype, C.c_void_p(mask_ptr), _ptr(t), None if act is None else _ptr(act),
                                    C.byref(prm), C.c_void_p(out_ptr), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return done, sums, ms.value


# ---- the loop in the reference's precision ---------------------------------------------------
class Plan64:
    """p3d_plan64 wrapper: the FFT POCS loop in double precision (include/p3d.h) for complex128 / float64 cubes, and for complex64 /
    float32 cubes whose reference run is a double-precision one (soft / garrote / FPOCS / APOCS, or any run under NumPy < 2)."""
    _DT = {np.dtype(np.complex128): P3D_C128, np.dtype(np.float64): P3D_F64, np.dtype(np.complex64): P3D_C64, np.dtype(np.float32): P3D_F32}

    def __init__(self, nil, nxl, max_slices, device=0):
        self.nil, self.nxl, self.max_slices, self.device = int(nil), int(nxl), int(max_slices), int(device)
        h = C.c_void_p()
        check(lib().p3d_plan64_create(C.byref(h), self.device, self.nil, self.nxl, self.max_slices))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            lib().p3d_plan64_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if lib is not None:
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _cube(self, x):
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.nil, self.nxl) or x.shape[0] > self.max_slices:
            raise ValueError(f"expected (<= {self.max_slices}, {self.nil}, {self.nxl}), got {x.shape}")
        if x.dtype not in self._DT:
            x = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
        return np.ascontiguousarray(x), self._DT[x.dtype]

    def fft2(self, x, inverse=False):
        """Test hook: fft2 / ifft2 of complex128 slices through the loop's own passes (``p3d_fft2_c128``)."""
        xc = np.ascontiguousarray(np.asarray(x, dtype=np.complex128))
        squeeze = xc.ndim == 2
        if squeeze:
            xc = xc[None]
        if xc.shape[1:] != (self.nil, self.nxl) or xc.shape[0] > self.max_slices:
            raise ValueError(f"expected (<= {self.max_slices}, {self.nil}, {self.nxl}), got {xc.shape}")
        out = np.empty_like(xc)
        check(lib().p3d_fft2_c128(self.handle, _ptr(xc), _ptr(out), xc.shape[0], 1 if inverse else 0))
        return out[0] if squeeze else out

    def stats(self, x):
        """(nslices, 6) float64, the layout of ``Plan.stats``: statistics of the double-precision ``fft2(x)``."""
        xc, dt = self._cube(x)
        st = np.empty((xc.shape[0], 6), np.float64)
        check(lib().p3d_pocs64_stats(self.handle, _ptr(xc), dt, xc.shape[0], _ptr(st)))
        return st

    def run(self, x, mask, tau, niter, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """Host arrays in, host arrays out (dtype of ``x``).  Returns (out, niter_done, sums, elapsed_ms)."""
        xc, dt = self._cube(x)
        n = xc.shape[0]
        m = np.ascontiguousarray(mask, dtype=np.float64)
        if m.shape != (self.nil, self.nxl):
            raise ValueError(f"mask shape {m.shape} != {(self.nil, self.nxl)}")
        t = Plan._tau(tau, n, niter)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        out = np.empty_like(xc)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_pocs64_run(self.handle, _ptr(xc), dt, _ptr(m), _ptr(t), None if act is None else _ptr(act), C.byref(prm), _ptr(out), n,
                                   _ptr(done), _ptr(sums), C.byref(ms)))
        return out, done, sums, ms.value

    def stats_dev(self, x_ptr, dtype, n):
        """`stats` for a cube resident on the device (raw pointer; dtype P3D_C128 / P3D_F64 / P3D_C64 / P3D_F32)."""
        st = np.empty((n, 6), np.float64)
        check(lib().p3d_pocs64_stats(self.handle, C.c_void_p(x_ptr), dtype, n, _ptr(st)))
        return st

    def run_dev(self, x_ptr, dtype, mask_ptr, tau, niter, out_ptr, n, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """`run` on device-resident buffers (raw pointers; the mask is DOUBLE [nil][nxl]).  Returns (niter_done, sums, device ms of the loop)."""
        t = Plan._tau(tau, n, niter)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_pocs64_run(self.handle, C.c_void_p(x_ptr), dtype, C.c_void_p(mask_ptr), _ptr(t), None if act is None else _ptr(act), C.byref(prm),
                                   C.c_void_p(out_ptr), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return done, sums, ms.value


# ---- SHEARLET variant ----------------------------------------------------------------------
class ShearletPlan:
    """p3d_splan wrapper: frequency-domain shearlet frame with caller-supplied spectra ``psi`` (nil, nxl, nsh) -- the layout of
    ``FFST.scalesShearsAndSpectra`` -- and the SHEARLET POCS loop for up to ``max_slices`` slices per call."""

    def __init__(self, psi, max_slices=1, device=0):
        psi = np.asarray(psi)
        if psi.ndim != 3:
            raise ValueError(f"Psi must be (nil, nxl, nshearlets), got shape {psi.shape}")
        if np.iscomplexobj(psi):
            raise NotImplementedError("complex shearlet spectra (realCoefficients=False) are not implemented")
        self.nil, self.nxl, self.nsh = (int(v) for v in psi.shape)
        self.max_slices, self.device = int(max_slices), int(device)
        dev_psi = np.ascontiguousarray(np.moveaxis(psi, -1, 0), dtype=np.float32)
        h = C.c_void_p()
        check(lib().p3d_shearlet_plan_create(C.byref(h), self.device, self.nil, self.nxl, self.nsh, _ptr(dev_psi), self.max_slices))
        self.handle = h
        frac, paired = C.c_double(1.0), C.c_int(0)
        check(lib().p3d_shearlet_info(self.handle, C.byref(frac), C.byref(paired)))
        # share of the (shearlet, 8-row group) pairs on which the spectrum does not vanish: what the fused passes actually touch;
        # float32 cubes on symmetric spectra additionally work on Hermitian half slices, two columns per transform
        self.row_group_fraction, self.paired = frac.value, bool(paired.value)

    def close(self):
        if getattr(self, "handle", None):
            lib().p3d_shearlet_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if lib is not None:   # (module globals are gone while the interpreter shuts down: the process's device memory goes with it)
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    _cube = Plan._cube

    def transform(self, x):
        """(…, nil, nxl) -> (…, nil, nxl, nsh) complex64 (the reference's layout: shearlets on the last axis)."""
        x = np.asarray(x)
        squeeze = x.ndim == 2
        xc, _ = self._cube(x.astype(np.complex64, copy=False))
        st = np.empty((xc.shape[0], self.nsh, self.nil, self.nxl), np.complex64)
        check(lib().p3d_shearlet_transform_c64(self.handle, _ptr(xc), _ptr(st), xc.shape[0]))
        st = np.moveaxis(st, 1, -1)
        return st[0] if squeeze else st

    def inverse(self, st):
        st = np.asarray(st)
        squeeze = st.ndim == 3
        if squeeze:
            st = st[None]
        if st.shape[1:] != (self.nil, self.nxl, self.nsh) or st.shape[0] > self.max_slices:
            raise ValueError(f"expected (<= {self.max_slices}, {self.nil}, {self.nxl}, {self.nsh}), got {st.shape}")
        dev = np.ascontiguousarray(np.moveaxis(st, -1, 1), dtype=np.complex64)
        out = np.empty((st.shape[0], self.nil, self.nxl), np.complex64)
        check(lib().p3d_shearlet_inverse_c64(self.handle, _ptr(dev), _ptr(out), st.shape[0]))
        return out[0] if squeeze else out

    def stats(self, x):
        """(nslices, nsh, 5): Re/Im of the lexicographic (real cubes: signed) max, max |c|, min |c|, sum |c|^2 per shearlet."""
        xc, dt = self._cube(x)
        st = np.empty((xc.shape[0], self.nsh, 5), np.float64)
        check(lib().p3d_shearlet_stats(self.handle, _ptr(xc), dt, xc.shape[0], _ptr(st)))
        return st

    def run(self, x, mask, tau, niter, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """tau: (nslices, niter, nsh) real or complex.  Returns (out, niter_done, sums, elapsed_ms)."""
        xc, dt = self._cube(x)
        n = xc.shape[0]
        m = np.ascontiguousarray(mask, dtype=np.float32)
        if m.shape != (self.nil, self.nxl):
            raise ValueError(f"mask shape {m.shape} != {(self.nil, self.nxl)}")
        tau = np.broadcast_to(np.asarray(tau), (n, niter, self.nsh))
        t = np.empty(tau.shape + (2,), np.float64)
        t[..., 0] = tau.real
        t[..., 1] = tau.imag if np.iscomplexobj(tau) else 0.0
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        out = np.empty_like(xc)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_shearlet_run(self.handle, _ptr(xc), dt, _ptr(m), _ptr(t), None if act is None else _ptr(act),
                                     C.byref(prm), _ptr(out), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return out, done, sums, ms.value

    def stats_dev(self, x_ptr, dtype, n):
        """`stats` for a cube resident on the device (raw pointer, P3D_C64 / P3D_F32)."""
        st = np.empty((n, self.nsh, 5), np.float64)
        check(lib().p3d_shearlet_stats(self.handle, C.c_void_p(x_ptr), dtype, n, _ptr(st)))
        return st

    def run_dev(self, x_ptr, dtype, mask_ptr, tau, niter, out_ptr, n, thresh_op="hard", version="regular", eps=0.0, alpha=1.0,
                active=None):
        """`run` on device-resident buffers (raw pointers).  Returns (niter_done, sums, device ms of the loop)."""
        tau = np.broadcast_to(np.asarray(tau), (n, niter, self.nsh))
        t = np.empty(tau.shape + (2,), np.float64)
        t[..., 0] = tau.real
        t[..., 1] = tau.imag if np.iscomplexobj(tau) else 0.0
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_shearlet_run(self.handle, C.c_void_p(x_ptr), dtype, C.c_void_p(mask_ptr), _ptr(t), None if act is None else _ptr(act),
                                     C.byref(prm), C.c_void_p(out_ptr), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return done, sums, ms.value


def shearlet64_fused_shape(nil, nxl):
    """True when the double-precision SHEARLET loop runs its fused passes for (nil, nxl) slices (both extents on the register engine)."""
    return bool(lib().p3d_shearlet64_fused_shape(int(nil), int(nxl)))


class ShearletPlan64:
    """p3d_splan64 wrapper: the SHEARLET POCS loop in double precision (include/p3d.h) for complex128 / float64 cubes, and for complex64 /
    float32 cubes on request (``precision='reference'``); ``psi`` (nil, nxl, nsh) as for :class:`ShearletPlan`, kept in double."""
    _DT = WaveletPlan64._DT

    def __init__(self, psi, max_slices=1, device=0):
        psi = np.asarray(psi)
        if psi.ndim != 3:
            raise ValueError(f"Psi must be (nil, nxl, nshearlets), got shape {psi.shape}")
        if np.iscomplexobj(psi):
            raise NotImplementedError("complex shearlet spectra (realCoefficients=False) are not implemented")
        self.nil, self.nxl, self.nsh = (int(v) for v in psi.shape)
        self.max_slices, self.device = int(max_slices), int(device)
        dev_psi = np.ascontiguousarray(np.moveaxis(psi, -1, 0), dtype=np.float64)
        h = C.c_void_p()
        check(lib().p3d_shearlet64_plan_create(C.byref(h), self.device, self.nil, self.nxl, self.nsh, _ptr(dev_psi), self.max_slices))
        self.handle = h
        fused, frac = C.c_int(0), C.c_double(1.0)
        check(lib().p3d_shearlet64_info(self.handle, C.byref(fused), C.byref(frac)))
        self.fused = bool(fused.value & 1)   # three fused passes per iteration on the double-precision register engine (both extents have a plan there)
        self.paired = bool(fused.value & 2)  # ... and real cubes on Hermitian coefficient slices, two columns per transform (symmetric spectra, even extents)
        self.row_group_fraction = frac.value   # share of the (shearlet, row group) pairs those passes touch (rows off a spectrum's support are skipped)

    def close(self):
        if getattr(self, "handle", None):
            lib().p3d_shearlet64_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if lib is not None:
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    _cube = WaveletPlan64._cube

    def stats(self, x):
        """(nslices, nsh, 5): Re / Im of the lexicographic (real cubes: signed) max, max |c|, min |c|, sum |c|^2 per shearlet, in double."""
        xc, dt = self._cube(x)
        return self.stats_dev(xc.ctypes.data, dt, xc.shape[0])

    def stats_dev(self, x_ptr, dtype, n):
        st = np.empty((n, self.nsh, 5), np.float64)
        check(lib().p3d_shearlet64_stats(self.handle, C.c_void_p(x_ptr), dtype, n, _ptr(st)))
        return st

    def run(self, x, mask, tau, niter, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """Host arrays in, host array out (dtype of ``x``).  tau: (nslices, niter, nsh) real or complex.  Returns (out, niter_done, sums, ms)."""
        xc, dt = self._cube(x)
        m = np.ascontiguousarray(mask, dtype=np.float64)
        if m.shape != (self.nil, self.nxl):
            raise ValueError(f"mask shape {m.shape} != {(self.nil, self.nxl)}")
        out = np.empty_like(xc)
        done, sums, ms = self.run_dev(xc.ctypes.data, dt, m.ctypes.data, tau, niter, out.ctypes.data, xc.shape[0], thresh_op=thresh_op, version=version,
                                      eps=eps, alpha=alpha, active=active)
        return out, done, sums, ms

    def run_dev(self, x_ptr, dtype, mask_ptr, tau, niter, out_ptr, n, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
        """`run` on raw pointers (host or device; the mask is DOUBLE [nil][nxl]).  Returns (niter_done, sums, device ms of the loop)."""
        tau = np.broadcast_to(np.asarray(tau), (n, niter, self.nsh))
        t = np.empty(tau.shape + (2,), np.float64)
        t[..., 0] = tau.real
        t[..., 1] = tau.imag if np.iscomplexobj(tau) else 0.0
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
        done = np.zeros(n, np.int32)
        sums = np.zeros((niter + 1, n), np.float64)
        ms = C.c_double(0.0)
        check(lib().p3d_shearlet64_run(self.handle, C.c_void_p(x_ptr), dtype, C.c_void_p(mask_ptr), _ptr(t), None if act is None else _ptr(act),
                                       C.byref(prm), C.c_void_p(out_ptr), n, _ptr(done), _ptr(sums), C.byref(ms)))
        return done, sums, ms.value


def time2freq(x, dt, t0=0.0, nfft=None, real_only=False, window=None, device=0):
    """(nt, ...) float32 -> (nfreq, ...) complex64 with xrft's true_phase / true_amplitude convention
    (include/p3d.h, p3d_time2freq).  Trailing axes are flattened to traces and restored."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    nt = x.shape[0]
    ntr = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    nfft = int(nfft or nt)
    nfreq = nfft // 2 + 1 if real_only else nfft
    out = np.empty((nfreq,) + x.shape[1:], np.complex64)
    win = None if window is None else np.ascontiguousarray(window, dtype=np.float32)
    if win is not None and win.shape != (nfreq,):
        raise ValueError(f"window must have {nfreq} entries")
    check(lib().p3d_time2freq(int(device), _ptr(x), nt, ntr, float(dt), float(t0), nfft, int(bool(real_only)),
                              None if win is None else _ptr(win), _ptr(out)))
    return out


def freq2time(X, dt, t0=0.0, nfft=None, real_only=False, kidx=None, device=0):
    """(nfreq, ...) complex64 -> (nfft, ...) float32: exact inverse of :func:`time2freq` (real part)."""
    X = np.ascontiguousarray(X, dtype=np.complex64)
    nfreq = X.shape[0]
    ntr = int(np.prod(X.shape[1:])) if X.ndim > 1 else 1
    if nfft is None:
        nfft = 2 * (nfreq - 1) if real_only else nfreq
    nfft = int(nfft)
    k = np.arange(nfreq, dtype=np.int32) if kidx is None else np.ascontiguousarray(kidx, dtype=np.int32)
    out = np.empty((nfft,) + X.shape[1:], np.float32)
    check(lib().p3d_freq2time(int(device), _ptr(X), nfreq, _ptr(k), ntr, float(dt), float(t0), nfft, int(bool(real_only)),
                              _ptr(out)))
    return out


def smooth_slices(x, kind, device=0, **kw):
    """scipy.ndimage.gaussian_filter / median_filter ('reflect' boundary) of every (ny, nx) slice of a float32 stack
    (include/p3d.h, p3d_smooth_gaussian / p3d_smooth_median)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 3:
        raise ValueError("expected a stack of slices (n, ny, nx)")
    out = np.empty_like(x)
    n, ny, nx = x.shape
    if kind == "gaussian":
        check(lib().p3d_smooth_gaussian(int(device), _ptr(x), n, ny, nx, float(kw["sigma"]), float(kw.get("truncate", 4.0)), _ptr(out)))
    elif kind == "median":
        check(lib().p3d_smooth_median(int(device), _ptr(x), n, ny, nx, int(kw["size"]), _ptr(out)))
    else:
        raise ValueError(f"unknown smoothing filter {kind!r}")
    return out


def _host_cube(x):
    x = np.asarray(x)
    if x.ndim != 3:
        raise ValueError("expected a cube (nslices, nil, nxl)")
    if np.iscomplexobj(x):
        return np.ascontiguousarray(x, dtype=np.complex64), P3D_C64
    return np.ascontiguousarray(x, dtype=np.float32), P3D_F32


def multi_stats(x, devices):
    """p3d_multi_stats: the statistics of every slice of a host cube, blocks of slices on the listed devices (one process)."""
    xc, dt = _host_cube(x)
    n, nil, nxl = xc.shape
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    st = np.empty((n, STATS_PER_SLICE), np.float64)
    check(lib().p3d_multi_stats(len(dev), _ptr(dev), nil, nxl, _ptr(xc), dt, n, _ptr(st)))
    return st


def multi_run(x, mask, tau, niter, devices, thresh_op="hard", version="regular", eps=0.0, alpha=1.0, active=None):
    """p3d_multi_run: the POCS loop on a host cube, blocks of slices on the listed devices (one process, one thread and one plan
    per entry).  Returns (out, niter_done, sums)."""
    xc, dt = _host_cube(x)
    n, nil, nxl = xc.shape
    m = np.ascontiguousarray(mask, dtype=np.float32)
    if m.shape != (nil, nxl):
        raise ValueError(f"mask shape {m.shape} != {(nil, nxl)}")
    t = Plan._tau(tau, n, niter)
    act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
    prm = Plan._params(niter, thresh_op, version, eps, alpha, False)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    out = np.empty_like(xc)
    done = np.zeros(n, np.int32)
    sums = np.zeros((niter + 1, n), np.float64)
    check(lib().p3d_multi_run(len(dev), _ptr(dev), nil, nxl, _ptr(xc), dt, _ptr(m), _ptr(t), None if act is None else _ptr(act),
                              C.byref(prm), _ptr(out), n, _ptr(done), _ptr(sums)))
    return out, done, sums
