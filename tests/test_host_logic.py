"""CPU-only checks of the product's host side: threshold schedule vs the reference's golden vectors,
error contract of POCS_algorithm, C-ABI surface of libp3d_hip.so, host emulation of the FFT engine."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

import pseudo_3d_interpolation_amd as pkg
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P
from pseudo_3d_interpolation_amd.functions import backends

CSRC = os.path.join(ROOT, "pseudo-3d-interpolation_amd", "csrc")


@pytest.fixture(scope="session")
def built_lib():
    if not os.path.isfile(_ffi.LIB_PATH):
        subprocess.run(["make", "-C", CSRC, "-j8"], check=True, capture_output=True)
    return _ffi.LIB_PATH


def test_package_layout():
    assert pkg.__version__
    assert os.path.basename(os.path.dirname(pkg.__file__)) == "pseudo-3d-interpolation_amd"
    assert backends.hip_library_path == _ffi.LIB_PATH
    for name in ("POCS_algorithm", "POCS", "FPOCS", "APOCS", "get_threshold_decay", "pocs_cube"):
        assert hasattr(P, name)
    assert P.POCS.keywords == {"version": "regular"}
    assert P.FPOCS.keywords == {"version": "fast"}
    assert P.APOCS.keywords == {"version": "adaptive"}


def test_schedule_matches_reference_golden():
    g = load_golden("decay.npz")
    keys = sorted(k[:-5] for k in g.files if k.endswith("_meta"))
    n = 0
    for key in keys:
        sname, model, kind, p_min, niter = [str(v) for v in g[key + "_meta"]]
        p_min = "adaptive" if p_min == "adaptive" else float(p_min)
        with np.errstate(all="ignore"):
            tau = P.get_threshold_decay(model, int(niter), "FFT", 0.99, p_min, g["X0_" + sname], kind)
        want = g[key + "_tau"]
        assert np.asarray(tau).shape == want.shape, (key, model, kind)
        assert np.asarray(tau).dtype == want.dtype, (key, model, kind)
        assert np.array_equal(np.asarray(tau), want, equal_nan=True), (key, model, kind, p_min, niter)
        n += 1
    assert n > 100


def test_batched_schedule_equals_scalar_schedule():
    """The (nslices, 6) statistics form used by the GPU path must reproduce the per-slice function."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((5, 16, 8)) + 1j * rng.standard_normal((5, 16, 8))
    stats = np.zeros((5, 6))
    for s in range(5):
        pk = X[s].max()
        stats[s] = [pk.real, pk.imag, np.abs(X[s]).max(), np.abs(X[s]).min(), np.linalg.norm(X[s]) ** 2, 0]
    for model in ("linear", "exponential", "exponential-2", "inverse_proportional-2"):
        for kind in ("values", "factors"):
            for p_min in (1e-3, "adaptive"):
                if kind == "factors" and p_min == "adaptive":
                    continue
                got = P._schedule_from_stats(stats, X[0].size, model, 12, 0.99, p_min, kind)
                for s in range(5):
                    want = P.get_threshold_decay(model, 12, "FFT", 0.99, p_min, X[s], kind)
                    assert np.allclose(got[s], want, rtol=1e-13, atol=0), (model, kind, p_min)


class _NumpyPlan:
    """Stand-in for the device plan in the host-side tests of the 'data-driven' schedule: the three calls _data_driven_batch makes,
    answered by NumPy exactly as include/p3d.h specifies them (complex64 spectrum, NumPy's lexicographic complex order)."""

    def fft2(self, x):
        return np.fft.fft2(np.asarray(x, np.complex64)).astype(np.complex64)

    def sorted_spectrum(self, x):
        X = self.fft2(x)
        self._sorted = [np.sort(s.ravel())[::-1] for s in X]
        return np.asarray([s[0] for s in self._sorted], np.complex64)

    def data_driven_pick(self, lo, hi, niter):
        tau = np.zeros((len(self._sorted), niter), np.complex64)
        count = np.zeros(len(self._sorted), np.int64)
        for s, v in enumerate(self._sorted):
            sel = v[(v > np.complex64(lo[s])) & (v < np.complex64(hi[s]))]
            count[s] = sel.size
            if sel.size:
                idx = [0] + [-(-(i * (sel.size - 1)) // (niter - 1)) for i in range(1, niter)]   # integer ceil
                tau[s] = sel[idx]
        return tau, count


def test_data_driven_batch_is_the_reference_schedule_per_slice():
    """_data_driven_batch (bounds on the host in the reference's arithmetic, sort + picks behind the plan) against _data_driven
    (POCS.py:356-362 restated, pinned by the golden schedules above) -- including the integer form of the reference's float64 ceil,
    an all-zero slice, the 'adaptive' lower bound (host path) and the IndexError of an empty selection."""
    rng = np.random.default_rng(5)
    cube = (rng.standard_normal((4, 24, 20)) + 1j * rng.standard_normal((4, 24, 20))).astype(np.complex64)
    cube[2] = 0
    active = cube.reshape(4, -1).any(axis=1)
    plan = _NumpyPlan()
    for niter in (1, 2, 7, 33):
        got = P._data_driven_batch(plan, cube, active, niter, 0.99, 1e-3)
        X0 = plan.fft2(cube)
        for s in (0, 1, 3):
            np.testing.assert_array_equal(got[s].astype(np.complex64), P._data_driven(X0[s], niter, 0.99, 1e-3))
        assert not got[2].any()
    adaptive = P._data_driven_batch(plan, cube, active, 5, 0.99, 'adaptive')
    np.testing.assert_array_equal(adaptive[0].astype(np.complex64), P._data_driven(plan.fft2(cube)[0], 5, 0.99, 'adaptive'))
    with pytest.raises(IndexError):
        P._data_driven_batch(plan, cube, active, 5, 1e-30, 1e-3)


def test_error_contract_matches_reference():
    tab = {str(r[0]): (str(r[1]), str(r[2])) for r in load_golden("errors.npz")["table"]}
    x = np.ones((8, 8), np.complex64)
    m = np.ones((8, 8), np.uint8)
    f, i = np.fft.fft2, np.fft.ifft2
    with pytest.raises(ValueError) as e:
        P.POCS_algorithm(x, m * 2, transform=f, itransform=i, transform_kind="FFT")
    assert str(e.value) == tab["mask_gt_1"][1]
    with pytest.raises(ValueError) as e:
        P.POCS_algorithm(x, m, transform=None, itransform=None, transform_kind="FFT")
    assert str(e.value) == tab["no_transform"][1]
    with pytest.raises(ValueError) as e:
        P.POCS_algorithm(x, m, transform=f, itransform=i, transform_kind="HAAR")
    assert str(e.value) == tab["bad_kind"][1]
    with pytest.raises(ValueError) as e:
        P.POCS_algorithm(x, m, transform=f, itransform=i, transform_kind="SHEARLET")
    assert str(e.value) == tab["shearlet_no_psi"][1]
    # options the HIP build does not cover yet fail loudly instead of falling back to the CPU
    with pytest.raises(NotImplementedError):
        P.POCS_algorithm(x, m, transform=f, itransform=i, transform_kind="CURVELET")
    from functools import partial

    def wavedec2(a, wavelet=None, mode=None):   # stand-ins for pywt's pair: recognised by name, never called
        raise AssertionError("the host transform must not be called")

    def waverec2(c, wavelet=None, mode=None):
        raise AssertionError("the host transform must not be called")

    wf, wi = partial(wavedec2, wavelet="db2", mode="smooth"), partial(waverec2, wavelet="db2", mode="smooth")
    with pytest.raises(NotImplementedError):
        P.POCS_algorithm(x, m, transform=wf, itransform=wi, transform_kind="WAVELET", thresh_op="soft-percentile")
    with pytest.raises(IndexError):  # the reference's threshold_wavelet indexes a (1, 1) tau per level (POCS.py:135-166)
        P.POCS_algorithm(x, m, transform=wf, itransform=wi, transform_kind="WAVELET", decay_kind="factors")
    with pytest.raises(ValueError):
        P.POCS_algorithm(x, m, transform=f, itransform=i, transform_kind="FFT", thresh_op="median")
    # the reference calls the callables it is given (POCS.py:592, 613); the HIP path cannot, so anything that is not the transform
    # `transform_kind` names is refused instead of being ignored
    for bad_f, bad_i, kind in ((lambda a: a, lambda a: a, "FFT"), (np.fft.fft, np.fft.ifft, "FFT"), (i, f, "FFT"),
                               (partial(f, s=(16, 16)), i, "FFT"), (f, i, "WAVELET"), (wf, wi, "FFT")):
        with pytest.raises(NotImplementedError, match="cannot run an arbitrary callable"):
            P.POCS_algorithm(x, m, transform=bad_f, itransform=bad_i, transform_kind=kind)
    P._check_transform_callables("FFT", partial(np.fft.fft2, axes=(-2, -1)), partial(np.fft.ifft2, norm="backward"))
    import scipy.fft
    P._check_transform_callables("FFT", scipy.fft.fft2, scipy.fft.ifft2)


def test_product_never_imports_oracle():
    pk = os.path.join(ROOT, "pseudo-3d-interpolation_amd")
    for dirpath, _, files in os.walk(pk):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), fn
                assert "pocs_oracle" not in txt, fn


def test_cabi_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "p3d.h")).read()
    declared = set(re.findall(r"\b(p3d_[a-z0-9_]+)\s*\(", header))
    declared -= {"p3d_plan", "p3d_pocs_params"}
    assert len(declared) >= 15
    lib = ctypes.CDLL(built_lib)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/p3d.h but not exported"
    assert declared == set(_ffi.PROTOTYPES), "ctypes prototype table out of sync with include/p3d.h"
    assert lib.p3d_abi_version() == 1
    # shape query needs no GPU
    assert lib.p3d_shape_supported(1024, 1024) == 1
    assert lib.p3d_shape_supported(64, 2048) == 1
    assert lib.p3d_shape_supported(90, 50) == 1       # any-length fallback
    assert lib.p3d_shape_supported(20000, 8) == 0     # beyond the LDS-resident line limit
    assert ctypes.sizeof(_ffi.PocsParams) == 32


def test_fft_engine_host_emulation(tmp_path):
    """Drive the device FFT templates (butterflies, pass schedule, Stockham scatter, twiddle layout,
    both LDS views) thread-by-thread on the CPU for every supported line length."""
    exe = tmp_path / "fft_host"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", CSRC, os.path.join(ROOT, "tests", "csrc", "test_fft_host.cpp"),
                    "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:]
    assert "ALL OK" in res.stdout


def test_which_calls_take_a_double_precision_loop(monkeypatch):
    """functions.POCS._double_loop_wanted: the one routing rule of pocs_cube and sharding.pocs_block_on_device (no GPU needed: the library only looks up
    its plan tables).  Double precision when asked for, for double-precision cubes, for WAVELET banks the float32 tiles do not hold, and for SHEARLET on
    7-smooth extents that are not powers of two (the fused double-precision passes beat the unfused float32 ones there)."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    monkeypatch.delenv("P3D_PRECISION", raising=False)
    psi = object()   # (only its presence matters here)
    w = P._double_loop_wanted
    assert w(np.complex64, 1024, 1024, "FFT", "hard", None) == (None, False)
    assert w(np.complex64, 1024, 1024, "FFT", "soft", "reference") == ("reference", True)
    assert w(np.complex128, 64, 64, "FFT", "soft", None) == (None, True)
    assert w(np.float64, 64, 64, "WAVELET", "soft", "float32") == ("float32", False)
    assert w(np.float32, 256, 256, "WAVELET", "soft", None, "db4")[1] is False
    assert w(np.float32, 256, 256, "WAVELET", "soft", None, "db38")[1] is True          # 76 taps
    assert w(np.float32, 256, 256, "WAVELET", "soft", "float32", "db38")[1] is False    # (the float32 kernels then refuse the bank)
    assert w(np.float32, 1000, 1000, "SHEARLET", "hard", None, auxiliary_data=psi)[1] is True
    assert w(np.float32, 1024, 2048, "SHEARLET", "hard", None, auxiliary_data=psi)[1] is False   # powers of two: the fused float32 passes
    assert w(np.float32, 1009, 1000, "SHEARLET", "hard", None, auxiliary_data=psi)[1] is False   # a prime extent: no fused double-precision passes
    assert w(np.float32, 1000, 1000, "SHEARLET", "hard", "float32", auxiliary_data=psi)[1] is False
    assert w(np.float32, 1000, 1000, "SHEARLET", "hard-percentile", None, auxiliary_data=psi)[1] is False
    monkeypatch.setenv("P3D_PRECISION", "reference")
    assert w(np.complex64, 1024, 1024, "FFT", "hard", None) == ("reference", True)
    with pytest.raises(ValueError):
        w(np.complex64, 8, 8, "FFT", "hard", "double")


@pytest.mark.parametrize("args,inc", [(["--min", "120", "--max", "4096", "--parts", "8"], "p3d_mix_plans.inc"),
                                       (["--f64", "--all-smooth", "--parts", "8"], "p3d_mix64_plans.inc")])
def test_plan_lists_are_what_the_generator_writes(args, inc):
    """The instantiation lists of the mixed-radix register engines are generated files (tools/gen_mix_plans.py): an edit of the chooser without a
    regeneration, or a hand edit of a list, shows up here."""
    out = subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_mix_plans.py")] + args, check=True, capture_output=True, text=True).stdout
    assert out == open(os.path.join(ROOT, "pseudo-3d-interpolation_amd", "csrc", inc)).read()
