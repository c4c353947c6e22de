"""Unusual host inputs of ``pocs_cube`` (GPU): Fortran-ordered, strided, read-only and memory-mapped cubes, a memory-mapped result array, bool / uint8 /
float64 masks -- through the page-locking chunk pipeline (forced onto a small cube by environment switches that are read at import: hence a child
process, tools/odd_inputs.py) -- give the bits of the plain call."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unusual_host_inputs_give_the_bits_of_the_plain_call():
    env = dict(os.environ, P3D_PIN_MIN_MIB="1", P3D_CHUNK_MIB="2")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "odd_inputs.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "DIFFERS" not in res.stdout and res.stdout.count("same bits") >= 10, res.stdout


def test_concurrent_calls_on_one_device_take_turns():
    """Two threads, different shapes and transforms, one device (the dask-threaded ``apply_ufunc`` case): plans, chunk workers and the batch
    buffers are cached per device and not re-entrant -- ``pocs_cube`` serialises calls per device, so each thread gets the bits of its serial run."""
    import threading

    import numpy as np

    from oracle import pocs_oracle as orc
    from pseudo_3d_interpolation_amd.functions import POCS as P

    jobs = []
    for shape, n, kw in (((64, 128), 5, dict(transform_kind="WAVELET", wavelet="db2", thresh_op="soft", thresh_model="linear", p_max=0.9, p_min=0.1, niter=6)),
                         ((128, 64), 3, dict(transform_kind="FFT", thresh_op="hard", thresh_model="data-driven", niter=7, p_min=1e-2)),
                         ((96, 80), 4, dict(transform_kind="FFT", thresh_op="soft", niter=5, p_min=1e-2, precision="reference")),
                         ((256, 256), 9, dict(transform_kind="FFT", thresh_op="hard", niter=8, p_min=1e-2, batch_slices=2))):
        _, mask, obs = orc.synthetic_cube(shape[0], shape[1], n, 0.5, real=kw["transform_kind"] == "WAVELET")
        jobs.append((obs, mask, dict(kw, eps=0.0)))
    serial = [P.pocs_cube(o, m, **kw) for o, m, kw in jobs]
    got = [[None] * len(jobs) for _ in range(3)]
    errors = []

    def worker(t):
        try:
            for r in range(3):
                for j in range(len(jobs)):
                    k = (j + t) % len(jobs)      # the threads walk the jobs out of phase: neighbours always run different shapes / transforms
                    o, m, kw = jobs[k]
                    res = P.pocs_cube(o, m, **kw)
                    if r == 2:
                        got[t][k] = res
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for t in range(3):
        for k in range(len(jobs)):
            assert np.array_equal(got[t][k], serial[k]), (t, k)


def test_device_arrays_without_a_plan():
    """p3d_dev_*: device memory that belongs to no plan (what bench.py keeps its cubes in at N = 1, no other GPU runtime in the process)."""
    import numpy as np

    from pseudo_3d_interpolation_amd import _ffi

    rng = np.random.default_rng(5)
    host = (rng.standard_normal((7, 32, 16)) + 1j * rng.standard_normal((7, 32, 16))).astype(np.complex64)
    a = _ffi.DeviceArray(host.shape, np.complex64).upload(host)
    assert np.array_equal(a.download(), host) and np.array_equal(a.download(2, 3), host[2:5])
    a.upload(host[:2] * 2, first=5)
    assert np.array_equal(a.download(5), host[:2] * 2)
    b = _ffi.DeviceArray(host.shape, np.complex64).copy_from(a)
    assert np.array_equal(b.download(0, 5), host[:5])
    assert not b.zero().download().any()
    plan = _ffi.Plan(32, 16, 7)
    plan.fft2_dev(a.ptr, b.ptr, 5)                 # the *_dev entry points take these pointers
    _ffi.device_synchronize(0)
    assert np.allclose(b.download(0, 5), np.fft.fft2(host[:5]), rtol=1e-5, atol=1e-3)
    free, total = _ffi.device_mem_info(0)
    assert 0 < free <= total
    with pytest.raises(ValueError):
        a.download(6, 2)
    plan.close()
    a.free(); b.free()
