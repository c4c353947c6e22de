#!/usr/bin/env python3
"""Randomised checks of the helper entry points against NumPy / SciPy closed forms (run on the GPU box):  python tools/fuzz_helpers_r05.py <seconds> [seed]
  * p3d_time2freq / p3d_freq2time (steps 12 / 14) on random trace lengths, paddings, real_only, t0: F[k] = dt exp(-2 pi i f_k t0) FFT(x)[k] and the round trip,
  * p3d_smooth_gaussian / p3d_smooth_median (step 15) against scipy.ndimage ('reflect'),
  * p3d_fft2_c64 on random (any) shapes against np.fft.fft2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as ndi
from pseudo_3d_interpolation_amd import _ffi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
t_end, fails, runs = time.time() + budget, 0, {"t2f": 0, "smooth": 0, "fft2": 0}
while time.time() < t_end:
    kind = str(rng.choice(["t2f", "smooth", "fft2"]))
    try:
        if kind == "t2f":
            nt, ntr = int(rng.integers(8, 3000)), int(rng.integers(1, 40))
            nfft = nt if rng.integers(2) else int(nt + rng.integers(0, nt))
            real_only = bool(rng.integers(2))
            dt, t0 = float(rng.choice([1.0, 0.004, 0.05])), float(rng.choice([0.0, 0.3, -1.25]))
            x = rng.standard_normal((nt, ntr)).astype(np.float32)
            F = _ffi.time2freq(x, dt, t0=t0, nfft=nfft, real_only=real_only)
            f = (np.fft.rfftfreq if real_only else np.fft.fftfreq)(nfft, dt)
            want = dt * np.exp(-2j * np.pi * f * t0)[:, None] * (np.fft.rfft if real_only else np.fft.fft)(x.astype(np.float64), n=nfft, axis=0)
            e1 = rel(F, want)
            back = _ffi.freq2time(F, dt, t0=t0, nfft=nfft, real_only=real_only)
            e2 = rel(back[:nt], x)
            ok, what = e1 < 5e-6 and e2 < 5e-6, (nt, ntr, nfft, real_only, dt, t0, e1, e2)
        elif kind == "smooth":
            n, ny, nx = int(rng.integers(1, 5)), int(rng.integers(8, 300)), int(rng.integers(8, 300))
            x = rng.standard_normal((n, ny, nx)).astype(np.float32)
            if rng.integers(2):
                sigma = float(rng.uniform(0.5, 4.0))
                got = _ffi.smooth_slices(x, "gaussian", sigma=sigma)
                want = np.stack([ndi.gaussian_filter(s.astype(np.float64), sigma, mode="reflect") for s in x])
                ok, what = rel(got, want) < 5e-6, (n, ny, nx, "gaussian", sigma, rel(got, want))
            else:
                size = int(rng.choice([3, 5, 7]))
                got = _ffi.smooth_slices(x, "median", size=size)
                want = np.stack([ndi.median_filter(s, size=size, mode="reflect") for s in x])
                ok, what = np.array_equal(got, want), (n, ny, nx, "median", size)
        else:
            shape = (int(rng.integers(2, 1400)), int(rng.integers(2, 1400)))
            x = (rng.standard_normal((2,) + shape) + 1j * rng.standard_normal((2,) + shape)).astype(np.complex64)
            with _ffi.Plan(shape[0], shape[1], 2) as plan:
                F = plan.fft2(x)
                b = plan.fft2(F, inverse=True)
            e1, e2 = rel(F, np.fft.fft2(x.astype(np.complex128))), rel(b, x)
            ok, what = e1 < 3e-6 and e2 < 3e-6, (shape, e1, e2)
        runs[kind] += 1
        if not ok:
            fails += 1
            print("FAIL", kind, what, flush=True)
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("ERROR", kind, repr(e)[:300], flush=True)
print("done", runs, "failures", fails)
sys.exit(1 if fails else 0)
