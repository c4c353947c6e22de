#!/usr/bin/env python3
"""Randomised campaign for the paths of round 5 (run on the GPU box; not part of the test suite: minutes of oracle time).
    python tools/fuzz_r05.py <seconds> [seed]
  * the double-precision FFT loop (register engine on any planned extent) against the float64 oracle,
  * the float32 FFT loop on 7-smooth extents against the oracle,
  * the double-precision SHEARLET loop (fused passes where the extents allow) against its oracle.
Prints one line per failure and a summary; exit status 1 if anything failed."""
import os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc, shearlet_oracle as so, wavelet_oracle as wo
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P, shearlets

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
csrc = os.path.join(os.path.dirname(_ffi.__file__), "csrc")
L64 = sorted(int(m) for m in re.findall(r"^X\((\d+),", open(os.path.join(csrc, "p3d_mix64_plans.inc")).read(), re.M))
L32 = sorted(int(m) for m in re.findall(r"^X\((\d+),", open(os.path.join(csrc, "p3d_mix_plans.inc")).read(), re.M))
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
t_end, fails, runs = time.time() + budget, 0, {"fft64": 0, "fft32": 0, "shear64": 0, "wav64": 0}
import json
WAVELETS = sorted(json.load(open(os.path.join(os.path.dirname(_ffi.__file__), "wavelets.json")))["wavelets"])


def pick(lengths, hi):
    c = [n for n in lengths if n <= hi]
    return int(rng.choice(c))


while time.time() < t_end:
    kind = rng.choice(["fft64", "fft32", "shear64", "wav64"], p=[float(v) for v in os.environ.get("FUZZ_MIX", "0.3,0.2,0.25,0.25").split(",")])
    real = bool(rng.integers(2))
    op = str(rng.choice(["hard", "soft", "garrote"])) if real else str(rng.choice(["hard", "soft"]))
    kw = dict(niter=int(rng.integers(2, 12)), thresh_op=op, thresh_model=str(rng.choice(["exponential", "linear", "exponential-2"])),
              eps=float(rng.choice([0.0, 0.0, 1e-4])), alpha=float(rng.choice([1.0, 1.0, 0.8])), p_max=0.99, p_min=float(rng.choice([1e-2, 1e-3, 0.05])),
              version=str(rng.choice(["regular", "regular", "adaptive"])))
    missing = float(rng.uniform(0.3, 0.8))
    try:
        if kind == "wav64":
            nil, nxl = int(rng.integers(24, 260)), int(rng.integers(24, 260))
            wavelet = str(rng.choice(WAVELETS))
            flen = len(_ffi.wavelet_filters(wavelet)[0])
            if min(nil, nxl) < 2 * (flen - 1):
                continue   # (no decomposition level)
            if not real and op == "garrote":
                op = kw["thresh_op"] = "soft"
            if op == "hard" and "linear" not in kw["thresh_model"] and kw["thresh_model"] != "exponential":
                kw["thresh_model"] = "exponential"
            kw["niter"] = min(kw["niter"], 6)   # (the 'smooth' iteration is expansive: long runs amplify the last bit)
            mask = orc.synthetic_mask(nil, nxl, missing)
            cube = np.stack([orc.synthetic_slice(nil, nxl, int(rng.integers(1000)) + s, real=real) for s in range(2)]) * mask
            cube = cube.astype(np.float64 if real else np.complex128)
            infos, res = [], []
            want = wo.pocs_cube_wavelet(cube, mask, wavelet=wavelet, infos=infos, **kw)
            if not np.isfinite(want).all():
                continue
            got = P.pocs_cube(cube, mask, transform_kind="WAVELET", wavelet=wavelet, results=res, **kw)
            kw["wavelet"] = wavelet
            tol = 1e-7 if op == "hard" else 1e-9
        elif kind == "shear64":
            nil, nxl = pick(L64, 300), pick(L64, 300)
            if rng.integers(2):   # any extent: the unfused passes around the line transforms
                nil = int(rng.integers(24, 200))
            psi = shearlets.scalesShearsAndSpectra((nil, nxl))
            if not np.all(np.abs(psi).reshape(-1, psi.shape[2]).max(axis=0) > 0):
                continue   # (a degenerate frame: a shearlet without any sample, the reference's schedule divides by zero)
            mask = orc.synthetic_mask(nil, nxl, missing)
            cube = np.stack([orc.synthetic_slice(nil, nxl, int(rng.integers(1000)) + s, real=real) for s in range(2)]) * mask
            cube = (cube + (1.5 if real else 0)).astype(np.float64 if real else np.complex128) * mask
            infos, res = [], []
            want = so.pocs_cube_shearlet(cube, mask, psi, infos=infos, **kw)
            if not np.isfinite(want).all():
                continue
            got = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res, **kw)
            tol = 1e-8 if op == "hard" else 1e-10
        else:
            lengths = L64 if kind == "fft64" else L32
            nil, nxl = pick(lengths, 1300), pick(lengths, 1300)
            if kind == "fft64" and rng.integers(2):   # any length (primes included): the LDS-image passes of p3d_f64.hip, alone or beside the register engine
                nil = int(rng.integers(17, 700))
                if rng.integers(2):
                    nxl = int(rng.integers(17, 700))
            mask = orc.synthetic_mask(nil, nxl, missing)
            cube = np.stack([orc.synthetic_slice(nil, nxl, int(rng.integers(1000)) + s, real=real) for s in range(2)]) * mask
            if kind == "fft64":
                cube = cube.astype(np.float64 if real else np.complex128)
                tol = 1e-8 if op == "hard" else 1e-10
            else:
                cube = cube.astype(np.float32 if real else np.complex64)
                kw["thresh_op"] = "soft"   # (float32 kernels: continuous operator for an end-to-end bar)
                kw["eps"] = 0.0
                tol = 2e-5
            infos, res = [], []
            want = orc.pocs_cube(cube.astype(np.float64 if real else np.complex128), mask, infos=infos, **kw)
            got = P.pocs_cube(cube, mask, results=res, **kw)
        runs[kind] += 1
        bad = [s for s in range(2) if rel(got[s], want[s]) > tol or (kw["eps"] == 0.0 and res[s]["niterations"] != infos[s]["niterations"])]
        if kw["eps"] > 0 and kind != "fft32":
            bad += [s for s in range(2) if res[s]["niterations"] != infos[s]["niterations"]]
        if bad and kind == "fft32":
            # float32 kernels against the double-fed oracle: is it the engine, or a threshold decision that float32 rounding flips (a soft threshold with a
            # COMPLEX tau is not continuous: where Re(1 - tau / |X|) crosses zero the output jumps by |Im tau|)?  The same job on the LDS-image passes
            # (another sequence of roundings) and in double precision tells: the double loop must match the oracle, the two float32 engines differ from it alike
            P.release_plans()
            os.environ["P3D_NO_MIX"] = "1"
            other = P.pocs_cube(cube, mask, **kw)
            del os.environ["P3D_NO_MIX"]
            P.release_plans()
            dbl = P.pocs_cube(cube, mask, precision="reference", **kw)
            e_other = [rel(other[s], want[s]) for s in range(2)]
            e_dbl = [rel(dbl[s], want[s]) for s in range(2)]
            verdict = "float32 decision flips" if max(e_dbl) < 1e-6 else "ENGINE?"
            if verdict == "ENGINE?":
                fails += 1
            flips = flips + 1 if "flips" in dir() else 1
            print("NOTE" if verdict != "ENGINE?" else "FAIL", kind, (nil, nxl), "real" if real else "complex", kw, "register engine", [rel(got[s], want[s]) for s in range(2)],
                  "LDS-image passes", e_other, "double loop (float32 in / out)", e_dbl, "->", verdict, flush=True)
        elif bad:
            fails += 1
            print("FAIL", kind, (nil, nxl), "real" if real else "complex", kw, [(rel(got[s], want[s]), res[s]["niterations"], infos[s]["niterations"]) for s in range(2)], flush=True)
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("ERROR", kind, (nil, nxl), kw, repr(e)[:300], flush=True)
    if sum(runs.values()) % 20 == 0:
        print("...", runs, "failures", fails, flush=True)
print("done", runs, "failures", fails)
sys.exit(1 if fails else 0)
