#!/usr/bin/env python3
"""Decision-independent properties of the float32 paths on random shapes (run on the GPU box):  python tools/fuzz_props_r05.py <seconds> [seed]
For every job (FFT on any extents 2 ... 1500 -- tuned, register, LDS-image, chirp-z engines --, WAVELET with a random bank, SHEARLET):
  * the result is finite and has the cube's dtype and shape,
  * alpha = 1: the observed traces come back bit for bit (POCS.py:616-619 with weight 0 there),
  * a slice processed alone gives the bits it gives inside the batch, and the batch split in two (batch_slices) gives the same bits,
  * an all-zero slice comes back untouched with niterations = 0."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P, shearlets

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WAVELETS = sorted(json.load(open(os.path.join(os.path.dirname(_ffi.__file__), "wavelets.json")))["wavelets"])
t_end, fails, runs = time.time() + budget, 0, {"FFT": 0, "WAVELET": 0, "SHEARLET": 0}
while time.time() < t_end:
    kind = str(rng.choice(["FFT", "FFT", "WAVELET", "SHEARLET"]))
    real = bool(rng.integers(2))
    op = str(rng.choice(["hard", "soft", "garrote"]))
    kw = dict(niter=int(rng.integers(1, 8)), thresh_op=op, thresh_model=str(rng.choice(["exponential", "linear"])), eps=float(rng.choice([0.0, 1e-3])), alpha=1.0,
              p_max=0.99, p_min=1e-2, version=str(rng.choice(["regular", "adaptive"])), precision="float32")
    extra = {}
    if kind == "FFT":
        nil, nxl = (int(rng.integers(2, 1500)) for _ in range(2))
        if rng.integers(3) == 0:
            nil = int(rng.choice([2, 3, 4, 7, 8, 16, 31, 32, 64, 97, 128, 256, 509, 512, 1009, 1024, 2048, 2053, 4096]))
        ns = int(rng.integers(2, 6))
    elif kind == "WAVELET":
        nil, nxl = (int(rng.integers(16, 400)) for _ in range(2))
        extra["wavelet"] = str(rng.choice(WAVELETS))
        flen = len(_ffi.wavelet_filters(extra["wavelet"])[0])
        if min(nil, nxl) < 2 * (flen - 1) or flen > 64:
            continue
        if not real and op == "garrote":
            kw["thresh_op"] = "soft"
        ns = int(rng.integers(2, 5))
    else:
        nil, nxl = (int(rng.choice([32, 48, 64, 96, 100, 128, 150, 256])) for _ in range(2))
        psi = shearlets.scalesShearsAndSpectra((nil, nxl))
        if not np.all(np.abs(psi).reshape(-1, psi.shape[2]).max(axis=0) > 0):
            continue
        extra["auxiliary_data"] = psi
        if not real and op == "garrote":
            kw["thresh_op"] = "soft"
        ns = int(rng.integers(2, 4))
    try:
        mask = orc.synthetic_mask(nil, nxl, float(rng.uniform(0.3, 0.8)))
        if mask.sum() == 0:
            continue
        cube = np.stack([orc.synthetic_slice(nil, nxl, int(rng.integers(1000)) + s, real=real) for s in range(ns)]) * mask
        cube = cube.astype(np.float32 if real else np.complex64)
        zero = int(rng.integers(ns))
        cube[zero] = 0
        res = []
        got = P.pocs_cube(cube, mask, transform_kind=kind, results=res, **kw, **extra)
        runs[kind] += 1
        problems = []
        if got.dtype != cube.dtype or got.shape != cube.shape or not np.isfinite(got).all():
            problems.append("dtype / shape / finite")
        keep = mask.astype(bool)
        if not all(np.array_equal(got[s][keep], cube[s][keep]) for s in range(ns)):
            problems.append("observed traces changed")
        if got[zero].any() or res[zero]["niterations"] != 0:
            problems.append("all-zero slice touched")
        pick = int(rng.integers(ns))
        alone = P.pocs_cube(cube[pick:pick + 1], mask, transform_kind=kind, **kw, **extra)
        if not np.array_equal(alone[0], got[pick]):
            problems.append(f"slice {pick} alone differs from the batch ({float(np.abs(alone[0] - got[pick]).max()):.3e})")
        split = P.pocs_cube(cube, mask, transform_kind=kind, batch_slices=max(1, ns // 2), **kw, **extra)
        if not np.array_equal(split, got):
            problems.append("batch_slices changes the bits")
        if problems:
            fails += 1
            print("FAIL", kind, (ns, nil, nxl), "real" if real else "complex", {k: v for k, v in kw.items()}, extra.get("wavelet"), problems, flush=True)
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("ERROR", kind, (ns, nil, nxl), kw, extra.get("wavelet"), repr(e)[:300], flush=True)
    if sum(runs.values()) % 50 == 0:
        print("...", runs, "failures", fails, flush=True)
print("done", runs, "failures", fails)
sys.exit(1 if fails else 0)
