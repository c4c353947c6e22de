#!/usr/bin/env python3
"""Randomised differential run of the WAVELET and SHEARLET variants of pocs_cube against their oracles (GPU box):
python tools/fuzz_wavelet.py [cases] [seed].  Few iterations: the wavelet iteration is expansive on decimated data."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pseudo_3d_interpolation_amd.functions.POCS as P
from pseudo_3d_interpolation_amd.functions import shearlets
from oracle import pocs_oracle as orc, wavelet_oracle as wo, shearlet_oracle as sho

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
WAVELETS = ["haar", "db2", "db4", "db8", "sym5", "coif2", "coif5", "bior2.2", "bior3.5", "rbio2.4"]
bad = 0
for case in range(ncases):
    kind = "WAVELET" if rng.random() < 0.75 else "SHEARLET"
    if kind == "WAVELET":
        nil, nxl = int(rng.integers(24, 300)), int(rng.integers(24, 300))
    else:
        nil, nxl = int(rng.choice([32, 64, 48, 96, 128])), int(rng.choice([32, 64, 48, 96, 128]))
    ns = int(rng.integers(1, 4))
    real = rng.random() < 0.6
    op = str(rng.choice(["hard", "soft", "garrote"]))
    kw = dict(niter=int(rng.integers(2, 6)), thresh_op=op, thresh_model=str(rng.choice(["exponential", "linear"])),
              eps=float(rng.choice([0.0, 1e-9])), alpha=float(rng.choice([1.0, 0.8])), p_max=float(rng.choice([0.99, 0.9])),
              p_min=float(rng.choice([1e-1, 1e-2])))
    mask = orc.synthetic_mask(nil, nxl, float(rng.choice([0.3, 0.5])))
    first = int(rng.integers(0, 500))
    cube = np.stack([orc.synthetic_slice(nil, nxl, first + s) for s in range(ns)]) * mask
    cube = cube.real.astype(np.float32) if real else cube.astype(np.complex64)
    ref_in = cube.astype(np.float64 if real else np.complex128)
    try:
        if kind == "WAVELET":
            wname = str(rng.choice(WAVELETS))
            got = P.pocs_cube(cube, mask, transform_kind="WAVELET", wavelet=wname, **kw)
            want = wo.pocs_cube_wavelet(ref_in, mask, wavelet=wname, **kw)
            tag = wname
        else:
            psi = shearlets.scalesShearsAndSpectra((nil, nxl))
            got = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
            want = sho.pocs_cube_shearlet(ref_in, mask, sho.scales_shears_and_spectra((nil, nxl)), **kw)
            tag = f"{psi.shape[-1]} shearlets"
    except Exception as e:  # noqa: BLE001
        print("CASE", case, kind, (nil, nxl, ns), cube.dtype.name, kw, "raised", repr(e)[:200], flush=True)
        bad += 1
        continue
    err = np.array([np.linalg.norm(got[s] - want[s]) / max(np.linalg.norm(want[s]), 1e-30) for s in range(ns)])
    flag = not np.isfinite(err).all() or np.median(err) > (2e-5 if op == "soft" else 5e-4) or err.max() > 2e-2
    print("case %2d %-8s %-14s %3dx%-3d x%d %-9s %-7s %-11s it=%d eps=%g a=%.1f pmin=%g  med %.1e max %.1e %s" % (
        case, kind, tag, nil, nxl, ns, cube.dtype.name, op, kw["thresh_model"], kw["niter"], kw["eps"], kw["alpha"], kw["p_min"],
        np.median(err), err.max(), "<-- CHECK" if flag else ""), flush=True)
    bad += bool(flag)
P.release_plans()
print("flagged:", bad)
sys.exit(1 if bad else 0)
