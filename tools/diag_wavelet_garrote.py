#!/usr/bin/env python3
"""Complex cube + garrote + WAVELET: device against the oracle per slice, for a few shapes / wavelets (GPU box).  Environment switches apply
(P3D_WAVELET_NO_L1FUSE, P3D_WAVELET_UNFUSED, P3D_WAVELET_NO_COARSE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pseudo_3d_interpolation_amd.functions.POCS as P
from oracle import pocs_oracle as orc, wavelet_oracle as wo
for (nil, nxl, wname, model, niter, alpha, pmin, op) in [(201, 215, "sym5", "linear", 4, 1.0, 0.1, "garrote"), (155, 255, "db4", "linear", 5, 0.8, 0.01, "garrote"),
                                                        (123, 115, "coif2", "linear", 3, 1.0, 0.01, "garrote"), (169, 294, "bior3.5", "exponential", 2, 0.8, 0.1, "garrote"),
                                                        (201, 215, "sym5", "linear", 4, 1.0, 0.1, "soft"), (201, 215, "sym5", "exponential", 4, 1.0, 0.1, "garrote")]:
    mask = orc.synthetic_mask(nil, nxl, 0.5)
    cube = (np.stack([orc.synthetic_slice(nil, nxl, 100 + s) for s in range(2)]) * mask).astype(np.complex64)
    kw = dict(niter=niter, thresh_op=op, thresh_model=model, eps=0.0, alpha=alpha, p_max=0.99, p_min=pmin)
    got = P.pocs_cube(cube, mask, transform_kind="WAVELET", wavelet=wname, **kw)
    want = wo.pocs_cube_wavelet(cube.astype(np.complex128), mask, wavelet=wname, **kw)
    err = [float(np.linalg.norm(got[s] - want[s]) / np.linalg.norm(want[s])) for s in range(2)]
    print(f"{nil}x{nxl} {wname:8s} {op:8s} {model:12s} it={niter} a={alpha} pmin={pmin}: rel l2 {err[0]:.2e} {err[1]:.2e}   |got| {np.linalg.norm(got[0]):.3e} |want| {np.linalg.norm(want[0]):.3e} finite {np.isfinite(got).all()}", flush=True)
