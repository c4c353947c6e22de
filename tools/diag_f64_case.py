#!/usr/bin/env python3
"""A saved fuzz case (FUZZ_ONLY=<case> FUZZ_SAVE=file.npz python tools/fuzz_parity.py ...) through the double-precision loop, step by step against
the float64 oracle: statistics, schedule, cost sums per iteration, first iteration at which a slice leaves the oracle's trajectory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats

d = np.load(sys.argv[1], allow_pickle=True)
cube, mask, kw = d["cube"], d["mask"], eval(str(d["kw"][0]))
print(cube.shape, cube.dtype, kw)
ns, nil, nxl = cube.shape
wide = cube.astype(np.float64 if cube.dtype.kind == "f" else np.complex128)
with _ffi.Plan64(nil, nxl, ns) as plan:
    st = plan.stats(cube)
    for s in range(ns):
        X = np.fft.fft2(wide[s])
        a = np.abs(X)
        print(f"slice {s}: device lexmax {st[s,0]:.17g}{st[s,1]:+.17g}j max|X| {st[s,2]:.17g} min|X| {st[s,3]:.17g} sum|X|^2 {st[s,4]:.17g}")
        print(f"         numpy  lexmax {X.max().real:.17g}{X.max().imag:+.17g}j max|X| {a.max():.17g} min|X| {a.min():.17g} sum|X|^2 {(a**2).sum():.17g}")
    tau = _schedule_from_stats(st, nil * nxl, kw["thresh_model"], kw["niter"], kw["p_max"], kw["p_min"], "values")
    for s in range(ns):
        t_ref = orc.threshold_schedule(kw["thresh_model"], kw["niter"], "FFT", kw["p_max"], kw["p_min"], np.fft.fft2(wide[s]), "values")
        print(f"slice {s}: tau device-side {np.asarray(tau[s])[:3]} ... {np.asarray(tau[s])[-2:]}\n         tau oracle      {np.asarray(t_ref)[:3]} ... {np.asarray(t_ref)[-2:]}")
    refs = []
    for s_ in range(ns):
        t_ref = orc.threshold_schedule(kw["thresh_model"], kw["niter"], "FFT", kw["p_max"], kw["p_min"], np.fft.fft2(wide[s_]), "values")
        prev, traj = wide[s_], []
        for k in range(kw["niter"]):   # (regular version)
            tk = complex(np.ravel(t_ref[k])[0])
            prev, spec, shr = orc.pocs_step(prev, wide[s_], mask, tk, thresh_op=kw["thresh_op"], alpha=kw["alpha"])
            near = np.sort(np.abs(np.abs(spec) - tk.real).ravel())[:2] / tk.real
            traj.append((prev, int(np.count_nonzero(shr)), near))
        refs.append(traj)
    tau = np.asarray(tau).reshape(ns, kw["niter"])
    for K in range(1, kw["niter"] + 1):
        out, done, sums, _ = plan.run(cube, mask, np.ascontiguousarray(tau[:, :K]), K, thresh_op=kw["thresh_op"], version=kw["version"], eps=0.0, alpha=kw["alpha"])
        print("after", K, "iterations:", "  ".join("slice %d: %.1e (oracle keeps %d, nearest |X| to tau: %.1e %.1e)" % (
            s_, np.linalg.norm(out[s_] - refs[s_][K - 1][0]) / np.linalg.norm(refs[s_][K - 1][0]), refs[s_][K - 1][1], float(refs[s_][K - 1][2][0]), float(refs[s_][K - 1][2][1])) for s_ in range(ns)))
