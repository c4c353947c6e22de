#!/usr/bin/env python3
"""Randomised differential run of pocs_cube against the oracle (GPU box): python tools/fuzz_parity.py [cases] [seed].
Shapes mix tuned (power-of-two) and flexible lengths; prints every case whose median per-slice rel-L2 exceeds 1e-5 or whose
iteration counts differ, and exits non-zero if there is one."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pseudo_3d_interpolation_amd.functions.POCS as P
from oracle import pocs_oracle as orc

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
LENS = [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 20, 48, 60, 75, 100, 120, 250, 300, 500, 1000,
        110, 130, 143, 286, 770, 37, 74, 999, 1500, 1100,   # 11- / 13-point butterflies, chirp-z, four-wavefront rows
        34, 62, 331, 530, 1009, 1451, 2039]                 # chirp-z on 64 ... 4096 points (p3d_chirp.hip)
bad = 0
for case in range(ncases):
    while True:
        nil, nxl = int(rng.choice(LENS)), int(rng.choice(LENS))
        if nil * nxl <= 1 << 20: break
    ns = int(rng.integers(1, 6))
    dtype = np.complex64 if rng.random() < 0.7 else np.float32
    op = str(rng.choice(["hard", "soft", "garrote"])) if dtype == np.complex64 else "hard"
    if op == "garrote": op = "soft" if rng.random() < 0.5 else "garrote"
    kw = dict(niter=int(rng.integers(2, 14)), thresh_op=op, thresh_model=str(rng.choice(["exponential", "linear", "inverse_proportional"])),
              eps=float(rng.choice([0.0, 1e-9, 1e-5])), alpha=float(rng.choice([1.0, 1.0, 0.8])), p_max=0.99,
              p_min=float(rng.choice([1e-2, 1e-3])), version=str(rng.choice(["regular", "regular", "adaptive"])))
    if op == "garrote" and dtype == np.complex64: kw["thresh_model"] = "exponential"
    missing = float(rng.choice([0.3, 0.5, 0.8]))
    mask = orc.synthetic_mask(nil, nxl, missing)
    seed0 = int(rng.integers(0, 1000))
    cube = np.stack([orc.synthetic_slice(nil, nxl, seed0 + s) for s in range(ns)]) * mask
    cube = (cube.real if dtype == np.float32 else cube).astype(dtype)
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):
        continue   # (FUZZ_ONLY=<case>: that case alone, with the float32-fed oracle beside the float64 one)
    if os.environ.get("FUZZ_SAVE"):   # the case's inputs for a closer look (tools/diag_f64_case.py)
        np.savez(os.environ["FUZZ_SAVE"], cube=cube, mask=mask, kw=np.array([repr(kw)]))
    res, infos = [], []
    try:
        got = P.pocs_cube(cube, mask, results=res, **kw)
        want = orc.pocs_cube(cube.astype(np.float64 if dtype == np.float32 else np.complex128), mask, infos=infos, **kw)
    except Exception as e:  # noqa: BLE001
        print("CASE", case, (nil, nxl, ns), dtype.__name__, kw, "raised", repr(e)[:200], flush=True)
        bad += 1
        continue
    err = np.array([np.linalg.norm(got[s] - want[s]) / max(np.linalg.norm(want[s]), 1e-30) for s in range(ns)])
    its = [r["niterations"] for r in res]
    its_ref = [i["niterations"] for i in infos]
    flag = (np.median(err) > 1e-5) or (err.max() > 5e-3) or not np.isfinite(err).all()
    if kw["eps"] == 0.0 and its != its_ref: flag = True
    print("case %2d %4dx%-4d x%d %-9s %-7s %-20s it=%2d eps=%g a=%.1f %-8s miss=%.1f  med %.1e max %.1e %s%s" % (
        case, nil, nxl, ns, dtype.__name__, op, kw["thresh_model"], kw["niter"], kw["eps"], kw["alpha"], kw["version"], missing,
        np.median(err), err.max(), "" if its == its_ref else f"its {its} vs {its_ref} ", "<-- CHECK" if flag else ""), flush=True)
    bad += bool(flag)
    if os.environ.get("FUZZ_ONLY"):
        w32 = orc.pocs_cube(cube, mask, **kw)     # NumPy's own single-precision run of the same loop (complex64 / float32 fed)
        e32 = np.array([np.linalg.norm(w32[s] - want[s]) / max(np.linalg.norm(want[s]), 1e-30) for s in range(ns)])
        g32 = np.array([np.linalg.norm(got[s] - w32[s]) / max(np.linalg.norm(want[s]), 1e-30) for s in range(ns)])
        print("        per slice, device vs float64 oracle:", " ".join("%.1e" % v for v in err))
        print("        NumPy single vs float64 oracle:      ", " ".join("%.1e" % v for v in e32))
        print("        device vs NumPy single:              ", " ".join("%.1e" % v for v in g32), flush=True)
        print("        iterations: device", its, "oracle", its_ref, " costs (oracle, last 3):", [np.array(i["costs"][-3:]).round(8).tolist() for i in infos])
        if kw["eps"] > 0:   # which iterate of the oracle does the device hand back for a slice that stopped early?
            for s_ in range(ns):
                if its_ref[s_] >= kw["niter"]:
                    continue
                for kk in (its_ref[s_] - 1, its_ref[s_], its_ref[s_] + 1):
                    w = orc.pocs_cube(cube[s_:s_ + 1].astype(np.float64 if dtype == np.float32 else np.complex128), mask, **dict(kw, niter=kk, eps=0.0))[0]
                    print("        slice %d: device vs the oracle's iterate after %d iterations: %.1e" % (s_, kk, np.linalg.norm(got[s_] - w) / np.linalg.norm(w)), flush=True)
    if (flag or os.environ.get("FUZZ_ALL")) and os.environ.get("FUZZ_SWITCHES"):   # the same on the slower equivalent paths?
        for sw in os.environ["FUZZ_SWITCHES"].split(","):
            P.release_plans()
            os.environ[sw] = "1"
            try:
                alt = P.pocs_cube(cube, mask, **kw)
            finally:
                del os.environ[sw]
                P.release_plans()
            d = np.array([np.linalg.norm(alt[s] - got[s]) / max(np.linalg.norm(got[s]), 1e-30) for s in range(ns)])
            same = bool(np.array_equal(alt, got))
            exact = sw in ("P3D_NO_SPARSE", "P3D_NO_PIPE64", "P3D_NO_PIPE", "P3D_NO_COMPACT") and not (sw != "P3D_NO_SPARSE" and dtype == np.float32)
            if not same and (exact or d.max() > 1e-4):
                bad += 1
            if not same or flag:
                print("        with %s=1: rel diff to the default path max %.1e, equal: %s%s" % (sw, d.max(), same, "  <-- PATHS DIFFER" if (not same and exact) else ""), flush=True)
            if os.environ.get("FUZZ_ONLY"):
                ea = np.array([np.linalg.norm(alt[s] - want[s]) / max(np.linalg.norm(want[s]), 1e-30) for s in range(ns)])
                print("        %s=1 vs float64 oracle per slice:" % sw, " ".join("%.1e" % v for v in ea), flush=True)
P.release_plans()
print("flagged:", bad)
sys.exit(1 if bad else 0)
