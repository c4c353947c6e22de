#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF.

Build-container only:  the reference (read-only at /root/reference) is imported here, fed with
seeded inputs, and its outputs are stored as small .npz fixtures.  Nothing of the reference is
copied -- the fixtures are inputs + expected outputs (data).  The GPU box never runs this
script (it has no /root/reference); tests only read the .npz files.

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 \
        python3 tests/golden/make_golden.py

Interpreter used for the committed fixtures: /usr/bin/python3 3.10.12, NumPy 2.2.6 (pywt absent,
so the reference's own threshold_operator.py fallbacks are what ran).
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

from pseudo_3D_interpolation.functions import POCS as ref  # noqa: E402
from pseudo_3D_interpolation.functions import threshold_operator as ref_thr  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def field(nil, nxl, seed, real=False, dtype=None):
    """Few plane waves + a little noise (same recipe as the synthetic bench cube)."""
    rng = np.random.default_rng(seed)
    il = np.arange(nil)[:, None] / nil
    xl = np.arange(nxl)[None, :] / nxl
    acc = np.zeros((nil, nxl), dtype=np.complex128)
    for _ in range(5):
        k1 = rng.integers(-max(nil // 8, 1), max(nil // 8, 1) + 1)
        k2 = rng.integers(-max(nxl // 8, 1), max(nxl // 8, 1) + 1)
        amp = rng.standard_normal() + 1j * rng.standard_normal()
        acc += amp * np.exp(2j * np.pi * (k1 * il + k2 * xl))
    acc += 0.02 * (rng.standard_normal((nil, nxl)) + 1j * rng.standard_normal((nil, nxl)))
    if real:
        return acc.real.astype(dtype or np.float32)
    return acc.astype(dtype or np.complex64)


def trace_mask(nil, nxl, missing, seed=7):
    return (np.random.default_rng(seed).random((nil, nxl)) >= missing).astype(np.uint8)


# ------------------------------------------------------------------------------------------
def gen_decay():
    out = {}
    rng = np.random.default_rng(11)
    specs = {
        "s8": np.fft.fft2((rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))).astype(np.complex64)),
        "s64": np.fft.fft2(field(64, 64, 3)),
        "d16": np.fft.fft2(field(16, 24, 4, dtype=np.complex128)),
    }
    models = ["linear", "exponential", "exponential-2", "exponential-0.5", "inverse_proportional",
              "inverse_proportional-2", "inverse-proportional-x", "data-driven"]
    idx = 0
    for sname, X0 in specs.items():
        out[f"X0_{sname}"] = X0
        for model in models:
            for kind in ("values", "factors"):
                for p_min in (1e-3, "adaptive"):
                    for niter in (10, 37):
                        if kind == "factors" and p_min == "adaptive":
                            continue
                        if model == "data-driven" and kind == "factors":
                            continue  # complex data vs real bounds: still lexicographic but v may be empty
                        try:
                            tau = ref.get_threshold_decay(model, niter, "FFT", 0.99, p_min, X0, kind)
                        except Exception as e:  # noqa
                            print("skip", sname, model, kind, p_min, niter, type(e).__name__, e)
                            continue
                        key = f"case{idx:03d}"
                        out[key + "_tau"] = np.asarray(tau)
                        out[key + "_meta"] = np.array(
                            [sname, model, kind, str(p_min), str(niter)], dtype="U32")
                        idx += 1
    # niter == 1 -> 0/0 ramp
    with np.errstate(all="ignore"):
        tau = ref.get_threshold_decay("exponential", 1, "FFT", 0.99, 1e-3, specs["s8"], "values")
    out[f"case{idx:03d}_tau"] = np.asarray(tau)
    out[f"case{idx:03d}_meta"] = np.array(["s8", "exponential", "values", "0.001", "1"], dtype="U32")
    np.savez_compressed(os.path.join(HERE, "decay.npz"), **out)
    print("decay cases:", idx + 1)


def gen_threshold():
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((12, 9)) + 1j * rng.standard_normal((12, 9))).astype(np.complex64)
    X[0, 0] = 0
    X[3, 4] = 0.75  # exactly on a threshold value used below
    Xd = X.astype(np.complex128)
    Xr = rng.standard_normal((7, 5)).astype(np.float32)
    out = {"X": X, "Xd": Xd, "Xr": Xr}
    taus = {"r": 0.75, "c": 0.75 + 0.4j, "cn": 0.6 - 0.3j, "z": 0.0, "big": 10.0, "c128": np.complex128(0.9 + 0.2j)}
    ops = {"hard": ref_thr._hard_threshold, "soft": ref_thr._soft_threshold, "garrote": ref_thr._nn_garrote}
    with np.errstate(all="ignore"):
        for tn, tv in taus.items():
            out[f"tau_{tn}"] = np.asarray(tv)
            for on, op in ops.items():
                out[f"{on}_{tn}_X"] = op(X, tv, 0)
                out[f"{on}_{tn}_Xd"] = op(Xd, tv, 0)
                out[f"{on}_{tn}_Xr"] = op(Xr, tv, 0)
        # dispatcher incl. percentile kinds (POCS.py:61-102)
        for kind in ("soft", "hard", "garrote", "garotte", "soft-percentile", "hard-percentile",
                     "garrote-percentile", "garotte-percentile"):
            out[f"disp_{kind}"] = ref.threshold(Xd, 35.0 if "percentile" in kind else 0.8, kind=kind)
    np.savez_compressed(os.path.join(HERE, "threshold.npz"), **out)
    print("threshold entries:", len(out))


POCS_CASES = {
    # name: (nil, nxl, seed, real, missing, params)
    "fft_hard_exp": (64, 64, 21, False, 0.5, dict(niter=20, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)),
    "fft_real_in": (64, 64, 22, True, 0.5, dict(niter=20, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)),
    "fft_soft_lin": (32, 64, 23, False, 0.6, dict(niter=15, thresh_op="soft", thresh_model="linear", eps=0, p_max=0.9, p_min=1e-2)),
    "fft_garrote_exp2": (64, 32, 24, False, 0.4, dict(niter=12, thresh_op="garrote", thresh_model="exponential-2", eps=0, p_max=0.99, p_min=1e-3)),
    "fft_sqrt_decay": (32, 32, 25, False, 0.5, dict(niter=10, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3, sqrt_decay=True)),
    "fft_alpha08": (32, 32, 26, False, 0.5, dict(niter=10, thresh_op="hard", thresh_model="exponential", eps=0, alpha=0.8, p_max=0.99, p_min=1e-3)),
    "fft_invprop": (32, 32, 27, False, 0.5, dict(niter=10, thresh_op="hard", thresh_model="inverse_proportional-2", eps=0)),
    "fft_factors": (32, 32, 28, False, 0.5, dict(niter=10, thresh_op="hard", thresh_model="linear", decay_kind="factors", eps=0, p_max=200.0, p_min=1.0)),
    "fft_datadriven": (32, 32, 29, False, 0.5, dict(niter=10, thresh_op="hard", thresh_model="data-driven", eps=0, p_max=0.99, p_min=1e-3)),
    "apocs_doc": (64, 64, 30, False, 0.5, dict(niter=20, thresh_op="hard", thresh_model="exponential-1", eps=0, alpha=0.75, p_max=0.99, p_min="adaptive", version="adaptive")),
    "apocs_soft": (32, 32, 31, False, 0.5, dict(niter=10, thresh_op="soft", thresh_model="exponential", eps=0, alpha=0.9, p_max=0.99, p_min=1e-3, version="adaptive")),
    "fpocs": (32, 32, 32, False, 0.5, dict(niter=10, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3, version="fast")),
    "early_exit": (64, 64, 33, False, 0.3, dict(niter=60, thresh_op="hard", thresh_model="exponential", eps=1e-9, p_max=0.99, p_min=1e-3)),
    "rect_90x50": (90, 50, 34, False, 0.5, dict(niter=10, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)),
    "rect_48x20_real": (48, 20, 35, True, 0.5, dict(niter=8, thresh_op="soft", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)),
    "prime_31x17": (31, 17, 36, False, 0.4, dict(niter=6, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)),
    "tiny_8x8": (8, 8, 37, False, 0.5, dict(niter=5, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)),
    "niter1": (16, 16, 38, False, 0.5, dict(niter=1, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)),
    "hard_pct": (32, 32, 39, False, 0.5, dict(niter=8, thresh_op="hard-percentile", thresh_model="linear", decay_kind="factors", eps=0, p_max=99.0, p_min=5.0)),
}


def run_ref(x, mask, params):
    info = {}
    path = os.path.join("/tmp", f"golden_{os.getpid()}.out")
    if os.path.exists(path):
        os.remove(path)
    with np.errstate(all="ignore"):
        y = ref.POCS_algorithm(x, mask, transform=np.fft.fft2, itransform=np.fft.ifft2,
                               transform_kind="FFT", results_dict=info, path_results=path, **params)
    with open(path) as f:
        parts = f.read().strip().split(";")
    os.remove(path)
    costs = np.array([float(p) for p in parts[2:]], dtype=np.float64)
    return y, int(info["niterations"]), costs


def gen_pocs():
    out = {}
    names = []
    for name, (nil, nxl, seed, real, missing, params) in POCS_CASES.items():
        x64 = field(nil, nxl, seed, real)
        m = trace_mask(nil, nxl, missing, seed)
        xo = x64 * m
        xd = xo.astype(np.float64 if real else np.complex128)
        y_s, n_s, c_s = run_ref(xo, m, params)     # single-precision fed (what the pipeline does)
        y_d, n_d, c_d = run_ref(xd, m, params)     # double fed ("truth" for tolerances)
        out[f"{name}_x"] = xo
        out[f"{name}_mask"] = m
        out[f"{name}_out"] = y_s
        out[f"{name}_out_f64"] = y_d
        out[f"{name}_niter"] = np.array([n_s, n_d])
        out[f"{name}_costs"] = c_s
        out[f"{name}_costs_f64"] = c_d
        out[f"{name}_params"] = np.array([f"{k}={v!r}" for k, v in params.items()], dtype="U64")
        names.append(name)
        rel = np.linalg.norm(y_s - y_d) / np.linalg.norm(y_d)
        print(f"{name:18s} {nil}x{nxl} out {y_s.dtype}/{y_d.dtype} niter {n_s}/{n_d} single-vs-double rel-L2 {rel:.2e}")
    # all-zero slice: handed back untouched, niterations == 0 (POCS.py:515-521)
    z = np.zeros((16, 16), np.complex64)
    y, n, c = run_ref(z, trace_mask(16, 16, 0.5), dict(niter=5, eps=0))
    out["zero_out"] = y
    out["zero_niter"] = np.array([n])
    out["zero_costs"] = c
    out["names"] = np.array(names, dtype="U32")
    # FPOCS == POCS bit for bit
    nil, nxl, seed, real, missing, params = POCS_CASES["fpocs"]
    xd = (field(nil, nxl, seed) * trace_mask(nil, nxl, missing, seed)).astype(np.complex128)
    p2 = dict(params); p2["version"] = "regular"
    a, _, _ = run_ref(xd, trace_mask(nil, nxl, missing, seed), params)
    b, _, _ = run_ref(xd, trace_mask(nil, nxl, missing, seed), p2)
    out["fpocs_equals_pocs"] = np.array([np.array_equal(a, b)])
    np.savez_compressed(os.path.join(HERE, "pocs.npz"), **out)


def gen_errors():
    """Error behaviour of the per-slice contract (POCS.py:488-503)."""
    x = field(8, 8, 1)
    m = trace_mask(8, 8, 0.5)
    res = []
    for label, kw in [
        ("mask_gt_1", dict(mask=m * 2, transform=np.fft.fft2, itransform=np.fft.ifft2, transform_kind="FFT")),
        ("no_transform", dict(mask=m, transform=None, itransform=None, transform_kind="FFT")),
        ("bad_kind", dict(mask=m, transform=np.fft.fft2, itransform=np.fft.ifft2, transform_kind="HAAR")),
        ("shearlet_no_psi", dict(mask=m, transform=np.fft.fft2, itransform=np.fft.ifft2, transform_kind="SHEARLET")),
    ]:
        try:
            ref.POCS_algorithm(x, **kw)
            res.append((label, "none", ""))
        except Exception as e:  # noqa
            res.append((label, type(e).__name__, str(e)))
    np.savez_compressed(os.path.join(HERE, "errors.npz"), table=np.array(res, dtype="U160"))
    for r in res:
        print(r)


if __name__ == "__main__":
    gen_decay()
    gen_threshold()
    gen_pocs()
    gen_errors()
