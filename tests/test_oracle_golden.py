"""The oracle (oracle/pocs_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, parse_params, rel_l2
from oracle import pocs_oracle as orc


def test_schedule_matches_reference():
    g = load_golden("decay.npz")
    keys = sorted(k[:-5] for k in g.files if k.endswith("_meta"))
    assert len(keys) > 100
    for key in keys:
        sname, model, kind, p_min, niter = [str(v) for v in g[key + "_meta"]]
        p_min = "adaptive" if p_min == "adaptive" else float(p_min)
        with np.errstate(all="ignore"):
            tau = orc.threshold_schedule(model, int(niter), "FFT", 0.99, p_min, g["X0_" + sname], kind)
        want = g[key + "_tau"]
        assert np.asarray(tau).dtype == want.dtype, (key, model, kind)
        assert np.array_equal(np.asarray(tau), want, equal_nan=True), (key, model, kind, p_min, niter)


def test_threshold_operators_match_reference():
    g = load_golden("threshold.npz")
    ops = {"hard": orc.shrink_hard, "soft": orc.shrink_soft, "garrote": orc.shrink_garrote}
    for tn in ("r", "c", "cn", "z", "big", "c128"):
        tau = g["tau_" + tn][()]
        if tn != "c128":
            tau = tau.item()  # python scalars are "weak" under NEP 50, exactly what the reference was fed
        for on, op in ops.items():
            for xn in ("X", "Xd", "Xr"):
                with np.errstate(all="ignore"):
                    got = op(g[xn], tau, 0)
                want = g[f"{on}_{tn}_{xn}"]
                assert got.dtype == want.dtype
                assert np.array_equal(got, want, equal_nan=True), (on, tn, xn)
    for kind in ("soft", "hard", "garrote", "garotte", "soft-percentile", "hard-percentile",
                 "garrote-percentile", "garotte-percentile"):
        with np.errstate(all="ignore"):
            got = orc.apply_threshold(g["Xd"], 35.0 if "percentile" in kind else 0.8, kind=kind)
        assert np.array_equal(got, g["disp_" + kind], equal_nan=True), kind
    assert orc.apply_threshold(g["Xd"], 0.5, kind="nonsense") is None


def test_pocs_loop_matches_reference(golden_pocs):
    g = golden_pocs
    for name in [str(n) for n in g["names"]]:
        params = parse_params(g[name + "_params"])
        x, mask = g[name + "_x"], g[name + "_mask"]
        for suffix, xin in (("", x), ("_f64", x.astype(np.float64 if x.dtype.kind == "f" else np.complex128))):
            info = {}
            with np.errstate(all="ignore"):
                y = orc.pocs_slice(xin, mask, info=info, **params)
            want = g[name + "_out" + suffix]
            assert y.dtype == want.dtype, (name, suffix)
            # same NumPy ops in the same order -> identical bits under the same NumPy build;
            # leave a hair of slack for other BLAS/pocketfft builds
            assert rel_l2(y, want) <= (1e-12 if suffix else 2e-6), (name, suffix, rel_l2(y, want))
            n_ref = int(g[name + "_niter"][0 if suffix == "" else 1])
            assert info["niterations"] == n_ref, (name, suffix)
            c_ref = g[name + "_costs" + suffix]
            assert np.allclose(np.asarray(info["costs"], dtype=np.float64), c_ref, rtol=1e-4 if suffix == "" else 1e-9, atol=0)


def test_zero_slice_passthrough(golden_pocs):
    z = np.zeros((16, 16), np.complex64)
    info = {}
    y = orc.pocs_slice(z, np.ones((16, 16), np.uint8), niter=5, eps=0, info=info)
    assert y is z and info["niterations"] == 0
    assert int(golden_pocs["zero_niter"][0]) == 0 and not golden_pocs["zero_out"].any()


def test_fast_is_regular(golden_pocs):
    assert bool(golden_pocs["fpocs_equals_pocs"][0])
    x = golden_pocs["fpocs_x"].astype(np.complex128)
    m = golden_pocs["fpocs_mask"]
    p = parse_params(golden_pocs["fpocs_params"])
    a = orc.pocs_slice(x, m, **p)
    p["version"] = "regular"
    b = orc.pocs_slice(x, m, **p)
    assert np.array_equal(a, b)


def test_error_contract():
    tab = {str(r[0]): (str(r[1]), str(r[2])) for r in load_golden("errors.npz")["table"]}
    x = np.ones((8, 8), np.complex64)
    m = np.ones((8, 8), np.uint8)
    with pytest.raises(ValueError) as e:
        orc.pocs_slice(x, m * 2)
    assert str(e.value) == tab["mask_gt_1"][1]
    with pytest.raises(ValueError) as e:
        orc.pocs_slice(x, m, fwd=None, inv=None)
    assert str(e.value) == tab["no_transform"][1]
    with pytest.raises(ValueError) as e:
        orc.pocs_slice(x, m, transform_kind="HAAR")
    assert str(e.value) == tab["bad_kind"][1]


def test_cube_helper_casts_to_input_dtype():
    full, mask, obs = orc.synthetic_cube(16, 16, 3, 0.5)
    out = orc.pocs_cube(obs, mask, niter=4, eps=0, thresh_op="soft")
    assert out.dtype == obs.dtype and out.shape == obs.shape
