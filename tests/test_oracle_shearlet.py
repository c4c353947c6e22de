"""oracle/shearlet_oracle.py: (1) the SHEARLET branches of the reference, pinned by tests/golden/shearlet.npz (the reference's own
POCS_algorithm / get_threshold_decay run with the oracle's transform pair injected); (2) self-consistency of the transform
restated from the FFST tutorial (parity with PyShearlets itself is unpinned, see the oracle's header)."""
import ast

import numpy as np
import pytest

from conftest import load_golden, parse_params, rel_l2
from oracle import shearlet_oracle as so


@pytest.mark.parametrize("shape", [(32, 32), (33, 33), (24, 40), (33, 64), (64, 33), (17, 16), (128, 96)])
def test_parseval_frame(shape):
    psi = so.scales_shears_and_spectra(shape)
    J = so.number_of_scales(shape)
    assert psi.shape == shape + (1 + sum(2 ** (j + 2) for j in range(J)),)
    assert np.abs((psi ** 2).sum(-1) - 1).max() < 1e-13                       # sum_s Psi_s^2 = 1
    assert all(np.abs(psi[..., s]).max() > 0.5 for s in range(psi.shape[-1]))   # no empty element
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape)
    st = np.fft.ifft2(psi * np.fft.fft2(x)[..., None], axes=(0, 1))
    assert np.abs(st.imag).max() < 1e-13                                       # real shearlets
    assert np.abs(so.inverse_shearlet_transform(so.shearlet_transform(x, psi), psi) - x).max() < 1e-13
    xc = x + 1j * rng.standard_normal(shape)
    assert np.abs(so.inverse_shearlet_transform(so.shearlet_transform(xc, psi), psi) - xc).max() < 1e-13
    assert abs(np.sum(np.abs(so.shearlet_transform(x, psi)) ** 2) - np.sum(x ** 2)) < 1e-10 * np.sum(x ** 2)   # Parseval


def test_number_of_shearlets_matches_the_reference_formula():
    # POCS.py:305-311: 1 + sum_j 2^(j+2); configs[4] (2048 x 1024): J = 5 -> 125
    assert so.number_of_scales((2048, 1024)) == 5 and 1 + sum(2 ** (j + 2) for j in range(5)) == 125
    assert so.scales_shears_and_spectra((64, 64)).shape[-1] == 29


def test_shearlet_schedules():
    g = load_golden("shearlet.npz")
    psi = so.scales_shears_and_spectra(g["decay_x"].shape)
    coeffs = {"r": so.shearlet_transform(g["decay_x"], psi), "c": so.shearlet_transform(g["decay_xc"], psi)}
    keys = sorted(k[:-5] for k in g.files if k.startswith("decay") and k.endswith("_meta"))
    assert len(keys) == 16
    for key in keys:
        tag, model, kind, p_min = [str(v) for v in g[key + "_meta"]]
        tau = so.shearlet_schedule(model, 7, 0.99, ast.literal_eval(p_min), coeffs[tag], kind)
        want = g[key + "_tau"]
        assert np.allclose(np.broadcast_to(tau, want.shape), want, rtol=1e-12, atol=1e-14 * np.abs(want).max()), (key, model, kind, p_min)


def test_shearlet_pocs_runs():
    g = load_golden("shearlet.npz")
    for name in [str(n) for n in g["names"]]:
        params = parse_params(g[name + "_params"])
        x, mask, want = g[name + "_x"], g[name + "_mask"], g[name + "_out"]
        info = {}
        with np.errstate(all="ignore"):
            y = so.pocs_slice_shearlet(x, mask, so.scales_shears_and_spectra(x.shape), info=info, **params)
        assert y.shape == want.shape and np.iscomplexobj(y) == np.iscomplexobj(want)
        ok = np.isfinite(want) & np.isfinite(y)
        assert ok.mean() > 0.99 and rel_l2(y[ok], want[ok]) < 1e-10, (name, rel_l2(y[ok], want[ok]))
        assert info["niterations"] == int(g[name + "_niter"][0])
        if ok.all():
            assert np.isclose(info["costs"][-1], g[name + "_cost"][0], rtol=1e-6)


def test_product_host_side_matches():
    """The product's host code (spectra generator, SHEARLET schedule) against the oracle / the reference's recorded schedules."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from pseudo_3d_interpolation_amd.functions import shearlets
    for shape in [(32, 32), (33, 33), (24, 40), (33, 64), (17, 16)]:
        assert np.abs(shearlets.scalesShearsAndSpectra(shape) - so.scales_shears_and_spectra(shape)).max() < 1e-13
    assert shearlets.get_number_scales((2048, 1024)) == 5
    g = load_golden("shearlet.npz")
    psi = so.scales_shears_and_spectra(g["decay_x"].shape)
    coeffs = {"r": so.shearlet_transform(g["decay_x"], psi), "c": so.shearlet_transform(g["decay_xc"], psi)}
    for key in sorted(k[:-5] for k in g.files if k.startswith("decay") and k.endswith("_meta")):
        tag, model, kind, p_min = [str(v) for v in g[key + "_meta"]]
        tau = P.get_threshold_decay(model, 7, transform_kind="SHEARLET", p_max=0.99, p_min=ast.literal_eval(p_min), x_fwd=coeffs[tag], kind=kind)
        want = g[key + "_tau"]
        assert np.shape(tau) == want.shape and np.array_equal(tau, want), key
