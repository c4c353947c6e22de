"""GPU parity of the SHEARLET variant against oracle/shearlet_oracle.py and the reference's recorded
SHEARLET runs (tests/golden/shearlet.npz).  What is pinned and what is not: see the oracle's header."""
import numpy as np
import pytest

from conftest import load_golden, parse_params, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    from pseudo_3d_interpolation_amd import _ffi
    _ffi.lib()
    return _ffi


@pytest.fixture(scope="module")
def so():
    from oracle import shearlet_oracle
    return shearlet_oracle


@pytest.mark.parametrize("shape", [(32, 32), (64, 64), (32, 64), (24, 40), (33, 31), (128, 256)])
def test_transform_and_inverse(ffi, so, shape):
    from pseudo_3d_interpolation_amd.functions import shearlets
    psi = shearlets.scalesShearsAndSpectra(shape)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2,) + shape) + 1j * rng.standard_normal((2,) + shape)).astype(np.complex64)
    with ffi.ShearletPlan(psi, max_slices=2) as plan:
        st = plan.transform(x)
        assert st.shape == (2,) + shape + (psi.shape[-1],)
        want = so.shearlet_transform(x[0].astype(np.complex128), psi)
        assert rel_l2(st[0], want) <= 2e-6
        assert np.abs(st[0] - want).max() <= 5e-6 * np.abs(want).max()
        back = plan.inverse(st)
        assert rel_l2(back, x) <= 2e-6                                    # Parseval frame: perfect reconstruction
        c = (rng.standard_normal(shape + (psi.shape[-1],)) + 1j * rng.standard_normal(shape + (psi.shape[-1],))).astype(np.complex64)
        assert rel_l2(plan.inverse(c), so.inverse_shearlet_transform(c.astype(np.complex128), psi)) <= 2e-6
        stats = plan.stats(x.real.astype(np.float32))
        ref = so.shearlet_transform(x[1].real.astype(np.float64), psi)
        assert np.allclose(stats[1, :, 0], ref.max(axis=(0, 1)), rtol=2e-5, atol=1e-5 * np.abs(ref).max())
        assert np.all(stats[1, :, 1] == 0)
        assert np.allclose(stats[1, :, 2], np.abs(ref).max(axis=(0, 1)), rtol=2e-5)
        assert np.allclose(stats[1, :, 4], (ref ** 2).sum(axis=(0, 1)), rtol=1e-4)


@pytest.mark.parametrize("shape", [(512, 512), (1024, 512)])
def test_statistics_of_a_float32_cube_through_the_paired_passes(ffi, so, shape, monkeypatch):
    """float32 cubes on symmetric spectra take their coefficient statistics (the schedule's input) from the passes of the loop --
    support rows, Hermitian half, two columns per transform -- instead of nsh dense transforms; both routes agree with each other and
    with the float64 oracle to float32 rounding."""
    from pseudo_3d_interpolation_amd.functions import shearlets
    from oracle import pocs_oracle as po
    psi = shearlets.scalesShearsAndSpectra(shape, dtype=np.float32)
    x = np.stack([po.synthetic_slice(*shape, 70 + s, real=True) for s in range(3)]).astype(np.float32)
    with ffi.ShearletPlan(psi, max_slices=3) as plan:
        assert plan.paired
        fast = plan.stats(x)
    monkeypatch.setenv("P3D_SHEARLET_NO_PAIR", "1")
    with ffi.ShearletPlan(psi, max_slices=3) as plan:
        assert not plan.paired
        dense = plan.stats(x)
    monkeypatch.delenv("P3D_SHEARLET_NO_PAIR")
    ref = so.shearlet_transform(x[2].astype(np.float64), so.scales_shears_and_spectra(shape, contiguous=False))
    top = np.abs(ref).max()
    for st in (fast, dense):
        assert np.allclose(st[2, :, 0], ref.max(axis=(0, 1)), rtol=2e-5, atol=1e-6 * top)
        assert np.all(st[:, :, 1] == 0)
        assert np.allclose(st[2, :, 2], np.abs(ref).max(axis=(0, 1)), rtol=2e-5, atol=1e-6 * top)
        assert np.allclose(st[2, :, 4], (ref ** 2).sum(axis=(0, 1)), rtol=1e-4)
    assert np.allclose(fast[..., 0], dense[..., 0], rtol=1e-5, atol=1e-6 * top)
    assert np.allclose(fast[..., 2], dense[..., 2], rtol=1e-5, atol=1e-6 * top)
    assert np.allclose(fast[..., 3], dense[..., 3], rtol=1e-2, atol=1e-6 * top)     # min |c|: a sample next to a zero crossing
    assert np.allclose(fast[..., 4], dense[..., 4], rtol=1e-5, atol=1e-4 * dense[..., 4].max())     # (energies span 4 decades: float32 noise of the strong ones)


def _case(so, shape, seed, complex_=False, missing=0.5, nslices=2, **kw):
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    psi = shearlets.scalesShearsAndSpectra(shape)
    cube = np.stack([po.synthetic_slice(shape[0], shape[1], seed + i, real=not complex_) for i in range(nslices)])
    mask = po.synthetic_mask(shape[0], shape[1], missing)
    cube = (cube * mask).astype(np.complex64 if complex_ else np.float32)
    infos, res = [], []
    want = so.pocs_cube_shearlet(cube.astype(np.complex128 if complex_ else np.float64), mask, psi, infos=infos, **kw)
    got = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res, **kw)
    return got, want, res, infos


@pytest.mark.parametrize("kw", [
    dict(thresh_op="soft", thresh_model="linear", niter=8, p_max=0.9, p_min=0.05, eps=0.0),
    dict(thresh_op="soft", thresh_model="exponential", niter=10, p_max=0.99, p_min=1e-2, eps=0.0),
    dict(thresh_op="hard", thresh_model="exponential", niter=8, p_max=0.99, p_min=0.05, eps=0.0),
    dict(thresh_op="garrote", thresh_model="exponential-2", niter=6, p_max=0.8, p_min=0.1, eps=0.0, alpha=0.8),
    dict(thresh_op="soft", thresh_model="linear", niter=6, p_max=0.9, p_min=0.1, eps=0.0, version="adaptive", alpha=0.9),
    dict(thresh_op="soft", thresh_model="linear", niter=6, p_max=0.5, p_min=0.01, eps=0.0, decay_kind="factors"),
    dict(thresh_op="soft", thresh_model="exponential", niter=6, p_max=0.99, p_min="adaptive", eps=0.0),
    dict(thresh_op="soft", thresh_model="inverse_proportional-2", niter=6, eps=0.0),
    dict(thresh_op="soft", thresh_model="linear", niter=6, p_max=0.9, p_min=0.1, eps=0.0, sqrt_decay=True),
])
@pytest.mark.parametrize("complex_", [False, True])
def test_shearlet_pocs_vs_oracle(so, kw, complex_):
    if complex_ and (kw["thresh_op"] == "garrote" or kw.get("p_min") == "adaptive"):
        pytest.skip("complex tau with garrote amplifies (see test_gpu_wavelet); 'adaptive' p_min is real-only in practice")
    got, want, res, infos = _case(so, (48, 64), 5, complex_=complex_, **kw)
    assert got.dtype == (np.complex64 if complex_ else np.float32)
    tol = 1e-5 if kw["thresh_op"] == "soft" else 2e-4   # hard / garrote: decision flips at float32 ties
    for s in range(got.shape[0]):
        ok = np.isfinite(want[s]) & np.isfinite(got[s])
        assert ok.mean() > 0.99
        assert rel_l2(got[s][ok], want[s][ok]) <= tol, (s, rel_l2(got[s][ok], want[s][ok]))
        assert res[s]["niterations"] == infos[s]["niterations"]
        if ok.all():
            np.testing.assert_allclose(res[s]["costs"], infos[s]["costs"], rtol=2e-2, atol=1e-12)


def test_shearlet_early_exit_and_empty_slice(so):
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    shape = (32, 32)
    psi = shearlets.scalesShearsAndSpectra(shape)
    cube = np.stack([po.synthetic_slice(*shape, 3, real=True), np.zeros(shape), po.synthetic_slice(*shape, 4, real=True)])
    cube = np.real(cube).astype(np.float32)
    mask = po.synthetic_mask(*shape, 0.4)
    cube *= mask
    kw = dict(thresh_op="soft", thresh_model="exponential", niter=40, p_max=0.99, p_min=1e-2, eps=1e-5)
    infos, res = [], []
    want = so.pocs_cube_shearlet(cube.astype(np.float64), mask, psi, infos=infos, **kw)
    got = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res, batch_slices=2, **kw)
    assert res[1]["niterations"] == 0 and not got[1].any()
    for s in (0, 2):
        assert 3 < res[s]["niterations"] < 40
        assert res[s]["niterations"] == infos[s]["niterations"]
        assert rel_l2(got[s], want[s]) <= 1e-5


def test_shearlet_golden_runs(so):
    """The reference's POCS_algorithm (float64, oracle's transform pair injected) on the runs of tests/golden/shearlet.npz."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from pseudo_3d_interpolation_amd.functions import shearlets
    g = load_golden("shearlet.npz")

    def shearletTransformSpect(*a, **k):   # stand-ins for FFST's pair: recognised by name, never called
        raise AssertionError("the host transform must not be called")

    def inverseShearletTransformSpect(*a, **k):
        raise AssertionError("the host transform must not be called")

    for name in [str(n) for n in g["names"]]:
        prm = parse_params(g[name + "_params"])
        x, mask, want = g[name + "_x"], g[name + "_mask"], g[name + "_out"]
        xin = x.astype(np.complex64 if np.iscomplexobj(x) else np.float32)
        res = {}
        got = P.POCS_algorithm(xin, mask, auxiliary_data=shearlets.scalesShearsAndSpectra(x.shape), transform=shearletTransformSpect,
                               itransform=inverseShearletTransformSpect, transform_kind="SHEARLET", results_dict=res, **prm)
        assert got.dtype == xin.dtype and got.shape == want.shape
        ok = np.isfinite(want) & np.isfinite(got)
        assert ok.mean() > 0.99
        tol = 1e-5 if prm.get("thresh_op", "hard") == "soft" else 2e-4
        assert rel_l2(got[ok], want[ok]) <= tol, (name, rel_l2(got[ok], want[ok]))
        if ok.all():
            assert res["niterations"] == int(g[name + "_niter"][0]), name
    with pytest.raises(ValueError):
        P.POCS_algorithm(xin, mask, transform=shearletTransformSpect, itransform=inverseShearletTransformSpect, transform_kind="SHEARLET")


def test_shearlet_config4_slice_at_its_own_size(so, monkeypatch):
    """BASELINE configs[4]'s slice as stated: 2048 x 1024 (iline x xline), J = 5 scales = 125 shearlets, 80 % missing, hard threshold,
    exponential decay -- the first iterations of the schedule against the oracle (its multi-threaded real-transform form, which is
    held to the plain oracle loop on a small slice first), plus the size-independent properties on the device: observed traces
    come back bit-exact (alpha = 1), a slice processed alone gives the same bits as inside a batch.  (The frame's conventions
    themselves are parity-unpinned: FFST is absent from the image; see the oracle's header.)"""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    kw = dict(thresh_op="hard", thresh_model="exponential", niter=4, p_max=0.99, p_min=0.05)
    # the fast form of the oracle is the oracle: same loop, same numbers
    small = (64, 32)
    psi_s = shearlets.scalesShearsAndSpectra(small)
    m_s = po.synthetic_mask(*small, 0.5)
    x_s = (po.synthetic_slice(*small, 7, real=True) * m_s).astype(np.float64)
    a = so.pocs_slice_shearlet(x_s, m_s, psi_s, eps=0.0, **kw)
    b = so.pocs_slice_shearlet_real(x_s, m_s, psi_s, **kw)
    assert rel_l2(b, a) <= 1e-12, rel_l2(b, a)

    shape = (2048, 1024)
    psi = shearlets.scalesShearsAndSpectra(shape)
    assert psi.shape == shape + (125,)
    # the oracle runs on ITS OWN spectra (written independently of the product's generator); the two frames agree to rounding
    psi_orc = so.scales_shears_and_spectra(shape, contiguous=False)
    assert psi_orc.shape == psi.shape and max(float(np.abs(psi[..., i] - psi_orc[..., i]).max()) for i in range(125)) < 1e-13
    mask = po.synthetic_mask(*shape, 0.8)
    cube = np.stack([po.synthetic_slice(*shape, 40 + i, real=True) for i in range(2)]) * mask
    cube = cube.astype(np.float32)
    res = []
    got = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res, eps=0.0, **kw)
    assert got.dtype == np.float32 and got.shape == cube.shape and [r["niterations"] for r in res] == [4, 4]
    keep = mask.astype(bool)
    for s in range(2):
        assert np.array_equal(got[s][keep], cube[s][keep])                       # observed traces are handed back exactly
    alone = P.pocs_cube(cube[1:2], mask, transform_kind="SHEARLET", auxiliary_data=psi, eps=0.0, **kw)
    assert np.array_equal(alone[0], got[1])                                      # batching is transparent
    # float32 cubes send two columns through one transform (p3d_col_shear.hpp); the general column pass (P3D_SHEARLET_NO_PAIR=1) agrees
    # with it to float32 rounding plus the odd hard-threshold flip, and its persistent form (2048-point columns) with its one-launch
    # form (P3D_NO_COLPIPE=1) bit for bit.  (The switches are read when a plan is created.)
    P.release_plans()
    monkeypatch.setenv("P3D_SHEARLET_NO_PAIR", "1")
    general = P.pocs_cube(cube[1:2], mask, transform_kind="SHEARLET", auxiliary_data=psi, eps=0.0, **kw)
    P.release_plans()
    monkeypatch.setenv("P3D_NO_COLPIPE", "1")
    plain = P.pocs_cube(cube[1:2], mask, transform_kind="SHEARLET", auxiliary_data=psi, eps=0.0, **kw)
    P.release_plans()
    monkeypatch.delenv("P3D_NO_COLPIPE")
    monkeypatch.delenv("P3D_SHEARLET_NO_PAIR")
    assert np.array_equal(plain[0], general[0])
    print(f"configs[4] slice: two-columns-per-transform vs general column pass rel-L2 {rel_l2(got[1], general[0]):.3e}")
    assert rel_l2(got[1], general[0]) <= 2e-4
    info = {}
    want = so.pocs_slice_shearlet_real(cube[0].astype(np.float64), mask, psi_orc, info=info, **kw)
    err = rel_l2(got[0], want)
    truth = po.synthetic_slice(*shape, 40, real=True)
    print(f"configs[4] slice 2048x1024x125 shearlets, 4 iterations (hard): device-vs-oracle rel-L2 {err:.3e}; "
          f"interpolation error {rel_l2(cube[0], truth):.3f} -> {rel_l2(got[0], truth):.3f}")
    assert err <= 2e-4, err                                                      # hard threshold: float32 decision flips (layer B)
    np.testing.assert_allclose(res[0]["costs"], info["costs"], rtol=2e-2, atol=1e-12)
    P.release_plans()


def test_shearlet_floor_step_at_config4_size(ffi, so):
    """Layer B (decision level) for BASELINE configs[4] in the regime its 100-iteration schedule ends in: 2048 x 1024, 125 shearlets,
    80 % missing, hard threshold, tau_s = p_min * peak_s = 1e-3 of every shearlet's peak coefficient -- the incoherent floor, where
    tens of thousands of coefficients crowd around the threshold.  The iterate is DEVICE-produced (60 iterations of the
    100-iteration schedule); from it the device takes one step with the final thresholds and the oracle (float64, on its own spectra)
    takes the same step.  The two may differ only by keep/zero decisions inside the float32 tie band around tau
    (||c| - tau_s| <= 2e-6 max|c_s|): the synthesis of a Parseval frame does not amplify, so
        || device - oracle ||_2  <=  || coefficients in the band ||_2  +  rounding (3e-6 || oracle ||_2)."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    shape, K, late = (2048, 1024), 100, 60
    psi = shearlets.scalesShearsAndSpectra(shape, dtype=np.float32)
    psi_orc = so.scales_shears_and_spectra(shape, contiguous=False)
    mask = po.synthetic_mask(*shape, 0.8)
    maskf = mask.astype(np.float32)
    x = (po.synthetic_slice(*shape, 41, real=True) * mask).astype(np.float32)[None]
    with ffi.ShearletPlan(psi, max_slices=1) as plan:
        stats = plan.stats(x)
        tau = P._shearlet_schedule_from_stats(stats, shape, "exponential", K, 0.99, 1e-3, "values")      # (1, K, nsh)
        assert tau.shape == (1, K, 125) and np.allclose(tau[0, -1], 1e-3 * stats[0, :, 0], rtol=1e-12)
        it_late = plan.run(x, maskf, tau[:, :late], late, thresh_op="hard")[0]                              # device iterate after 60 iterations
        assert np.isfinite(it_late).all()
        t_floor = tau[:, -1:]                                                                             # (1, 1, nsh): the schedule's last entry
        dev = plan.run(it_late, maskf, t_floor, 1, thresh_op="hard")[0][0]                                # one step with x_obs := the iterate
    prev = it_late[0].astype(np.float64)
    want, st, shr = so.shearlet_step_real(prev, prev, mask, psi_orc, t_floor[0, 0].real, "hard")
    peak = np.abs(st).max(axis=(1, 2))
    band = np.abs(np.abs(st) - t_floor[0, 0].real[:, None, None]) <= 2e-6 * peak[:, None, None]
    band_norm = float(np.sqrt(np.sum(st[band] ** 2)))
    kept = float(np.count_nonzero(shr)) / st.size
    diff = float(np.linalg.norm(dev - want))
    print(f"configs[4] floor step 2048x1024x125: {int(band.sum())} of {st.size} coefficients in the tie band ({band.mean():.2e}), kept "
          f"{kept:.3f}; ||device - oracle|| = {diff:.3e} (rel {diff / np.linalg.norm(want):.2e}), band energy {band_norm:.3e}, "
          f"rounding allowance {3e-6 * np.linalg.norm(want):.3e}")
    assert band.mean() < 0.01, "the tie band should be a sliver of the coefficients"
    assert 0.001 < kept < 0.9                                                                            # a threshold inside the floor, not above / below everything
    assert diff <= band_norm + 3e-6 * np.linalg.norm(want), (diff, band_norm)
    keep = mask.astype(bool)
    assert np.array_equal(dev[keep], it_late[0][keep])                                                   # observed positions handed back exactly


@pytest.mark.parametrize("shape,real,op", [((256, 128), True, "hard"), ((64, 64), False, "soft"), ((2048, 128), True, "hard"), ((128, 2048), True, "garrote"),
                                            ((1024, 256), False, "hard"), ((4096, 64), True, "soft")])
def test_skipping_rows_off_a_shearlets_support_changes_nothing(ffi, shape, real, op, monkeypatch):
    """A shearlet's spectrum vanishes on most rows of the frequency plane; the fused passes neither compute, store nor read the
    8-row groups on which it does (ShearArgs::sup).  Those rows carry only zeros through the iteration, so the results, cost sums
    and iteration counts are those of the dense passes (P3D_SHEARLET_NO_SUPPORT=1) -- bit for bit for the cubes (the sums add
    per-block partial sums atomically: last bits vary run to run).  Also reports the share of (shearlet, row group) pairs touched."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    nil, nxl = shape
    psi = shearlets.scalesShearsAndSpectra(shape, dtype=np.float32)
    mask = po.synthetic_mask(nil, nxl, 0.6)
    cube = np.stack([po.synthetic_slice(nil, nxl, 20 + s, real=real) for s in range(3)]) * mask
    cube = cube.astype(np.float32 if real else np.complex64)
    cube[1] = 0
    kw = dict(transform_kind="SHEARLET", auxiliary_data=psi, niter=5, thresh_op=op, thresh_model="exponential", p_max=0.99, p_min=1e-2, eps=1e-12,
              batch_slices=2)

    def run():
        P.release_plans()
        res = []
        out = P.pocs_cube(cube, mask, results=res, **kw)
        with ffi.ShearletPlan(psi, max_slices=1) as plan:
            frac = plan.row_group_fraction
        return out, res, frac

    a, res_a, frac = run()
    monkeypatch.setenv("P3D_SHEARLET_NO_SUPPORT", "1")
    b, res_b, dense = run()
    monkeypatch.delenv("P3D_SHEARLET_NO_SUPPORT")
    P.release_plans()
    print(f"shearlet frame {nil}x{nxl}x{psi.shape[-1]}: {frac:.3f} of the (shearlet, 8-row group) pairs hold a non-zero spectrum sample")
    assert dense == 1.0 and 0.05 < frac < 0.9
    assert np.array_equal(a, b)
    assert [r["niterations"] for r in res_a] == [r["niterations"] for r in res_b]
    for ra, rb in zip(res_a, res_b):
        np.testing.assert_allclose(ra["costs"], rb["costs"], rtol=1e-9)


@pytest.mark.parametrize("shape", [(512, 256), (1024, 512), (2048, 128), (4096, 64), (256, 256)])
@pytest.mark.parametrize("op", ["soft", "hard"])
def test_two_columns_per_transform_against_the_general_column_pass(so, shape, op, monkeypatch):
    """float32 cubes, symmetric spectra: the column pass packs two columns into one complex transform (real coefficients: Z = W_A +
    i W_B).  Against the general pass (P3D_SHEARLET_NO_PAIR=1) it differs by float32 rounding (soft: <= 2e-6) plus, for the hard
    operator, the odd decision in the tie band; both sit equally close to the float64 oracle.  Columns below 512 points and spectra
    that are not symmetric keep the general pass (the latter checked here with a deliberately lopsided frame)."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    nil, nxl = shape
    psi = shearlets.scalesShearsAndSpectra(shape, dtype=np.float32)
    mask = po.synthetic_mask(nil, nxl, 0.6)
    cube = (np.stack([po.synthetic_slice(nil, nxl, 30 + s, real=True) for s in range(2)]) * mask).astype(np.float32)
    kw = dict(transform_kind="SHEARLET", niter=4, thresh_op=op, thresh_model="exponential", p_max=0.99, p_min=1e-2, eps=0.0)

    def run(spectra):
        P.release_plans()
        return P.pocs_cube(cube, mask, auxiliary_data=spectra, **kw)

    paired = run(psi)
    monkeypatch.setenv("P3D_SHEARLET_NO_PAIR", "1")
    general = run(psi)
    monkeypatch.delenv("P3D_SHEARLET_NO_PAIR")
    psi_o = so.scales_shears_and_spectra(shape, contiguous=False)
    want = np.stack([so.pocs_slice_shearlet_real(c.astype(np.float64), mask, psi_o, thresh_op=op, thresh_model="exponential", niter=4, p_max=0.99, p_min=1e-2)
                     for c in cube])
    d = rel_l2(paired, general)
    assert np.isfinite(paired).all() and np.isfinite(general).all()
    print(f"{nil}x{nxl} {op}: paired-vs-general {d:.2e}; vs oracle: paired {rel_l2(paired, want):.2e}, general {rel_l2(general, want):.2e}")
    assert d <= (2e-6 if op == "soft" else 2e-4)
    if np.isfinite(want).all():   # (a very oblong grid leaves some fine-scale shearlets without a single sample: the reference's schedule is NaN there)
        assert rel_l2(paired, want) <= max(3 * rel_l2(general, want), 1e-5 if op == "soft" else 2e-4)
    # a frame that is NOT symmetric (real slices then have complex coefficients, of which the reference keeps the real part): the plan
    # notices and keeps the general pass -- with or without the switch the same bits
    lop = np.array(psi, copy=True)
    r, c, e = np.argwhere(psi[1:nil // 2] != 0)[0]     # one non-zero sample whose mirror image is left alone
    lop[1 + r, c, e] *= 0.5
    a = run(lop)
    monkeypatch.setenv("P3D_SHEARLET_NO_PAIR", "1")
    b = run(lop)
    monkeypatch.delenv("P3D_SHEARLET_NO_PAIR")
    P.release_plans()
    assert np.array_equal(a, b)


# ---- the SHEARLET loop in the reference's double precision (p3d_shearlet64.hip) ----------------------------------------------------------
@pytest.mark.parametrize("shape,complex_", [((48, 64), False), ((48, 64), True), ((33, 31), False), ((40, 24), True), ((128, 96), False), ((150, 240), False),
                                            ((64, 128), False), ((128, 64), True), ((256, 128), False)])   # (the last three: both extents on the register engine)
@pytest.mark.parametrize("kw", [
    dict(niter=6, thresh_op="soft", thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2),
    dict(niter=5, thresh_op="garrote", thresh_model="linear", eps=0.0, p_max=0.9, p_min=0.05, alpha=0.8, version="adaptive"),
    dict(niter=30, thresh_op="hard", thresh_model="exponential", eps=1e-4, p_max=0.99, p_min=1e-2),
    dict(niter=5, thresh_op="soft", thresh_model="exponential", eps=0.0, p_max=0.99, p_min="adaptive"),
    dict(niter=5, thresh_op="soft", thresh_model="inverse_proportional-2", eps=0.0, sqrt_decay=True),
])
def test_shearlet_loop_in_the_reference_precision(so, shape, complex_, kw):
    """complex128 / float64 cubes run the SHEARLET loop in double precision (np.fft inside FFST computes in double and POCS_algorithm never
    narrows: POCS.py:589-619), complex64 / float32 cubes on request: 1e-10 against the double-fed oracle where the float32 kernels hold
    1e-5 ... 2e-4, the same iteration counts under the early exit, the same costs, an all-zero slice untouched."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    if complex_ and (kw["thresh_op"] == "garrote" or kw.get("p_min") == "adaptive"):
        pytest.skip("complex tau with garrote amplifies rounding noise (see test_gpu_wavelet); 'adaptive' p_min is real-only in practice")
    psi = shearlets.scalesShearsAndSpectra(shape)
    mask = po.synthetic_mask(shape[0], shape[1], 0.5)
    cube = np.stack([po.synthetic_slice(shape[0], shape[1], 20 + s, real=not complex_) for s in range(3)])
    if kw.get("p_min") == "adaptive":
        cube = cube + 2.0   # (a positive mean: the signed maximum of the low-pass coefficients, which scales the schedule, is then positive)
    cube = (cube * mask).astype(np.complex128 if complex_ else np.float64)
    cube[1] = 0
    res, infos = [], []
    got = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res, **kw)
    want = so.pocs_cube_shearlet(cube, mask, psi, infos=infos, **kw)
    assert got.dtype == cube.dtype and not got[1].any()
    compared = 0
    for s in (0, 2):
        if not np.isfinite(want[s]).all():
            # a slice whose low-pass coefficients are all negative has a negative signed maximum there: the exponential schedule takes the
            # logarithm of a negative ratio and the reference's whole slice turns NaN (POCS.py:340-341) -- nothing to compare with
            continue
        compared += 1
        assert res[s]["niterations"] == infos[s]["niterations"], (s, res[s]["niterations"], infos[s]["niterations"])
        assert rel_l2(got[s], want[s]) <= 1e-10, (s, rel_l2(got[s], want[s]))
        np.testing.assert_allclose(res[s]["costs"], infos[s]["costs"], rtol=1e-6, atol=1e-24)
    assert compared >= 1
    narrow = cube.astype(np.complex64 if complex_ else np.float32)
    got32 = P.pocs_cube(narrow, mask, transform_kind="SHEARLET", auxiliary_data=psi, precision="reference", **kw)
    want32 = so.pocs_cube_shearlet(narrow.astype(cube.dtype), mask, psi, **kw)
    assert got32.dtype == narrow.dtype
    if kw["thresh_op"] != "hard":     # (the result is cast back to float32: 6e-8 per sample; a hard threshold may flip a decision of the narrowed input)
        for s in (0, 2):
            if np.isfinite(want32[s]).all():
                assert rel_l2(got32[s], want32[s]) <= 2e-7
    # batching is transparent, and the float32 kernels are still what a float32 cube gets by default
    alone = P.pocs_cube(cube[2:3], mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
    assert np.array_equal(alone[0], got[2])
    split = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, batch_slices=2, **kw)
    assert np.array_equal(split, got)


@pytest.mark.parametrize("shape,fused", [((40, 56), False), ((64, 128), True)])
def test_shearlet_plan64_statistics_and_errors(ffi, so, shape, fused):
    """The double-precision plan by itself: statistics of the coefficients against the oracle's transform, argument checks."""
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    psi = shearlets.scalesShearsAndSpectra(shape)
    nsh = psi.shape[2]
    with ffi.ShearletPlan64(psi, max_slices=2) as plan:
        assert plan.fused == fused
        for real in (True, False):
            x = np.stack([po.synthetic_slice(*shape, 3 + s, real=real) for s in range(2)]).astype(np.float64 if real else np.complex128)
            st = plan.stats(x)
            assert st.shape == (2, nsh, 5)
            for s in range(2):
                c = so.shearlet_transform(x[s], psi)
                peak = np.max(c, axis=(0, 1))
                np.testing.assert_allclose(st[s, :, 0] + 1j * st[s, :, 1], peak, rtol=1e-12, atol=1e-15)
                np.testing.assert_allclose(st[s, :, 2], np.abs(c).max(axis=(0, 1)), rtol=1e-12)
                np.testing.assert_allclose(st[s, :, 3], np.abs(c).min(axis=(0, 1)), rtol=1e-9, atol=1e-15)
                np.testing.assert_allclose(st[s, :, 4], (np.abs(c) ** 2).sum(axis=(0, 1)), rtol=1e-12)
        with pytest.raises(ValueError):
            plan.stats(np.zeros((3,) + shape))                      # more slices than the plan holds
        with pytest.raises(ffi.P3DError):
            plan.run(np.zeros((1,) + shape), np.ones(shape), np.full((1, 2, nsh), 1j), 2)      # complex thresholds on a real cube
    with pytest.raises(NotImplementedError):
        ffi.ShearletPlan64(psi.astype(np.complex128))


def test_shearlet_config4_slice_in_the_reference_precision(so):
    """BASELINE configs[4]'s slice as stated (2048 x 1024, 125 shearlets, 80 % missing, hard threshold, exponential decay), first iterations
    through the double-precision loop against the oracle's real-transform form: 1e-10 where the float32 kernels hold 2e-4."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    kw = dict(thresh_op="hard", thresh_model="exponential", niter=3, p_max=0.99, p_min=0.05)
    shape = (2048, 1024)
    psi = shearlets.scalesShearsAndSpectra(shape)
    mask = po.synthetic_mask(*shape, 0.8)
    x = (po.synthetic_slice(*shape, 40, real=True) * mask).astype(np.float32)
    info, res = {}, []
    want = so.pocs_slice_shearlet_real(x.astype(np.float64), mask, psi, info=info, **kw)
    got = P.pocs_cube(x[None], mask, transform_kind="SHEARLET", auxiliary_data=psi, precision="reference", results=res, eps=0.0, **kw)
    assert got.dtype == np.float32
    err = rel_l2(got[0], want)
    print(f"configs[4] slice in double: device-vs-oracle rel-L2 {err:.3e} (float32 result)")
    assert err <= 2e-7, err
    np.testing.assert_allclose(res[0]["costs"], info["costs"], rtol=1e-6, atol=1e-24)
    got64 = P.pocs_cube(x[None].astype(np.float64), mask, transform_kind="SHEARLET", auxiliary_data=psi, eps=0.0, **kw)
    assert got64.dtype == np.float64 and rel_l2(got64[0], want) <= 1e-10, rel_l2(got64[0], want)
    P.release_plans()


@pytest.mark.parametrize("shape,complex_,kw", [
    ((64, 128), False, dict(niter=6, thresh_op="hard", thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2)),
    ((128, 64), True, dict(niter=5, thresh_op="soft", thresh_model="linear", eps=0.0, p_max=0.9, p_min=0.05, alpha=0.8, version="adaptive")),
    ((512, 256), False, dict(niter=25, thresh_op="garrote", thresh_model="exponential", eps=1e-4, p_max=0.99, p_min=1e-2)),
    ((600, 500), False, dict(niter=4, thresh_op="soft", thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2)),
])
def test_fused_double_precision_shearlet_passes_equal_the_unfused_ones(ffi, shape, complex_, kw, monkeypatch):
    """The three fused passes on the register engine (p3d_mix64.hip: x Psi_s + inverse rows; inverse columns, threshold, forward columns; forward
    rows, x Psi_s, sum over s) against the separate kernels around the line transforms: the same loop to rounding (1e-12), the same iteration
    counts, statistics alike."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    psi = shearlets.scalesShearsAndSpectra(shape)
    mask = po.synthetic_mask(shape[0], shape[1], 0.6)
    cube = np.stack([po.synthetic_slice(shape[0], shape[1], 50 + s, real=not complex_) for s in range(3)]) * mask
    cube = cube.astype(np.complex128 if complex_ else np.float64)
    cube[1] = 0
    P.release_plans()
    with ffi.ShearletPlan64(psi, max_slices=3) as plan:
        assert plan.fused and 0.0 < plan.row_group_fraction <= 1.0
        st_f = plan.stats(cube)
    res_f = []
    fused = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res_f, batch_slices=2, **kw)
    P.release_plans()
    monkeypatch.setenv("P3D_SHEARLET64_UNFUSED", "1")
    with ffi.ShearletPlan64(psi, max_slices=3) as plan:
        assert not plan.fused
        st_u = plan.stats(cube)
    res_u = []
    plain = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res_u, batch_slices=2, **kw)
    P.release_plans()
    monkeypatch.delenv("P3D_SHEARLET64_UNFUSED")
    np.testing.assert_allclose(st_f[[0, 2]][..., [0, 2, 4]], st_u[[0, 2]][..., [0, 2, 4]], rtol=1e-11)
    np.testing.assert_allclose(st_f[[0, 2]][..., 3], st_u[[0, 2]][..., 3], rtol=1e-6, atol=1e-14)   # min |c|: next to a zero crossing
    assert not fused[1].any() and not plain[1].any()
    for s in (0, 2):
        assert res_f[s]["niterations"] == res_u[s]["niterations"]
        assert rel_l2(fused[s], plain[s]) <= 1e-12, (s, rel_l2(fused[s], plain[s]))
        np.testing.assert_allclose(res_f[s]["costs"], res_u[s]["costs"], rtol=1e-7, atol=1e-26)


@pytest.mark.parametrize("n", [96, 100, 125, 147, 189, 225, 243, 343, 375, 500, 625, 720, 875, 1029, 1125, 1250, 1715, 2058, 2401, 2500, 3000, 3645, 4096])
def test_fused_double_precision_shearlet_passes_on_every_kind_of_plan(ffi, n, monkeypatch):
    """The fused passes read and write the two register layouts of a plan (first / last forward pass); lengths with two-, three- and four-pass
    plans, odd lengths, lengths whose passes use different thread counts -- as the row and as the column extent, against the unfused passes."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    kw = dict(niter=2, thresh_op="soft", thresh_model="linear", eps=0.0, p_max=0.8, p_min=0.2)
    for shape in ((n, 64), (64, n)):
        psi = shearlets.scalesShearsAndSpectra(shape)
        mask = po.synthetic_mask(shape[0], shape[1], 0.5)
        cube = (po.synthetic_slice(shape[0], shape[1], 77, real=True) * mask).astype(np.float64)[None]
        P.release_plans()
        with ffi.ShearletPlan64(psi, max_slices=1) as plan:
            assert plan.fused
        fused = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
        P.release_plans()
        monkeypatch.setenv("P3D_SHEARLET64_UNFUSED", "1")
        plain = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
        P.release_plans()
        monkeypatch.delenv("P3D_SHEARLET64_UNFUSED")
        assert np.isfinite(plain).all() and rel_l2(fused[0], plain[0]) <= 1e-12, (shape, rel_l2(fused[0], plain[0]))


def test_skipping_rows_off_a_shearlets_support_changes_nothing_in_double_precision(ffi, monkeypatch):
    """The fused double-precision passes skip the row groups on which a shearlet's spectrum vanishes (p3d_shearlet64_info: row_group_fraction);
    P3D_SHEARLET64_NO_SUPPORT=1 moves every row: the same result to rounding (the skipped rows carry exact zeros), half of the (shearlet, row
    group) pairs or fewer at this size."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    shape = (512, 256)
    kw = dict(niter=5, thresh_op="hard", thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2)
    psi = shearlets.scalesShearsAndSpectra(shape)
    mask = po.synthetic_mask(shape[0], shape[1], 0.6)
    cube = np.stack([po.synthetic_slice(shape[0], shape[1], 60 + s, real=True) for s in range(2)]) * mask
    P.release_plans()
    with ffi.ShearletPlan64(psi, max_slices=1) as plan:
        assert plan.fused and plan.row_group_fraction < 0.6, plan.row_group_fraction
        st_s = plan.stats(cube[:1])
    fast = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
    P.release_plans()
    monkeypatch.setenv("P3D_SHEARLET64_NO_SUPPORT", "1")
    with ffi.ShearletPlan64(psi, max_slices=1) as plan:
        assert plan.fused and plan.row_group_fraction == 1.0
        st_d = plan.stats(cube[:1])
    dense = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
    P.release_plans()
    monkeypatch.delenv("P3D_SHEARLET64_NO_SUPPORT")
    np.testing.assert_allclose(st_s[..., [0, 2, 4]], st_d[..., [0, 2, 4]], rtol=1e-12)
    assert rel_l2(fast, dense) <= 1e-13, rel_l2(fast, dense)


def test_single_precision_cubes_take_the_double_loop_where_it_is_the_fused_one(so, monkeypatch):
    """Extents that are not powers of two but have plans on the double-precision register engine: the float32 loop is unfused there, the double one fused
    -- faster and the reference's arithmetic -- so `pocs_cube` routes float32 / complex64 cubes to it by default (1e-7 of the oracle: the cast back);
    precision='float32' keeps the float32 kernels, powers of two keep them anyway."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    kw = dict(niter=6, thresh_op="soft", thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2)
    calls = []
    real_double = P._pocs_cube_shearlet_double
    monkeypatch.setattr(P, "_pocs_cube_shearlet_double", lambda *a, **k: (calls.append(a[0].shape), real_double(*a, **k))[1])
    for shape, expect in (((150, 240), True), ((128, 64), False), ((48, 40), False)):
        psi = shearlets.scalesShearsAndSpectra(shape)
        mask = po.synthetic_mask(shape[0], shape[1], 0.5)
        cube = (np.stack([po.synthetic_slice(shape[0], shape[1], 30 + s, real=True) for s in range(2)]) * mask).astype(np.float32)
        want = so.pocs_cube_shearlet(cube.astype(np.float64), mask, psi, **kw)
        calls.clear()
        got = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
        assert got.dtype == np.float32 and bool(calls) == expect, (shape, calls)
        assert rel_l2(got, want) <= (2e-7 if expect else 1e-5), (shape, rel_l2(got, want))
        calls.clear()
        fast = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, precision="float32", **kw)
        assert not calls and rel_l2(fast, want) <= 1e-5


@pytest.mark.parametrize("shape", [(64, 128), (600, 500), (1024, 256), (96, 1000)])
@pytest.mark.parametrize("op", ["hard", "soft"])
def test_two_columns_per_transform_in_double_precision(ffi, so, shape, op, monkeypatch):
    """Real cubes on symmetric spectra (what scalesShearsAndSpectra builds) have real coefficients: the fused double-precision passes then work on Hermitian
    coefficient slices -- rows 0 ... nil/2 only, two columns through one complex transform (mix64::shear_col_pair_kernel).  Against the general passes
    (P3D_SHEARLET64_NO_PAIR=1) to rounding, statistics alike, and against the oracle; complex cubes are not affected."""
    import pseudo_3d_interpolation_amd.functions.POCS as P
    from oracle import pocs_oracle as po
    from pseudo_3d_interpolation_amd.functions import shearlets
    kw = dict(niter=5, thresh_op=op, thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2)
    psi = shearlets.scalesShearsAndSpectra(shape)
    mask = po.synthetic_mask(shape[0], shape[1], 0.6)
    cube = (np.stack([po.synthetic_slice(shape[0], shape[1], 80 + s, real=True) for s in range(2)]) * mask).astype(np.float64)
    P.release_plans()
    with ffi.ShearletPlan64(psi, max_slices=2) as plan:
        assert plan.fused and plan.paired
        st_p = plan.stats(cube)
    res_p = []
    paired = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res_p, **kw)
    cplx = P.pocs_cube(cube[:1].astype(np.complex128), mask, transform_kind="SHEARLET", auxiliary_data=psi, **kw)
    P.release_plans()
    monkeypatch.setenv("P3D_SHEARLET64_NO_PAIR", "1")
    with ffi.ShearletPlan64(psi, max_slices=2) as plan:
        assert plan.fused and not plan.paired
        st_g = plan.stats(cube)
    res_g = []
    general = P.pocs_cube(cube, mask, transform_kind="SHEARLET", auxiliary_data=psi, results=res_g, **kw)
    P.release_plans()
    monkeypatch.delenv("P3D_SHEARLET64_NO_PAIR")
    np.testing.assert_allclose(st_p[..., [0, 2, 4]], st_g[..., [0, 2, 4]], rtol=1e-11)
    assert rel_l2(paired, general) <= 1e-11, rel_l2(paired, general)   # (mirrored samples of a generated spectrum differ by up to 2e-13: the frame itself is symmetric to that)
    assert rel_l2(np.real(cplx[0]), general[0]) <= 1e-11           # (the same slice as a complex cube: the general passes by construction)
    for s in range(2):
        np.testing.assert_allclose(res_p[s]["costs"], res_g[s]["costs"], rtol=1e-9, atol=1e-26)
    if max(shape) <= 600:
        want = so.pocs_cube_shearlet(cube, mask, psi, **kw)
        assert rel_l2(paired, want) <= (1e-8 if op == "hard" else 1e-10)
