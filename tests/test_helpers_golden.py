"""The helpers either side of the POCS path (steps 12, 13, 15) against vectors produced by the REFERENCE's own functions
(tests/golden/make_golden_helpers.py: cube_postprocessing_3D.py:88-347, cube_apply_FFT.py:49-181,
cube_POCS_interpolation_3D.py:146-195, functions/utils.py:413-441 -- imported in the build container under the conda
interpreter with inert stand-ins for the two packages the image lacks).  CPU: the product's host logic and the oracle of the
step-15 filters; `-m gpu`: the step-15 filters through the HIP kernels."""
import json
import os

import numpy as np
import pytest

from conftest import rel_l2

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "helpers.json")) as fh:
        meta = json.load(fh)
    return meta, np.load(os.path.join(HERE, "golden", "helpers.npz"))


def _kw(d):
    return {k: (tuple(v) if isinstance(v, list) and k in ("n", "dims") else v) for k, v in d.items()}


def test_rescale_like_the_reference(gold):
    from pseudo_3d_interpolation_amd.functions.utils import rescale
    from oracle import postproc_oracle as po
    meta, z = gold
    for i, case in enumerate(meta["cases"]["rescale"]):
        for fn in (rescale, po.rescale):
            got = fn(z[f"rescale_{i}_in"], **case["kwargs"])
            assert np.array_equal(got, z[f"rescale_{i}_out"], equal_nan=True), (i, fn.__module__)
    assert np.array_equal(rescale(z["rescale_const_in"]), z["rescale_const_out"])


def test_gaussian_kernel_like_the_reference(gold):
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    from oracle import postproc_oracle as po
    meta, z = gold
    for i, case in enumerate(meta["cases"]["gaussian_kernel_2d"]):
        want = z[f"gk2d_{i}"]
        for mod in (pp, po):
            got = mod.gaussian_kernel_2d(**_kw(case["kwargs"]))
            assert got.shape == want.shape and np.allclose(got, want, rtol=1e-14, atol=0), (i, mod.__name__)


def test_kxky_filters_and_their_oracle_like_the_reference(gold):
    """The filter grids built on the host (product: own FFT convolution; oracle: SciPy's) and the oracle's filtered slices."""
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    from oracle import postproc_oracle as po
    meta, z = gold
    for i, case in enumerate(meta["cases"]["remove_acquisition_footprint"]):
        kw = _kw(case["kwargs"])
        data, want, wfilt = z[f"footprint_{i}_in"], z[f"footprint_{i}_out"], z[f"footprint_{i}_filter"]
        fkw = {k: v for k, v in kw.items() if k != "verbose"}
        assert np.abs(pp.footprint_filter(data.shape, **fkw) - wfilt).max() < 1e-12, i
        assert np.abs(po.footprint_filter(data.shape, **fkw) - wfilt).max() < 1e-12, i
        # (the fixture was written under NumPy 1.26, whose fft2 computes a float32 slice in double; NumPy 2 keeps it single: feed double)
        assert rel_l2(po.remove_acquisition_footprint(data.astype(np.float64), **fkw), want) < 1e-12, i
    for i, case in enumerate(meta["cases"]["spatial_antialiasing"]):
        data, want, wfilt = z[f"antialias_{i}_in"], z[f"antialias_{i}_out"], z[f"antialias_{i}_filter"]
        kw = _kw(case["kwargs"])
        assert np.abs(pp.antialias_filter(data.shape, case["direction"], case["factors"], **kw) - wfilt).max() < 1e-12, i
        assert np.abs(po.antialias_filter(data.shape, case["direction"], case["factors"], **kw) - wfilt).max() < 1e-12, i
        assert rel_l2(po.spatial_antialiasing(data.astype(np.float64), case["direction"], case["factors"], **kw), want) < 1e-12, i
    assert meta["antialias_bad_keys"] == "ValueError"
    with pytest.raises(ValueError):
        pp.antialias_filter((8, 8), "iline", {"a": 1, "b": 2})


def test_smoothing_oracle_like_the_reference(gold):
    from oracle import postproc_oracle as po
    meta, z = gold
    for i, case in enumerate(meta["cases"]["smoothing_filter"]):
        got = po.smoothing_filter(z[f"smooth_{i}_in"], case["filter_name"], case["kwargs_filter"], case["rescale_slice"], case["kwargs_rescale"])
        want = z[f"smooth_{i}_out"]
        if case["filter_name"] == "median" and not case["rescale_slice"]:
            assert np.array_equal(got, want), i
        else:   # (SciPy 1.7 wrote the fixture, this interpreter's SciPy evaluates the oracle: same algorithm, float32 output)
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), i


def test_frequency_windows_like_the_reference(gold):
    from pseudo_3d_interpolation_amd import cube_apply_FFT as F
    meta, z = gold
    for case in meta["cases"]["_get_stopband"]:
        want = z[f"stopband_{case['nstopband']}_{case['kind']}"]
        got = F._get_stopband(case["nstopband"], case["kind"])
        assert got.shape == want.shape and np.array_equal(got, want), case
    for kind, vals in meta["const_values"].items():
        assert list(F._get_const_values(kind)) == vals
    for i, case in enumerate(meta["cases"]["get_freq_filter_win"]):
        got = F.get_freq_filter_win(list(case["filter_freqs"]), z[f"fwin_{i}_freqs"], filter_type=case["filter_type"])
        want = z[f"fwin_{i}_win"]
        assert got.shape == want.shape and np.array_equal(got, want), case
    for i, case in enumerate(meta["cases"]["get_freq_filter_mask"]):
        got = F.get_freq_filter_mask(z["fmask_freqs"], freqs=list(case["freqs"]), filter_type=case["filter_type"])
        assert got.dtype == bool and np.array_equal(got, z[f"fmask_{i}"]), case


def test_batch_file_names_and_runtime_files_like_the_reference(gold, tmp_path):
    from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as D
    meta, _ = gold
    for case in meta["cases"]["create_file_path"]:
        kw = {k: v for k, v in case["kwargs"].items() if k != "dim"}   # (the product takes the coordinate itself, not a dataset + dim)
        assert D.create_file_path(np.asarray(case["coord"], dtype=np.float64), **kw) == case["path"], case
    rec = meta["combine_runtime_results"]
    for name, text in rec["inputs"].items():
        (tmp_path / name).write_text(text)
    D.combine_runtime_results(str(tmp_path), prefix=rec["prefix"], fsuffix=rec["fsuffix"])
    created = sorted(f for f in os.listdir(tmp_path) if f not in rec["inputs"])
    assert created == rec["created"]
    assert sorted((tmp_path / created[0]).read_text().splitlines()) == rec["content_lines_sorted"]


# ---- the step-15 filters through the HIP kernels, against the reference's own outputs -------------------------------------------
@pytest.mark.gpu
def test_kxky_filters_on_the_gpu_like_the_reference(gold):
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    meta, z = gold
    for i, case in enumerate(meta["cases"]["remove_acquisition_footprint"]):
        kw = _kw(case["kwargs"])
        got, filt = pp.remove_acquisition_footprint(z[f"footprint_{i}_in"], return_filter=True, verbose=0, **kw)
        assert np.abs(filt - z[f"footprint_{i}_filter"]).max() < 1e-12
        assert rel_l2(got, z[f"footprint_{i}_out"]) < 2e-6, (i, rel_l2(got, z[f"footprint_{i}_out"]))
    for i, case in enumerate(meta["cases"]["spatial_antialiasing"]):
        got = pp.spatial_antialiasing(z[f"antialias_{i}_in"], case["direction"], case["factors"], verbose=0, **_kw(case["kwargs"]))
        assert rel_l2(got, z[f"antialias_{i}_out"]) < 2e-6, i


@pytest.mark.gpu
def test_smoothing_filters_on_the_gpu_like_the_reference(gold):
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    meta, z = gold
    for i, case in enumerate(meta["cases"]["smoothing_filter"]):
        got = pp.smoothing_filter(z[f"smooth_{i}_in"], case["filter_name"], case["kwargs_filter"], case["rescale_slice"], case["kwargs_rescale"])
        want = z[f"smooth_{i}_out"]
        assert got.shape == want.shape
        if case["filter_name"] == "median" and not case["rescale_slice"]:
            assert np.array_equal(got, want), i
        else:
            assert np.abs(got - want).max() <= 4e-6 * max(1.0, np.abs(want).max()), (i, np.abs(got - want).max())
