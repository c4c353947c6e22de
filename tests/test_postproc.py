"""Step-15 kx-ky filters (partial mirror of cube_postprocessing_3D.py): host-side filter construction against the SciPy-based
oracle on CPU, the GPU application against the oracle on a GPU box."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import postproc_oracle as po


def test_filters_match_the_scipy_restatement():
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    for sigma, orient in ((7, 'equal'), (3, 'iline'), (5, 'xline')):
        assert np.allclose(pp.gaussian_kernel_2d(sigma, orientation=orient), po.gaussian_kernel_2d(sigma, orientation=orient), rtol=1e-12, atol=0)
    for shape in ((96, 128), (75, 61), (200, 40)):
        for direction in ('both', 'iline', 'xline', 'twt'):
            a = pp.footprint_filter(shape, sigma=4, direction=direction, buffer_center=0.3, buffer_filter=2)
            b = po.footprint_filter(shape, sigma=4, direction=direction, buffer_center=0.3, buffer_filter=2)
            assert a.shape == shape and np.abs(a - b).max() < 1e-10, (shape, direction)
        for direction, fac in (('iline', {'iline': 4, 'xline': 1}), ('xline', {'iline': 1, 'xline': 2})):
            a = pp.antialias_filter(shape, direction, fac, sigma=3)
            b = po.antialias_filter(shape, direction, fac, sigma=3)
            assert np.abs(a - b).max() < 1e-10, (shape, direction)
    with pytest.raises(ValueError):
        pp.antialias_filter((32, 32), 'iline', {'il': 2, 'xl': 1})


@pytest.mark.gpu
def test_kxky_filters_on_the_gpu():
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    rng = np.random.default_rng(2)
    for shape in ((96, 128), (75, 61)):
        stack = rng.standard_normal((5,) + shape).astype(np.float32)
        got, filt = pp.remove_acquisition_footprint(stack, sigma=4, direction='both', buffer_filter=2, return_filter=True)
        assert got.shape == stack.shape and got.dtype == np.float32 and filt.shape == shape
        for s in range(5):
            want = po.remove_acquisition_footprint(stack[s].astype(np.float64), sigma=4, direction='both', buffer_filter=2)
            assert rel_l2(got[s], want) < 2e-6
        one = pp.remove_acquisition_footprint(stack[0], sigma=4, direction='both', buffer_filter=2)
        assert one.shape == shape and np.array_equal(one, got[0])
        fac = {'iline': 4, 'xline': 1}
        got = pp.spatial_antialiasing(stack[0], 'iline', fac, sigma=3)
        assert rel_l2(got, po.spatial_antialiasing(stack[0].astype(np.float64), 'iline', fac, sigma=3)) < 2e-6


def test_smoothing_filter_rejects_what_is_not_implemented():
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    x = np.zeros((8, 8), np.float32)
    with pytest.raises(TypeError):
        pp.smoothing_filter(x, None, {})
    with pytest.raises(NotImplementedError):
        pp.smoothing_filter(x, 'median', {'size': 4})
    with pytest.raises(NotImplementedError):
        pp.smoothing_filter(x, 'gaussian', {'sigma': (1, 2)})
    with pytest.raises(NotImplementedError):
        pp.smoothing_filter(x, 'gaussian', {'sigma': 1, 'mode': 'nearest'})


@pytest.mark.gpu
def test_smoothing_filters_on_the_gpu():
    from pseudo_3d_interpolation_amd import cube_postprocessing_3D as pp
    rng = np.random.default_rng(3)
    for shape in ((96, 128), (75, 61), (5, 9)):               # the last one is smaller than the kernel: multiple reflections
        stack = rng.standard_normal((4,) + shape).astype(np.float32)
        for kw in ({'sigma': 1}, {'sigma': 2.5}, {'sigma': 3, 'truncate': 2.0}):
            got = pp.smoothing_filter(stack, 'gaussian', kw)
            assert got.shape == stack.shape and got.dtype == np.float32
            for s in range(4):
                want = po.smoothing_filter(stack[s].astype(np.float64), 'gaussian', kw)
                assert np.abs(got[s] - want).max() < 2e-6, (shape, kw)
        for size in (3, 5, 7):
            got = pp.smoothing_filter(stack, 'median', {'size': size})
            for s in range(4):
                assert np.array_equal(got[s], po.smoothing_filter(stack[s], 'median', {'size': size})), (shape, size)
        one = pp.smoothing_filter(stack[1], 'gaussian', {'sigma': 2}, rescale_slice=True, kwargs_rescale={'vminmax': (99, 1)})
        want = po.smoothing_filter(stack[1].astype(np.float64), 'gaussian', {'sigma': 2}, True, {'vminmax': (99, 1)})
        assert one.shape == shape and np.abs(one - want).max() < 1e-5 * np.abs(want).max()
