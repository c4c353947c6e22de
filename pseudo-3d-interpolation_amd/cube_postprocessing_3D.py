"""
Step 15 -- the kx-ky slice filters of ``pseudo_3D_interpolation/cube_postprocessing_3D.py`` on the GPU (partial mirror).

Covered: ``remove_acquisition_footprint`` (:179-260) and ``spatial_antialiasing`` (:263-347) -- both are
``ifft2(ifftshift(filter) * fft2(slice)).real`` with a filter that depends on the slice shape only -- plus their helper
``gaussian_kernel_2d`` (:127-176).  The filter is built once on the host (NumPy; the reference uses scipy.signal.fftconvolve for the
same convolution), the slices go through the 2-D FFT kernels of this package in batches.  ``smoothing_filter`` (:88-124) runs
scipy.ndimage's gaussian / median filter semantics in HIP gather kernels.  Not covered: AGC, upsampling and the xarray/netCDF
driver around them.
"""
import numpy as np

from . import _ffi
from .functions.utils import rescale


def smoothing_filter(x, filter_name=None, kwargs_filter=None, rescale_slice=False, kwargs_rescale=None, device=0):
    """Same arguments as the reference's wrapper of ``scipy.ndimage`` (cube_postprocessing_3D.py:88-124); ``x`` is one slice
    ``(ny, nx)`` or a stack ``(n, ny, nx)`` (each slice filtered -- and rescaled -- on its own).  Supported: ``gaussian`` with a
    scalar ``sigma`` (``truncate`` optional) and ``median`` with an odd ``size`` of 3, 5 or 7, both with SciPy's default
    'reflect' boundary; the arithmetic is float32."""
    kwargs_filter = dict(kwargs_filter or {})
    if filter_name not in ('gaussian', 'median'):
        raise TypeError(f'unknown filter {filter_name!r}')       # the reference calls None(...) here
    if kwargs_filter.pop('mode', 'reflect') != 'reflect':
        raise NotImplementedError("only the default boundary mode 'reflect' is implemented")
    if filter_name == 'gaussian':
        extra = set(kwargs_filter) - {'sigma', 'truncate'}
        if extra or np.ndim(kwargs_filter.get('sigma')) != 0 or kwargs_filter.get('sigma') is None:
            raise NotImplementedError(f'gaussian: scalar sigma (and truncate) only, got {sorted(kwargs_filter)}')
    else:
        extra = set(kwargs_filter) - {'size'}
        if extra or kwargs_filter.get('size') not in (3, 5, 7):
            raise NotImplementedError(f'median: size 3, 5 or 7 only, got {kwargs_filter}')
    x = np.asarray(x)
    squeeze = x.ndim == 2
    stack = x[None] if squeeze else x
    filt = _ffi.smooth_slices(stack, filter_name, device=device, **kwargs_filter)
    if rescale_slice:
        lo_hi = sorted(kwargs_rescale['vminmax'])
        for i in range(stack.shape[0]):
            vmin, vmax = np.percentile(stack[i], lo_hi)
            filt[i] = rescale(filt[i], vmin=vmin, vmax=vmax)
    return filt[0] if squeeze else filt


def _gaussian_window(m, sigma):
    k = np.arange(m) - (m - 1) / 2.0
    return np.exp(-0.5 * (k / sigma) ** 2)


def gaussian_kernel_2d(sigma=7, n=None, normalized=True, orientation='equal'):
    """2-D Gaussian kernel (same arguments and sizes as the reference's, cube_postprocessing_3D.py:127-176)."""
    ny, nx = n if isinstance(n, tuple) else (n, n)
    factor = {'equal': (8, 8), 'iline': (2, 8), 'xline': (8, 2)}
    ny = sigma * factor[orientation][0] + 1 if ny is None else ny
    ny = ny + 1 if ny % 2 == 0 else ny
    nx = sigma * factor[orientation][1] + 1 if nx is None else nx
    nx = nx + 1 if nx % 2 == 0 else nx
    kernel = np.outer(_gaussian_window(ny, sigma), _gaussian_window(nx, sigma))
    if normalized:
        kernel /= 2 * np.pi * (sigma ** 2)
    return kernel


def _convolve_same(a, k):
    """Linear convolution cropped to the shape of ``a`` around its centre (scipy.signal.fftconvolve(a, k, mode='same'))."""
    full = (a.shape[0] + k.shape[0] - 1, a.shape[1] + k.shape[1] - 1)
    out = np.fft.irfft2(np.fft.rfft2(a, full) * np.fft.rfft2(k, full), full)
    r0, c0 = (k.shape[0] - 1) // 2, (k.shape[1] - 1) // 2
    return out[r0:r0 + a.shape[0], c0:c0 + a.shape[1]]


def _orient(direction, dims, ny, nx):
    if direction == 'iline':
        return 'horizontal' if dims[0] == 'iline' else 'vertical'
    if direction == 'xline':
        return 'vertical' if dims[1] == 'xline' else 'horizontal'
    if direction == 'twt':
        return 'vertical' if ny > nx else 'horizontal'
    return direction


def footprint_filter(shape, sigma=7, direction='both', buffer_center=0.25, buffer_filter=3, dims=('iline', 'xline')):
    """Centred kx-ky weight (1 = keep) that notches the acquisition footprint (cube_postprocessing_3D.py:212-253)."""
    ny, nx = shape
    npad = sigma * 5
    ny_pad, nx_pad = ny + npad, nx + npad
    grid = np.zeros((ny_pad, nx_pad))
    direction = _orient(direction, dims, ny, nx)
    if direction in ('both', 'horizontal'):
        cidx = nx_pad // 2 + 1
        fwidth = round(ny_pad * (1 - buffer_center) + .5) // 2
        grid[:fwidth, cidx - buffer_filter: cidx + buffer_filter + 1] = 1
        grid[-fwidth:, cidx - buffer_filter: cidx + buffer_filter + 1] = 1
    if direction in ('both', 'vertical'):
        cidx = ny_pad // 2 + 1
        fwidth = round(nx_pad * (1 - buffer_center) + .5) // 2
        grid[cidx - buffer_filter: cidx + buffer_filter + 1, :fwidth] = 1
        grid[cidx - buffer_filter: cidx + buffer_filter + 1, -fwidth:] = 1
    ffilter = _convolve_same(grid, gaussian_kernel_2d(sigma=sigma))
    return 1 - rescale(ffilter[npad // 2: -npad // 2, npad // 2: -npad // 2])


def antialias_filter(shape, direction, factors_upsampling, sigma=7, dims=('iline', 'xline')):
    """Centred kx-ky weight of the de-aliasing filter after iline / xline upsampling (cube_postprocessing_3D.py:300-340)."""
    il, xl = dims
    if not sorted(dims) == sorted(factors_upsampling.keys()):
        raise ValueError(f'Coordinates {dims} not found in `factors_upsampling` {factors_upsampling.keys()}')
    ny, nx = shape
    npad = sigma * 5
    p = 0.98
    grid = np.zeros((ny + npad, nx + npad))
    direction = _orient(direction, dims, ny, nx)
    if direction == 'horizontal':
        perc = 1 - factors_upsampling.get(xl, 1) / factors_upsampling.get(il, 1)
        half = round(ny * perc * p) // 2 + npad
        grid[half:-half, :] = 1
    elif direction == 'vertical':
        perc = 1 - factors_upsampling.get(il, 1) / factors_upsampling.get(xl, 1)
        half = round(nx * perc * p) // 2 + npad
        grid[:, half:-half] = 1
    ffilter = _convolve_same(grid, gaussian_kernel_2d(sigma=sigma))
    return rescale(ffilter[npad // 2: -npad // 2, npad // 2: -npad // 2], vmin=1e-3, vmax=1)


def apply_kxky_filter(data, ffilter, device=0, batch_slices=None):
    """``ifft2(ifftshift(ffilter) * fft2(slice)).real`` for one slice ``(ny, nx)`` or a stack ``(n, ny, nx)`` on the GPU
    (one real spectrum weight = a one-element frame of ``_ffi.ShearletPlan``)."""
    data = np.asarray(data)
    squeeze = data.ndim == 2
    stack = data[None] if squeeze else data
    if stack.ndim != 3 or stack.shape[1:] != ffilter.shape:
        raise ValueError(f'data {data.shape} does not match the filter {ffilter.shape}')
    n = stack.shape[0]
    step = int(batch_slices) if batch_slices else max(1, min(n, (256 << 20) // (stack.shape[1] * stack.shape[2] * 8)))
    out = np.empty(stack.shape, np.float64 if stack.dtype == np.float64 else np.float32)
    weight = np.fft.ifftshift(np.asarray(ffilter, dtype=np.float64))[..., None]
    with _ffi.ShearletPlan(weight, max_slices=min(step, n), device=device) as plan:
        for lo in range(0, n, step):
            out[lo:lo + step] = plan.transform(stack[lo:lo + step].astype(np.complex64))[..., 0].real
    return out[0] if squeeze else out


def remove_acquisition_footprint(data, sigma=7, direction='both', buffer_center=0.25, buffer_filter=3, return_filter=False,
                                 dims=('iline', 'xline'), verbose=1, device=0):
    """Same arguments as the reference's function; ``data`` may also be a stack of slices ``(n, ny, nx)``."""
    data = np.asarray(data)
    ffilter = footprint_filter(data.shape[-2:], sigma, direction, buffer_center, buffer_filter, dims)
    filt = apply_kxky_filter(data, ffilter, device=device)
    return (filt, ffilter) if return_filter else filt


def spatial_antialiasing(data, direction, factors_upsampling, sigma=7, dims=('iline', 'xline'), return_filter=False, verbose=1,
                         device=0):
    """Same arguments as the reference's function; ``data`` may also be a stack of slices ``(n, ny, nx)``."""
    data = np.asarray(data)
    ffilter = antialias_filter(data.shape[-2:], direction, factors_upsampling, sigma, dims)
    filt = apply_kxky_filter(data, ffilter, device=device)
    return (filt, ffilter) if return_filter else filt
