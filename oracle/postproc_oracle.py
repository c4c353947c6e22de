"""
CPU oracle for the kx-ky slice filters of step 15 -- TEST INFRASTRUCTURE ONLY.

Restates remove_acquisition_footprint (cube_postprocessing_3D.py:179-260), spatial_antialiasing (:263-347) and
gaussian_kernel_2d (:127-176) with the same SciPy calls the reference makes (scipy.signal.fftconvolve, scipy.signal.windows.gaussian)
and NumPy's FFT.  **Pinned** (round 4): tests/golden/make_golden_helpers.py imported the reference's module in the build container (conda interpreter,
inert stand-ins for the absent xarray / xrft that raise on any use) and recorded its functions' outputs on seeded arrays;
tests/test_helpers_golden.py holds every function below to them.
"""
import numpy as np
from scipy import ndimage, signal


def rescale(a, vmin=0, vmax=1):
    """functions/utils.py:413-441."""
    a = np.asarray(a)
    amin, amax = np.nanmin(a), np.nanmax(a)
    vmin = amin if vmin is None else vmin   # (None: keep the array's own extremum, utils.py:435-436)
    vmax = amax if vmax is None else vmax
    if amin == amax:
        return a
    return vmin + (a - amin) * ((vmax - vmin) / (amax - amin))


def gaussian_kernel_2d(sigma=7, n=None, normalized=True, orientation="equal"):
    ny, nx = n if isinstance(n, tuple) else (n, n)
    factor = {"equal": (8, 8), "iline": (2, 8), "xline": (8, 2)}
    ny = sigma * factor[orientation][0] + 1 if ny is None else ny
    ny = ny + 1 if ny % 2 == 0 else ny
    nx = sigma * factor[orientation][1] + 1 if nx is None else nx
    nx = nx + 1 if nx % 2 == 0 else nx
    kernel = np.outer(signal.windows.gaussian(ny, sigma), signal.windows.gaussian(nx, sigma))
    if normalized:
        kernel /= 2 * np.pi * (sigma ** 2)
    return kernel


def _orient(direction, dims, ny, nx):
    if direction == "iline":
        return "horizontal" if dims[0] == "iline" else "vertical"
    if direction == "xline":
        return "vertical" if dims[1] == "xline" else "horizontal"
    if direction == "twt":
        return "vertical" if ny > nx else "horizontal"
    return direction


def footprint_filter(shape, sigma=7, direction="both", buffer_center=0.25, buffer_filter=3, dims=("iline", "xline")):
    ny, nx = shape
    npad = sigma * 5
    ny_pad, nx_pad = ny + npad, nx + npad
    kernel = gaussian_kernel_2d(sigma=sigma)
    grid = np.zeros((ny_pad, nx_pad), dtype="int8")
    direction = _orient(direction, dims, ny, nx)
    if direction in ("both", "horizontal"):
        cidx = nx_pad // 2 + 1
        fwidth = round(ny_pad * (1 - buffer_center) + 0.5) // 2
        grid[:fwidth, cidx - buffer_filter: cidx + buffer_filter + 1] = 1
        grid[-fwidth:, cidx - buffer_filter: cidx + buffer_filter + 1] = 1
    if direction in ("both", "vertical"):
        cidx = ny_pad // 2 + 1
        fwidth = round(nx_pad * (1 - buffer_center) + 0.5) // 2
        grid[cidx - buffer_filter: cidx + buffer_filter + 1, :fwidth] = 1
        grid[cidx - buffer_filter: cidx + buffer_filter + 1, -fwidth:] = 1
    ffilter = signal.fftconvolve(grid, kernel, mode="same")
    return 1 - rescale(ffilter[npad // 2: -npad // 2, npad // 2: -npad // 2])


def antialias_filter(shape, direction, factors_upsampling, sigma=7, dims=("iline", "xline")):
    il, xl = dims
    if not sorted(dims) == sorted(factors_upsampling.keys()):
        raise ValueError(f"Coordinates {dims} not found in `factors_upsampling` {factors_upsampling.keys()}")
    ny, nx = shape
    npad = sigma * 5
    p = 0.98
    kernel = gaussian_kernel_2d(sigma=sigma)
    grid = np.zeros((ny + npad, nx + npad), dtype="int8")
    direction = _orient(direction, dims, ny, nx)
    if direction == "horizontal":
        perc = 1 - factors_upsampling.get(xl, 1) / factors_upsampling.get(il, 1)
        half = round(ny * perc * p) // 2 + npad
        grid[half:-half, :] = 1
    elif direction == "vertical":
        perc = 1 - factors_upsampling.get(il, 1) / factors_upsampling.get(xl, 1)
        half = round(nx * perc * p) // 2 + npad
        grid[:, half:-half] = 1
    ffilter = signal.fftconvolve(grid, kernel, mode="same")
    return rescale(ffilter[npad // 2: -npad // 2, npad // 2: -npad // 2], vmin=1e-3, vmax=1)


def apply_filter(data, ffilter):
    """np.fft.ifft2(ifftshift(filter) * fft2(slice)).real  (cube_postprocessing_3D.py:255, 342)."""
    return np.fft.ifft2(np.fft.ifftshift(ffilter) * np.fft.fft2(data)).real


def remove_acquisition_footprint(data, **kw):
    return apply_filter(data, footprint_filter(data.shape, **kw))


def spatial_antialiasing(data, direction, factors_upsampling, **kw):
    return apply_filter(data, antialias_filter(data.shape, direction, factors_upsampling, **kw))


def smoothing_filter(x, filter_name=None, kwargs_filter=None, rescale_slice=False, kwargs_rescale=None):
    """cube_postprocessing_3D.py:88-124 -- the same scipy.ndimage calls."""
    func = {"gaussian": ndimage.gaussian_filter, "median": ndimage.median_filter}.get(filter_name)
    if rescale_slice:
        vmin, vmax = np.percentile(x, sorted(kwargs_rescale["vminmax"]))
        return rescale(func(x, **kwargs_filter), vmin=vmin, vmax=vmax)
    return func(x, **kwargs_filter)
