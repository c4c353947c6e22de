"""
Shearlet spectra for ``transform_kind='SHEARLET'`` -- host-side stand-in for ``FFST.scalesShearsAndSpectra`` of PyShearlets, which
the reference calls once per run (cube_POCS_interpolation_3D.py:269-272) and which is not installed in this image.

The spectra follow S. Haeuser & G. Steidl, "Fast Finite Shearlet Transform: a tutorial" (2014): Meyer scaling function, Meyer
wavelet along the cone axis, bump function across it, shears k = -2^j .. 2^j per scale j with the two seam elements |k| = 2^j
glued from both cones; the squares of all spectra sum to one at every frequency (Parseval frame), which is what makes
``inverse(transform(x)) == x``.  The sample-exact grid conventions of PyShearlets could not be compared (no copy of it here):
this generator and the GPU transform are self-consistent and are tested as such; DESIGN.md lists the row as "parity unpinned".

Only the one-off set-up lives here (NumPy, separable evaluation); the transform and the POCS loop run on the GPU
(``_ffi.ShearletPlan``).
"""
import os

import numpy as np


def get_number_scales(shape):
    """Number of scales for an array shape (POCS.py:21-31)."""
    scales = int(np.floor(0.5 * np.log2(np.max(shape))))
    return scales if scales >= 1 else 1


def _v(x):
    x = np.clip(x, 0.0, 1.0)
    return x ** 4 * (35.0 - 84.0 * x + 70.0 * x ** 2 - 20.0 * x ** 3)


def _ramp(w):
    """sin(pi/2 v(2|w|-1)) on [1/2, 1), cos(pi/2 v(|w|-1)) on [1, 2), else 0."""
    w = np.abs(w)
    out = np.zeros_like(w)
    up = (w >= 0.5) & (w < 1.0)
    down = (w >= 1.0) & (w < 2.0)
    out[up] = np.sin(0.5 * np.pi * _v(2.0 * w[up] - 1.0))
    out[down] = np.cos(0.5 * np.pi * _v(w[down] - 1.0))
    return out


def _wavelet(w):
    return np.hypot(_ramp(w), _ramp(0.5 * w))


def _bump(t):
    return np.sqrt(_v(1.0 - np.abs(t)))


def _scaling(w):
    w = np.abs(w)
    out = np.zeros_like(w)
    out[w < 0.5] = 1.0
    mid = (w >= 0.5) & (w < 1.0)
    out[mid] = np.cos(0.5 * np.pi * _v(2.0 * w[mid] - 1.0))
    return out


def scalesShearsAndSpectra(shape, numOfScales=None, realCoefficients=True, fftshift_spectra=True, dtype=np.float64):
    """
    Spectra of all shearlets for slices of ``shape`` = (nil, nxl): array ``(nil, nxl, 1 + sum_j 2**(j+2))``, low-pass element
    first, then scale by scale, shear by shear (cone around the xline-frequency axis before the one around the iline axis).
    With ``fftshift_spectra=True`` (what the reference passes) the zero frequency sits at index [0, 0].
    """
    if not realCoefficients:
        raise NotImplementedError('complex shearlets (realCoefficients=False) are not implemented')
    nil, nxl = int(shape[0]), int(shape[1])
    J = get_number_scales((nil, nxl)) if numOfScales is None else int(numOfScales)
    X = 2.0 ** (2 * (J - 1) + 1)
    # frequency axes of the next odd grid, cropped to the slice shape: axis 0 runs from +X downwards, axis 1 from -X upwards
    wy = np.linspace(-X, X, nil + (nil % 2 == 0))[::-1][:nil][:, None]
    wx = np.linspace(-X, X, nxl + (nxl % 2 == 0))[:nxl][None, :]
    cone_x = np.abs(wx) >= np.abs(wy)
    counts = [2 ** (j + 2) for j in range(J)]
    nsh = 1 + sum(counts)
    # built plane by plane in the layout the GPU plan uploads, (nsh, nil, nxl) -- a plane is one contiguous write -- and handed out
    # as the (nil, nxl, nsh) VIEW of it that FFST's layout asks for (same values; `_ffi.ShearletPlan` moves the axis back without
    # a copy).  Planes are independent: a few threads share them (NumPy's loops release the GIL).
    stack = np.empty((nsh, nil, nxl), dtype=dtype)
    stack[0] = np.where(cone_x, _scaling(wx), _scaling(wy))
    with np.errstate(divide='ignore', invalid='ignore'):
        slope_x = np.where(wx != 0, wy / np.where(wx != 0, wx, 1.0), wy * 4.0 ** J)   # wy / wx; the bump vanishes where wx = 0
        slope_y = np.where(wy != 0, wx / np.where(wy != 0, wy, 1.0), wx * 4.0 ** J)
    jobs = []
    n = 1
    for j in range(J):
        a = 4.0 ** (-j)
        radial = (_wavelet(a * wx), _wavelet(a * wy))
        for k in range(-2 ** j, 2 ** j + 1):
            jobs.append((j, k, n, radial))
            n += 1 if abs(k) == 2 ** j else 2

    def plane(job):
        j, k, n, (radial_x, radial_y) = job
        along_x = radial_x * _bump(2.0 ** j * slope_x + k)
        along_y = radial_y * _bump(2.0 ** j * slope_y + k)
        if abs(k) == 2 ** j:
            stack[n] = np.where(cone_x, along_x, along_y)
        else:
            stack[n] = along_x
            stack[n + 1] = along_y

    workers = max(1, min(8, (os.cpu_count() or 1), len(jobs))) if nil * nxl >= (1 << 16) else 1
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(plane, jobs))
    else:
        for job in jobs:
            plane(job)
    # an even extent keeps the Nyquist line (index 0 of the centred grid) without its mirror image: give it the root mean square
    # of both so that the spectra stay symmetric (real shearlets) and their squares still sum to one
    fine = slice(1 + sum(counts[:-1]), None)
    if nil % 2 == 0:
        c0 = 1 - nxl % 2
        line = stack[fine, 0, c0:].copy()
        stack[fine, 0, c0:] = np.sqrt(0.5 * (line ** 2 + line[:, ::-1] ** 2))
    if nxl % 2 == 0:
        r0 = 1 - nil % 2
        line = stack[fine, r0:, 0].copy()
        stack[fine, r0:, 0] = np.sqrt(0.5 * (line ** 2 + line[:, ::-1] ** 2))
    if fftshift_spectra:
        for i in range(nsh):   # plane by plane: np.fft.ifftshift of the whole stack would hold a second copy of it
            stack[i] = np.fft.ifftshift(stack[i])
    return np.moveaxis(stack, 0, -1)
