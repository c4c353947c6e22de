"""
CPU oracle for the SHEARLET variant of the POCS path -- TEST INFRASTRUCTURE ONLY (see oracle/pocs_oracle.py).

Two parts with different standing:

1. The SHEARLET branches of the reference's own code (schedule POCS.py:256-259, 282-285, 302-320, 340-341; loop POCS.py:526-527,
   589-590, 610-611; per-shearlet thresholds by NumPy broadcasting in threshold_operator.py).  These are **pinned**:
   tests/golden/make_golden_shearlet.py ran the reference's POCS_algorithm / get_threshold_decay with the transform pair below
   injected as `transform` / `itransform` (the reference takes any callables) and recorded inputs and outputs.

2. The transform itself.  The reference calls FFST.shearletTransformSpect / inverseShearletTransformSpect /
   scalesShearsAndSpectra of the third-party package PyShearlets (cube_POCS_interpolation_3D.py:269-274), which is NOT in the
   reference tree and not installed anywhere in this image (no version is pinned by the reference either).  **Parity of this part
   is unpinned**: what follows restates the published algorithm (S. Haeuser, G. Steidl, "Fast Finite Shearlet Transform: a
   tutorial", 2014): Meyer-type scaling function / wavelet / bump, cone-adapted discrete shearlets with the seam elements
   |k| = 2^j glued from both cones, spectra sampled on a [-X, X]^2 grid with X = 2^(2(J-1)+1), ST_s = ifft2(Psi_s * fft2(x)),
   x = ifft2(sum_s fft2(ST_s) * Psi_s) (Parseval frame: sum_s Psi_s^2 = 1).  Self-consistency (frame identity, perfect
   reconstruction, real input -> real coefficients) is tested; agreement with PyShearlets' sample-exact grid conventions is not.
   Choices made where the paper leaves room: number of scales J = floor(log2(max(shape)) / 2) (what the reference assumes,
   POCS.py:21-31); shearlets ordered low-pass first, then per scale by k = -2^j .. 2^j with the horizontal-cone element before
   the vertical-cone one; spectra stored in FFT order (what `fftshift_spectra=True` produces); even extents are sampled on the
   next odd grid and cropped, and the unpaired Nyquist row / column of the finest scale is symmetrised so that real slices
   keep real coefficients; complex slices are transformed as they are (real and imaginary parts independently).
"""
import time

import numpy as np

from . import pocs_oracle as base


# ---- Meyer building blocks (Haeuser & Steidl, section 2) --------------------------------------------------------------
def meyer_aux(x):
    """v(x) = 35x^4 - 84x^5 + 70x^6 - 20x^7 on [0,1], 0 below, 1 above; v(x) + v(1-x) = 1."""
    x = np.asarray(x, dtype=np.float64)
    p = x ** 4 * (35.0 + x * (-84.0 + x * (70.0 - 20.0 * x)))
    return np.where(x < 0, 0.0, np.where(x > 1, 1.0, p))


def _meyer_h(x):
    """b(2w) of the paper: sin(pi/2 v(2|w|-1)) on [1/2,1), cos(pi/2 v(|w|-1)) on [1,2), 0 elsewhere."""
    xa = np.abs(x)
    rise = (xa >= 0.5) & (xa < 1.0)
    fall = (xa >= 1.0) & (xa < 2.0)
    return rise * np.sin(0.5 * np.pi * meyer_aux(2.0 * xa - 1.0)) + fall * np.cos(0.5 * np.pi * meyer_aux(xa - 1.0))


def meyer_wavelet(x):
    """psi_1^(w) = sqrt(b^2(2w) + b^2(w)), support 1/2 <= |w| <= 4."""
    return np.sqrt(_meyer_h(x) ** 2 + _meyer_h(0.5 * np.asarray(x)) ** 2)


def meyer_bump(x):
    """psi_2^(w) = sqrt(v(1+w)) for w <= 0, sqrt(v(1-w)) for w > 0."""
    x = np.asarray(x, dtype=np.float64)
    return np.sqrt(np.where(x <= 0, meyer_aux(1.0 + x), meyer_aux(1.0 - x)))


def meyer_scaling(x):
    """phi^(w) = 1 for |w| < 1/2, cos(pi/2 v(2|w|-1)) for 1/2 <= |w| < 1, 0 elsewhere."""
    xa = np.abs(x)
    return (xa < 0.5) * 1.0 + ((xa >= 0.5) & (xa < 1.0)) * np.cos(0.5 * np.pi * meyer_aux(2.0 * xa - 1.0))


def _shearlet_spect(x, y, a, s):
    """psi^_{a,s}(x, y) = psi_1^(a x) psi_2^(a^(-1/2) (y/x + s)) on the cone around the x axis."""
    yy = s * np.sqrt(a) * x + np.sqrt(a) * y
    xx = a * x
    safe = np.where(xx == 0, 1.0, xx)
    return meyer_wavelet(xx) * meyer_bump(yy / safe)


def number_of_scales(shape):
    """POCS.py:21-31."""
    return max(int(np.floor(0.5 * np.log2(np.max(shape)))), 1)


def scales_shears_and_spectra(shape, num_scales=None, contiguous=True):
    """Psi (nil, nxl, nsh), real, FFT order; nsh = 1 + sum_j 2^(j+2).  `contiguous=False` hands out the (nil, nxl, nsh) VIEW of the
    (nsh, nil, nxl) stack the planes are built in (same values, no 2-GiB transposing copy for configs[4]'s frame; reductions over
    such a view add up in another order, so the pinned bit-for-bit schedules use the default).

    Evaluated plane by plane into an (nsh, nil, nxl) stack (one contiguous write per shearlet, a few threads share the planes) and
    returned as the (nil, nxl, nsh) view of it: the arithmetic per sample is the one written out in `_shearlet_spect`; only the
    order of evaluation was arranged so that configs[4]'s 2048 x 1024 x 125 frame takes seconds instead of minutes."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    nil, nxl = int(shape[0]), int(shape[1])
    J = number_of_scales(shape) if num_scales is None else int(num_scales)
    odd = (nil + (nil % 2 == 0), nxl + (nxl % 2 == 0))
    X = 2.0 ** (2 * (J - 1) + 1)
    gx = np.linspace(-X, X, odd[1])
    gy = np.linspace(-X, X, odd[0])[::-1]
    xi_x, xi_y = np.meshgrid(gx, gy, indexing="xy")
    hor = np.abs(xi_x) >= np.abs(xi_y)
    ver = ~hor
    per_scale = [2 ** (j + 2) for j in range(J)]
    nsh = 1 + sum(per_scale)
    psi = np.zeros((nsh, nil, nxl))

    def put(pos, plane):            # crop the odd grid to the slice shape
        psi[pos] = plane[:nil, :nxl]

    put(0, meyer_scaling(xi_x) * hor + meyer_scaling(xi_y) * ver)
    jobs, pos = [], 1
    for j in range(J):
        for k in range(-2 ** j, 2 ** j + 1):
            jobs.append((j, k, pos))
            pos += 1 if abs(k) == 2 ** j else 2
    assert pos == nsh

    def one(job):
        j, k, pos = job
        a = 2.0 ** (-2 * j)
        s = k * 2.0 ** (-j)
        p_hor = _shearlet_spect(xi_x, xi_y, a, s)
        p_ver = _shearlet_spect(xi_y, xi_x, a, s)
        if abs(k) == 2 ** j:            # seam: one element glued from both cones
            put(pos, p_hor * hor + p_ver * ver)
        else:
            put(pos, p_hor)
            put(pos + 1, p_ver)

    workers = max(1, min(8, os.cpu_count() or 1)) if nil * nxl >= (1 << 16) else 1
    if workers > 1:
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(one, jobs))
    else:
        for job in jobs:
            one(job)
    # even extent: row / column 0 of the centred grid is the Nyquist line, which has no mirror partner on the grid; pair it with
    # itself (finest scale only -- coarser shearlets vanish there) so that Psi(-xi) = Psi(xi) and the frame identity survives
    first = 1 + sum(per_scale[:-1])
    # (root mean square of a sample and its mirror image: equals (a + b)/sqrt(2) of the tutorial where only one of the two is
    # non-zero, and leaves the symmetric k = 0 element alone)
    if nil % 2 == 0:
        c0 = 1 - nxl % 2                     # the mirror of column c is nxl - c (even extent) or nxl - 1 - c (odd extent)
        row = psi[first:, 0, c0:].copy()
        psi[first:, 0, c0:] = np.sqrt(0.5 * (row ** 2 + row[:, ::-1] ** 2))
    if nxl % 2 == 0:
        r0 = 1 - nil % 2
        col = psi[first:, r0:, 0].copy()
        psi[first:, r0:, 0] = np.sqrt(0.5 * (col ** 2 + col[:, ::-1] ** 2))
    for i in range(nsh):
        psi[i] = np.fft.ifftshift(psi[i])
    view = np.moveaxis(psi, 0, -1)
    return np.ascontiguousarray(view) if contiguous else view


# ---- the transform pair handed to POCS_algorithm ------------------------------------------------------------------------
def shearlet_transform(x, Psi=None):
    """ST[..., s] = ifft2(Psi_s * fft2(x)); real slices give real coefficients."""
    st = np.fft.ifft2(Psi * np.fft.fft2(x)[..., None], axes=(0, 1))
    return st if np.iscomplexobj(x) else st.real


def inverse_shearlet_transform(ST, Psi=None):
    """x = ifft2(sum_s fft2(ST_s) * Psi_s); real coefficients give a real slice."""
    x = np.fft.ifft2((np.fft.fft2(ST, axes=(0, 1)) * Psi).sum(axis=-1))
    return x if np.iscomplexobj(ST) else x.real


# ---- the reference's SHEARLET branches --------------------------------------------------------------------------------
def shearlet_schedule(thresh_model, niter, p_max, p_min, st, kind="values"):
    """tau[k, s] (POCS.py:251-274 with :256-259; :277-354 with :282-285, :302-320, :340-341)."""
    steps = np.arange(1, niter + 1)[:, None]
    if "inverse" in thresh_model and "proportional" in thresh_model:
        hi = np.max(np.abs(st), axis=(0, 1))
        lo = np.min(np.abs(st), axis=(0, 1))
        q = base._suffix_number(thresh_model)
        nq = niter ** q
        return (nq * (hi - lo)) / (nq - 1) / (steps ** q) + (nq * lo - hi) / (nq - 1)
    if kind == "values":
        peak = np.max(st, axis=(0, 1))                      # signed (real) / lexicographic (complex) maximum per shearlet
        if isinstance(p_min, str) and p_min == "adaptive":
            nscales = number_of_scales(st.shape)            # POCS.py:304 passes the 3-D array: max over (nil, nxl, nsh)
            j = np.hstack((np.array([0]), np.repeat(np.arange(1, nscales + 1), [2 ** (i + 2) for i in range(nscales)])))
            tau_lo = 1 / 3 * np.median(np.log10(j + 1) * np.sqrt(np.linalg.norm(st, axis=(0, 1)) ** 2 / st.size))
        else:
            tau_lo = p_min * peak
        tau_hi = p_max * peak
    elif kind == "factors":
        tau_hi, tau_lo = p_max, p_min
    else:
        raise ValueError('Parameter `kind` only supports arguments "values" or "factors"')
    ramp = (steps - 1) / (niter - 1)
    if thresh_model == "linear":
        return tau_hi - (tau_hi - tau_lo) * ramp
    if "exponential" in thresh_model:
        q = base._suffix_number(thresh_model, strict=True)
        return tau_hi * np.exp(np.log(tau_lo / tau_hi) * ramp ** q)
    raise NotImplementedError(f"{thresh_model} is not implemented for SHEARLET transform!")


def pocs_slice_shearlet(x, mask, Psi, niter=50, thresh_op="hard", thresh_model="exponential", eps=1e-9, alpha=1.0, p_max=0.99,
                        p_min=1e-5, sqrt_decay=False, decay_kind="values", version="regular", info=None):
    """The per-slice loop (POCS.py:371-656) with transform_kind='SHEARLET'."""
    if np.max(mask) > 1:
        raise ValueError(f"mask should be quasi-boolean (0 or 1) but has maximum of {np.max(mask)}")
    if Psi is None:
        raise ValueError("SHEARLET requires pre-computed shearlets in Fourier domain (Psi)")
    niter, eps, p_max, alpha = int(niter), float(eps), float(p_max), float(alpha)
    complex_in = np.iscomplexobj(x)
    if np.count_nonzero(x) == 0:
        if isinstance(info, dict):
            info.update(niterations=0, costs=[0], tau=None, runtime=0.0)
        return x
    tau = shearlet_schedule(thresh_model, niter, p_max, p_min, shearlet_transform(x, Psi), decay_kind)
    prev = cur = x
    costs = []
    t0 = time.perf_counter()
    k = -1
    for k in range(niter):
        if version in ("regular", "fast"):
            feed = prev
        elif version == "adaptive":
            feed = alpha * x + (1 - alpha * mask) * prev + (1 - alpha) * (x - mask * prev)
        else:
            raise ValueError(version)
        st = shearlet_transform(feed, Psi)
        level = np.sqrt(tau[k]) if sqrt_decay else tau[k]
        cur = inverse_shearlet_transform(base.apply_threshold(st, level, kind=thresh_op), Psi)
        cur = cur * (1 - alpha * mask) + x * alpha
        cost = np.sum(np.abs(cur) - np.abs(prev)) ** 2 / np.sum(np.abs(cur)) ** 2
        costs.append(cost)
        prev = cur
        if k > 2 and cost < eps:
            break
    if isinstance(info, dict):
        info.update(niterations=k + 1, costs=costs, tau=tau, runtime=time.perf_counter() - t0)
    return cur if complex_in else np.real(cur)


def pocs_cube_shearlet(cube, mask, Psi, infos=None, **params):
    cube = np.asarray(cube)
    out = np.empty_like(cube)
    for s in range(cube.shape[0]):
        info = {} if infos is not None else None
        out[s] = pocs_slice_shearlet(cube[s], mask, Psi, info=info, **params)
        if infos is not None:
            infos.append(info)
    return out


# ---- the same loop for large REAL slices, evaluated with multi-threaded real transforms ------------------------------------------
def real_transform_pair(Psi, shape, workers=-1):
    """(fwd, inv) of the frame for REAL slices of `shape` and real spectra with Psi_s(-k) = Psi_s(k): coefficients [s][i][j] through
    `scipy.fft.rfft2 / irfft2` on the half spectrum (see pocs_slice_shearlet_real)."""
    import scipy.fft as sf
    if np.iscomplexobj(Psi) or len(shape) != 2:
        raise ValueError("real 2-D slice and real spectra expected")
    n1, n2 = shape
    half = np.ascontiguousarray(np.moveaxis(np.asarray(Psi)[:, : n2 // 2 + 1, :], -1, 0), dtype=np.float64)   # [s][k1][k2 <= n2/2]

    def fwd(v):
        return sf.irfft2(half * sf.rfft2(v)[None], s=(n1, n2), axes=(1, 2), workers=workers)                    # [s][i][j]

    def inv(st):
        return sf.irfft2((sf.rfft2(st, axes=(1, 2), workers=workers) * half).sum(axis=0), s=(n1, n2))

    return fwd, inv


def shearlet_step_real(prev, x, mask, Psi, tau_k, thresh_op="hard", alpha=1.0, workers=-1, pair=None):
    """ONE iteration (POCS.py:589-619, version 'regular') from the real iterate `prev`: returns the new iterate, the coefficients
    [s][i][j] before the threshold and after it.  tau_k: one threshold per shearlet."""
    prev = np.asarray(prev, dtype=np.float64)
    fwd, inv = pair if pair is not None else real_transform_pair(Psi, prev.shape, workers)
    st = fwd(prev)
    shr = base.apply_threshold(st, np.asarray(tau_k)[:, None, None], kind=thresh_op)
    nxt = inv(shr) * (1.0 - alpha * np.asarray(mask, dtype=np.float64)) + np.asarray(x, dtype=np.float64) * alpha
    return nxt, st, shr


def pocs_slice_shearlet_real(x, mask, Psi, niter=50, thresh_op="hard", thresh_model="exponential", alpha=1.0, p_max=0.99,
                             p_min=1e-5, workers=-1, info=None):
    """`pocs_slice_shearlet(..., version='regular', eps=0)` for a REAL slice and real spectra with Psi_s(-k) = Psi_s(k) (what
    `scales_shears_and_spectra` builds): ST_s = ifft2(Psi_s fft2(x)) is then real and equals irfft2 of the half spectrum, so the
    loop runs on `scipy.fft.rfft2 / irfft2` with `workers` threads -- half the arithmetic of the complex transforms and all cores,
    which is what makes BASELINE configs[4]'s slice (2048 x 1024, 125 shearlets: 250 two-million-point transforms per iteration)
    checkable in a test.  Same arithmetic otherwise (float64, POCS.py:589-619); tests/test_gpu_shearlet.py holds it to
    `pocs_slice_shearlet` on a small slice before using it."""
    x = np.asarray(x, dtype=np.float64)
    fwd, inv = real_transform_pair(Psi, x.shape, workers)
    st0 = fwd(x)
    tau = shearlet_schedule(thresh_model, int(niter), float(p_max), p_min, np.moveaxis(st0, 0, -1), "values")  # (niter, nsh)
    w = 1.0 - alpha * np.asarray(mask, dtype=np.float64)
    cur, costs = x, []
    for k in range(int(niter)):
        st = st0 if k == 0 else fwd(cur)
        shr = base.apply_threshold(st, tau[k][:, None, None], kind=thresh_op)
        nxt = inv(shr) * w + x * alpha
        costs.append(np.sum(np.abs(nxt) - np.abs(cur)) ** 2 / np.sum(np.abs(nxt)) ** 2)
        cur = nxt
    if isinstance(info, dict):
        info.update(niterations=int(niter), costs=costs, tau=tau)
    return cur
