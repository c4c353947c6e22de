"""
CPU oracle for the POCS hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch NumPy restatement of the algorithm implemented by the
reference in ``pseudo_3D_interpolation/functions/POCS.py`` and
``pseudo_3D_interpolation/functions/threshold_operator.py`` (reference paths are relative to
/root/reference).  It exists so that the HIP product path can be checked against something
that runs anywhere.  It is NOT part of the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
    may import it -- as the checker / the reported CPU baseline, never as the thing shipped;
  * the product package (``pseudo-3d-interpolation_amd/``) never imports anything from
    ``oracle/`` and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the reference itself in the
build container (``PYTHONPATH=/root/reference``, NumPy 2.2.6) and wrote the fixtures under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below against
them (bit-exact for the complex128 cases, <=1e-6 for the complex64 ones).

NumPy quirks of the reference that are reproduced on purpose:
  * the 'values' schedule scales by ``x_fwd.max()`` of a COMPLEX array -> lexicographic max
    (largest real part, ties broken by imaginary part), so tau is complex (POCS.py:288);
  * comparisons / clipping against a complex tau are lexicographic
    (threshold_operator.py:37-39, 76-78, 111);
  * ``version='fast'`` never differs from ``'regular'``: the momentum term is computed from
    two names bound to the same array (POCS.py:549-550, 566-571, 629);
  * the cost is ((sum|x_k| - sum|x_{k-1}|) / sum|x_k|)**2 (POCS.py:622).
"""
from __future__ import annotations

import time

import numpy as np

TRANSFORMS = ("FFT", "WAVELET", "SHEARLET", "CURVELET", "DCT")
_SINGLE_SCALE = ("FFT", "CURVELET", "DCT")


# --------------------------------------------------------------------------------------------
# threshold operators (threshold_operator.py:9-112; dispatch POCS.py:61-102)
# --------------------------------------------------------------------------------------------
def shrink_hard(coef, tau, fill=0):
    """Zero every coefficient whose modulus is 'less' than tau (threshold_operator.py:87-112).

    ``np.less`` on (real, complex) operands is lexicographic, which is what the reference gets
    when tau is complex.
    """
    coef = np.asarray(coef)
    return np.where(np.less(np.absolute(coef), tau), fill, coef)


def _shrink_scaled(coef, ratio, tau, fill):
    # common tail of soft / garrote: gain = clip(1 - ratio, 0, None); coef * gain
    coef = np.asarray(coef)
    mag = np.absolute(coef)
    with np.errstate(divide="ignore", invalid="ignore"):
        gain = 1 - ratio(mag)
        gain.clip(min=0, max=None, out=gain)
        shrunk = coef * gain
    if fill == 0:
        return shrunk
    return np.where(np.less(mag, tau), fill, shrunk)


def shrink_soft(coef, tau, fill=0):
    """coef * max(1 - tau/|coef|, 0)  (threshold_operator.py:9-45)."""
    return _shrink_scaled(coef, lambda mag: tau / mag, tau, fill)


def shrink_garrote(coef, tau, fill=0):
    """coef * max(1 - tau^2/|coef|^2, 0)  (threshold_operator.py:48-84)."""
    return _shrink_scaled(coef, lambda mag: tau**2 / mag**2, tau, fill)


_OPS = {
    "hard": shrink_hard,
    "soft": shrink_soft,
    "garrote": shrink_garrote,
    "garotte": shrink_garrote,
}


def apply_threshold(coef, tau, kind="soft", fill=0):
    """Dispatch on the operator name (POCS.py:61-102).

    ``*-percentile`` variants interpret ``tau`` as a percentile of |coef| (POCS.py:43-58).
    An unknown name yields ``None`` exactly like the reference.
    """
    coef = np.asarray(coef)
    if kind.endswith("-percentile"):
        base = kind[: -len("-percentile")]
        if base not in _OPS:
            return None
        return _OPS[base](coef, np.percentile(np.abs(coef), tau), fill)
    if kind not in _OPS:
        return None
    return _OPS[kind](coef, tau, fill)


# --------------------------------------------------------------------------------------------
# threshold schedule (POCS.py:169-368), single-scale transforms only (FFT / DCT / CURVELET)
# --------------------------------------------------------------------------------------------
def _suffix_number(name, default=1.0, strict=False):
    if "-" not in name:
        return default
    tail = name.split("-")[-1]
    if strict:
        return float(tail)
    try:
        return float(tail)
    except Exception:  # the reference swallows everything here (POCS.py:267-270)
        return default


def threshold_schedule(
    thresh_model,
    niter,
    transform_kind="FFT",
    p_max=0.99,
    p_min=1e-3,
    x_fwd=None,
    kind="values",
):
    """tau_k for k = 1..niter (POCS.py:169-368) for the single-scale transforms."""
    if transform_kind is not None:
        if transform_kind.upper() not in TRANSFORMS and (kind == "values" or thresh_model == "data-driven"):
            raise ValueError(f"Unsupported transform. Please select one of: {TRANSFORMS}")
        transform_kind = transform_kind.upper()
        if transform_kind not in _SINGLE_SCALE:
            raise NotImplementedError("oracle covers single-scale transforms only")
    if x_fwd is None and (kind == "values" or thresh_model == "data-driven"):
        raise ValueError('`x_fwd` must be specified for thresh_model="data-driven" or kind="values"!')

    steps = np.arange(1, niter + 1)

    # (A) a/k^q + b, fitted through max|X0| at k=1 and min|X0| at k=niter (POCS.py:251-274)
    if "inverse" in thresh_model and "proportional" in thresh_model:
        hi = np.abs(x_fwd).max()
        lo = np.abs(x_fwd).min()
        q = _suffix_number(thresh_model)
        nq = niter**q
        a = (nq * (hi - lo)) / (nq - 1)
        b = (nq * lo - hi) / (nq - 1)
        return a / (steps**q) + b

    # (B) end points of the classic models (POCS.py:277-333)
    if kind == "values":
        if transform_kind is None:
            raise ValueError('`transform_kind` must be specified for thresh_model="data-driven" or kind="values"!')
        peak = x_fwd.max()  # complex input -> lexicographic max, keeps its imaginary part
        if isinstance(p_min, str) and p_min == "adaptive":
            if transform_kind not in ("FFT", "DCT"):
                raise NotImplementedError(f"p_min=`adaptive` is not implemented for {transform_kind} transform")
            tau_lo = 0.01 * np.sqrt(np.linalg.norm(x_fwd, axis=None) ** 2 / x_fwd.size)
        else:
            tau_lo = p_min * peak
        tau_hi = p_max * peak
    elif kind == "factors":
        tau_hi, tau_lo = p_max, p_min
    else:
        raise ValueError('Parameter `kind` only supports arguments "values" or "factors"')

    ramp = (steps - 1) / (niter - 1)  # 0 .. 1 (niter == 1 -> nan, as in the reference)

    if thresh_model == "linear":
        return tau_hi - (tau_hi - tau_lo) * ramp
    if "exponential" in thresh_model:
        q = _suffix_number(thresh_model, strict=True)
        return tau_hi * np.exp(np.log(tau_lo / tau_hi) * ramp**q)
    if thresh_model == "data-driven" and transform_kind in _SINGLE_SCALE:
        tau = np.zeros((steps.size,), dtype=x_fwd.dtype)
        inside = (x_fwd > tau_lo) & (x_fwd < tau_hi)  # lexicographic on complex data
        ranked = np.sort(x_fwd[inside])[::-1]
        tau[0] = ranked[0]
        tau[1:] = ranked[np.ceil((steps[1:] - 1) * (ranked.size - 1) / (niter - 1)).astype("int")]
        return tau
    raise NotImplementedError(f"{thresh_model} is not implemented for {transform_kind} transform!")


# --------------------------------------------------------------------------------------------
# the per-slice loop (POCS.py:371-656), FFT transform
# --------------------------------------------------------------------------------------------
def pocs_slice(
    x,
    mask,
    niter=50,
    thresh_op="hard",
    thresh_model="exponential",
    eps=1e-9,
    alpha=1.0,
    p_max=0.99,
    p_min=1e-5,
    sqrt_decay=False,
    decay_kind="values",
    version="regular",
    transform_kind="FFT",
    fwd=np.fft.fft2,
    inv=np.fft.ifft2,
    info=None,
):
    """One 2-D slice through the weighted POCS iteration.  Returns the reconstructed slice.

    ``info`` (a dict, optional) receives ``niterations``, ``costs`` (list), ``tau`` (schedule)
    and ``runtime``.
    """
    if np.max(mask) > 1:  # POCS.py:488-489
        raise ValueError(f"mask should be quasi-boolean (0 or 1) but has maximum of {np.max(mask)}")
    if fwd is None or inv is None:  # POCS.py:491-492
        raise ValueError("Forward and inverse transform function have to be supplied")
    if transform_kind.upper() not in TRANSFORMS:  # POCS.py:494-498
        raise ValueError(f"Unsupported transform. Please select one of: {TRANSFORMS}")
    transform_kind = transform_kind.upper()
    if transform_kind not in _SINGLE_SCALE:
        raise NotImplementedError("oracle covers single-scale transforms only")

    niter, eps, p_max, alpha = int(niter), float(eps), float(p_max), float(alpha)
    complex_in = np.iscomplexobj(x)

    if np.count_nonzero(x) == 0:  # empty slice is handed back untouched (POCS.py:515-521)
        if isinstance(info, dict):
            info.update(niterations=0, costs=[0], tau=None, runtime=0.0)
        return x

    tau = threshold_schedule(thresh_model, niter, transform_kind, p_max, p_min, fwd(x), decay_kind)

    prev = x  # x_{k-1}
    cur = x
    momentum = 1
    costs = []
    t0 = time.perf_counter()
    k = -1
    for k in range(niter):
        if version == "regular":
            feed = prev
        elif version == "fast":
            nxt = (1 + np.sqrt(1 + 4 * momentum**2)) / 2
            frac = (momentum - 1) / (nxt + 1)
            momentum = nxt
            feed = cur + frac * (cur - prev)  # cur IS prev at this point -> feed == cur (POCS.py:566-571)
        elif version == "adaptive":
            blend = alpha * x + (1 - alpha * mask) * prev
            feed = blend + (1 - alpha) * (x - mask * prev)
        else:
            raise UnboundLocalError(f"unknown version {version!r}")  # reference: x_input unbound

        spec = fwd(feed)
        level = np.sqrt(tau[k]) if sqrt_decay else tau[k]
        spec = apply_threshold(spec, level, kind=thresh_op)
        cur = inv(spec)
        cur *= 1 - alpha * mask  # keep the estimate where nothing was observed ...
        cur += x * alpha  # ... and put the observed traces back

        cost = np.sum(np.abs(cur) - np.abs(prev)) ** 2 / np.sum(np.abs(cur)) ** 2
        costs.append(cost)
        prev = cur
        if k > 2 and cost < eps:
            break

    if isinstance(info, dict):
        info.update(niterations=k + 1, costs=costs, tau=tau, runtime=time.perf_counter() - t0)
    return cur if complex_in else np.real(cur)


def pocs_step(prev, x, mask, tau, thresh_op="hard", alpha=1.0, keep=None, fwd=np.fft.fft2, inv=np.fft.ifft2):
    """ONE regular iteration (POCS.py:592-619) from the iterate ``prev``.

    ``keep`` (bool array) overrides the keep/zero decision per coefficient: kept coefficients get the
    operator's (unclipped) gain, the others become 0.  Tests use it to replay the device's decisions
    for coefficients whose modulus is within float32 rounding of Re(tau), where no single-precision
    implementation (the reference's own complex64 path included) can be expected to decide like a
    double-precision one.  With the reference's complex tau all three operators are discontinuous
    there: hard jumps by |X|, soft and garrote by ~|Im tau|.
    Returns (x_new, spectrum before thresholding, thresholded spectrum).
    """
    spec = fwd(prev)
    if keep is None:
        shr = apply_threshold(spec, tau, kind=thresh_op)
    else:
        mag = np.absolute(spec)
        with np.errstate(divide="ignore", invalid="ignore"):
            if thresh_op == "hard":
                gain = 1.0
            elif thresh_op == "soft":
                gain = 1 - tau / mag
            elif thresh_op in ("garrote", "garotte"):
                gain = 1 - tau**2 / mag**2
            else:
                raise ValueError(thresh_op)
            shr = np.where(keep, spec * gain, 0)
    cur = inv(shr)
    cur *= 1 - alpha * mask
    cur += x * alpha
    return cur, spec, shr


def pocs_cube(cube, mask, infos=None, **params):
    """Apply :func:`pocs_slice` to every leading-axis slice of ``cube`` (the job the reference's
    ``xr.apply_ufunc(..., vectorize=True)`` does, cube_POCS_interpolation_3D.py:314-340); the
    result is cast back to the input dtype like ``np.vectorize(otypes=...)`` does there."""
    cube = np.asarray(cube)
    out = np.empty_like(cube)
    for s in range(cube.shape[0]):
        info = {} if infos is not None else None
        out[s] = pocs_slice(cube[s], mask, info=info, **params)
        if infos is not None:
            infos.append(info)
    return out


# --------------------------------------------------------------------------------------------
# synthetic workload shared by oracle, tests and bench
# --------------------------------------------------------------------------------------------
def synthetic_slice(nil, nxl, s, real=False):
    rng = np.random.default_rng(1234 + s)
    il = np.arange(nil)[:, None] / nil
    xl = np.arange(nxl)[None, :] / nxl
    acc = np.zeros((nil, nxl), dtype=np.complex128)
    for _ in range(6):
        k1 = rng.integers(-(nil // 8), max(nil // 8, 1))
        k2 = rng.integers(-(nxl // 8), max(nxl // 8, 1))
        amp = rng.standard_normal() + 1j * rng.standard_normal()
        acc += amp * np.exp(2j * np.pi * (k1 * il + k2 * xl))
    acc += 0.01 * (rng.standard_normal((nil, nxl)) + 1j * rng.standard_normal((nil, nxl)))
    return acc.real.astype(np.float32) if real else acc.astype(np.complex64)


def synthetic_mask(nil, nxl, missing):
    return (np.random.default_rng(42).random((nil, nxl)) >= missing).astype(np.uint8)


def synthetic_cube(nil, nxl, nslices, missing, real=False, first=0):
    mask = synthetic_mask(nil, nxl, missing)
    full = np.stack([synthetic_slice(nil, nxl, first + s, real) for s in range(nslices)])
    return full, mask, full * mask
