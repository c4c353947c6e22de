"""
POCS interpolation on MI355X -- host-side mirror of ``pseudo_3D_interpolation/functions/POCS.py``.

Same public names and keyword arguments as the reference module
(``POCS_algorithm`` POCS.py:371-391, ``POCS`` / ``FPOCS`` / ``APOCS`` POCS.py:659-661,
``get_threshold_decay`` POCS.py:169-177) so existing callers -- in particular
``xr.apply_ufunc(POCS, ...)`` of the step-13 driver (cube_POCS_interpolation_3D.py:314-340) --
keep working, plus :func:`pocs_cube`, which hands a whole batch of slices to the GPU at once.

What runs where
---------------
* GPU (libp3d_hip.so, hand-written HIP): the forward transform of the observed slice and its
  statistics, and the whole iteration loop (forward FFT2, thresholding -- for the ``-percentile``
  operators including the exact ``np.percentile`` of the spectrum moduli --, inverse FFT2,
  re-insertion of the observed traces, cost sums, early exit).
* Host (this file, a few scalars per slice): argument checks and the threshold schedule
  tau_k, k = 1..niter, evaluated with NumPy in complex128 exactly like POCS.py:251-368 so that the
  reference's NumPy conventions (lexicographic complex max, complex log/exp) carry over.

There is no CPU fallback for the loop: without the library or without a GPU these functions raise.
"""
import os
import time
from functools import partial

import numpy as np

from .. import _ffi
from .shearlets import get_number_scales

TRANSFORMS = ('FFT', 'WAVELET', 'SHEARLET', 'CURVELET', 'DCT')
_HIP_TRANSFORMS = ('FFT', 'WAVELET', 'SHEARLET')
_WAVELET_OPS = ('soft', 'hard', 'garrote', 'garotte')
_THRESH_OPS = ('soft', 'hard', 'garrote', 'garotte', 'soft-percentile', 'hard-percentile',
               'garrote-percentile', 'garotte-percentile')


# =================================================================================================
#                                      threshold schedule (host)
# =================================================================================================
def _tail_number(name, default=1.0, strict=False):
    """Number after the last '-' of a model name ('exponential-2' -> 2.0)."""
    if '-' not in name:
        return default
    tail = name.split('-')[-1]
    if strict:  # POCS.py:352 lets a malformed suffix raise
        return float(tail)
    try:  # POCS.py:266-270 falls back to 1
        return float(tail)
    except Exception:  # noqa
        return default


def _schedule(thresh_model, niter, p_max, p_min, kind, peak, abs_max, abs_min, sumsq, size, transform_kind='FFT', tau_min=None):
    """tau[..., k] for k = 1..niter from the statistics of X0 = transform(x).

    ``peak`` is the (lexicographic) complex maximum of X0, ``abs_max`` / ``abs_min`` the extrema of
    |X0|, ``sumsq`` = sum |X0|^2, ``size`` = X0.size.  All of them may carry leading batch axes
    (one entry per slice); the schedule is broadcast along a new last axis.  Follows
    POCS.py:251-274 (inverse proportional), :277-333 (end points), :336-354 (linear, exponential).
    """
    steps = np.arange(1, niter + 1)

    if all(s in thresh_model for s in ['inverse', 'proportional']):
        q = _tail_number(thresh_model)
        nq = niter ** q
        hi = np.asarray(abs_max)[..., None]
        lo = np.asarray(abs_min)[..., None]
        a = (nq * (hi - lo)) / (nq - 1)
        b = (nq * lo - hi) / (nq - 1)
        return a / (steps ** q) + b

    if kind == 'values':
        peak = np.asarray(peak)[..., None]
        if tau_min is not None:        # SHEARLET with p_min='adaptive': one value per slice (POCS.py:302-320)
            tau_min = np.asarray(tau_min)[..., None]
        elif isinstance(p_min, str) and p_min == 'adaptive':
            if transform_kind == 'WAVELET':  # POCS.py:322-325
                raise NotImplementedError(f'p_min=`adaptive` is not implemented for {transform_kind} transform')
            tau_min = 0.01 * np.sqrt(np.asarray(sumsq)[..., None] / size)
        else:
            tau_min = p_min * peak
        tau_max = p_max * peak
    elif kind == 'factors':
        tau_max, tau_min = p_max, p_min
    else:
        raise ValueError('Parameter `kind` only supports arguments "values" or "factors"')

    with np.errstate(invalid='ignore', divide='ignore'):
        ramp = (steps - 1) / (niter - 1)
        if thresh_model == 'linear':
            return tau_max - (tau_max - tau_min) * ramp
        if 'exponential' in thresh_model:
            q = _tail_number(thresh_model, strict=True)
            return tau_max * np.exp(np.log(tau_min / tau_max) * ramp ** q)
    raise NotImplementedError(f'{thresh_model} is not implemented for {transform_kind} transform!')


def _data_driven(x_fwd, niter, p_max, p_min):
    """'data-driven' model (POCS.py:356-362): thresholds picked from the sorted coefficients."""
    steps = np.arange(1, niter + 1)
    peak = x_fwd.max()
    if isinstance(p_min, str) and p_min == 'adaptive':
        tau_min = 0.01 * np.sqrt(np.linalg.norm(x_fwd, axis=None) ** 2 / x_fwd.size)
    else:
        tau_min = p_min * peak
    tau_max = p_max * peak
    tau = np.zeros((steps.size,), dtype=x_fwd.dtype)
    chosen = np.sort(x_fwd[(x_fwd > tau_min) & (x_fwd < tau_max)])[::-1]
    tau[0] = chosen[0]
    tau[1:] = chosen[np.ceil((steps[1:] - 1) * (chosen.size - 1) / (niter - 1)).astype('int')]
    return tau


def _data_driven_batch(plan, chunk, active, niter, p_max, p_min, x_dev=None):
    """'data-driven' schedules of a batch of slices (POCS.py:356-362).  The spectrum stays on the device: sorted there in NumPy's
    complex order, the bounds formed here exactly as the reference forms them (p * x_fwd.max(): a Python float times a complex64
    scalar), the picks read back.  p_min='adaptive' takes its bound from np.linalg.norm of the spectrum: that one is computed on
    the downloaded spectrum as before."""
    n = chunk.shape[0]
    tau = np.zeros((n, niter), np.complex128)
    live = np.flatnonzero(active)
    if live.size == 0:
        return tau
    if isinstance(p_min, str):
        X0 = plan.fft2(chunk.astype(np.complex64))
        for s in live:
            tau[s] = _data_driven(X0[s], niter, p_max, p_min)
        return tau
    # x_fwd.max() per slice, complex64 (x_dev: the batch is on the device already, as complex64)
    peaks = plan.sorted_spectrum(chunk) if x_dev is None else plan.sorted_spectrum_dev(x_dev, n)
    lo = np.asarray([p_min * pk for pk in peaks])            # weak Python scalar times complex64 scalar -> complex64 (POCS.py:293-294)
    hi = np.asarray([p_max * pk for pk in peaks])
    picks, count = plan.data_driven_pick(lo, hi, niter)
    if np.any(count[live] == 0):
        raise IndexError('index 0 is out of bounds for axis 0 with size 0')   # v[0] of an empty selection (POCS.py:360)
    tau[live] = picks[live]
    return tau


def get_threshold_decay(
    thresh_model,
    niter: int,
    transform_kind: str = None,
    p_max: float = 0.99,
    p_min: float = 1e-3,
    x_fwd=None,
    kind: str = 'values',
):
    """
    Iteration-based decay of the threshold (same signature as the reference, POCS.py:169-177).

    ``x_fwd`` is the forward-transformed input as a NumPy array; only the single-scale transforms
    (`FFT`, `DCT`, `CURVELET`) and `WAVELET` (``x_fwd`` = list of detail tuples) are covered.  Returns ``tau`` with ``niter`` entries: complex when the
    'values' kind scales by the (complex) maximum of ``x_fwd``, real otherwise.
    """
    if transform_kind is None:
        pass
    elif transform_kind.upper() not in TRANSFORMS and (kind == 'values' or thresh_model == 'data-driven'):
        raise ValueError(f'Unsupported transform. Please select one of: {TRANSFORMS}')
    else:
        transform_kind = transform_kind.upper()
    if transform_kind == 'SHEARLET' and x_fwd is not None:
        # x_fwd: (nil, nxl, nsh) coefficients -> tau (niter, nsh); maxima per shearlet over axes (0, 1) (POCS.py:256-259, 282-285)
        x_fwd = np.asarray(x_fwd)
        st = np.empty((1, x_fwd.shape[-1], 5))
        peak = np.max(x_fwd, axis=(0, 1))
        st[0, :, 0], st[0, :, 1] = peak.real, peak.imag if np.iscomplexobj(peak) else 0.0
        mag = np.abs(x_fwd)
        st[0, :, 2], st[0, :, 3], st[0, :, 4] = mag.max(axis=(0, 1)), mag.min(axis=(0, 1)), (mag ** 2).sum(axis=(0, 1))
        tau = _shearlet_schedule_from_stats(st, x_fwd.shape[:2], thresh_model, niter, p_max, p_min, kind)
        return tau[0] if np.ndim(tau) == 3 else tau
    if transform_kind == 'WAVELET' and x_fwd is not None:
        # x_fwd: list of (cH, cV, cD) per level (the low-pass array already removed, POCS.py:524-525) -> tau (niter, nlev, 3)
        inverse_prop = all(s in thresh_model for s in ['inverse', 'proportional'])
        peak = abs_max = abs_min = None
        if inverse_prop:
            abs_max = np.asarray([[np.abs(d).max() for d in level] for level in x_fwd])
            abs_min = np.asarray([[np.abs(d).min() for d in level] for level in x_fwd])
        elif kind == 'values':
            peak = np.asarray([[np.asarray(d).max() for d in level] for level in x_fwd])
        tau = _schedule(thresh_model, niter, p_max, p_min, kind, peak, abs_max, abs_min, None, None, 'WAVELET')
        return np.moveaxis(tau, -1, 0) if np.ndim(tau) == 3 else np.reshape(tau, (-1, 1, 1))

    if x_fwd is None and (kind == 'values' or thresh_model == 'data-driven'):
        raise ValueError('`x_fwd` must be specified for thresh_model="data-driven" or kind="values"!')

    inverse_prop = all(s in thresh_model for s in ['inverse', 'proportional'])
    if kind == 'values' and transform_kind is None and not inverse_prop:
        raise ValueError('`transform_kind` must be specified for thresh_model="data-driven" or kind="values"!')
    if thresh_model == 'data-driven' and not inverse_prop:
        if kind != 'values':
            raise NotImplementedError('data-driven thresholds need kind="values"')
        return _data_driven(np.asarray(x_fwd), niter, p_max, p_min)

    peak = abs_max = abs_min = sumsq = size = None
    if x_fwd is not None:
        x_fwd = np.asarray(x_fwd)
        if inverse_prop:
            mag = np.abs(x_fwd)
            abs_max, abs_min = mag.max(), mag.min()
        elif kind == 'values':
            peak = x_fwd.max()  # complex -> lexicographic, keeps the imaginary part (POCS.py:288)
            if isinstance(p_min, str) and p_min == 'adaptive':
                sumsq, size = np.linalg.norm(x_fwd, axis=None) ** 2, x_fwd.size
    return _schedule(thresh_model, niter, p_max, p_min, kind, peak, abs_max, abs_min, sumsq, size)


def _schedule_from_stats(stats, size, thresh_model, niter, p_max, p_min, kind):
    """Batch form used by the GPU path: ``stats`` is (nslices, 6) from ``p3d_pocs_stats``."""
    peak = stats[:, 0] + 1j * stats[:, 1]
    tau = _schedule(thresh_model, niter, p_max, p_min, kind, peak, stats[:, 2], stats[:, 3], stats[:, 4], size)
    return np.broadcast_to(tau, (stats.shape[0], niter))


# =================================================================================================
#                                      batched GPU entry point
# =================================================================================================
_plans = {}
_workers = {}


def _get_plan(nil, nxl, nslices, device, slot=0):
    key = (nil, nxl, device, slot)
    plan = _plans.get(key)
    if plan is None or plan.max_slices < nslices:
        if plan is not None:
            plan.close()
        plan = _ffi.Plan(nil, nxl, max(nslices, 1), device)
        _plans[key] = plan
    return plan


def _get_wavelet_plan(nil, nxl, nslices, wavelet, device):
    key = ('wavelet', nil, nxl, str(wavelet), device)
    plan = _plans.get(key)
    if plan is None or plan.max_slices < nslices:
        if plan is not None:
            plan.close()
        plan = _ffi.WaveletPlan(nil, nxl, max(nslices, 1), wavelet=wavelet, device=device)
        _plans[key] = plan
    return plan


def _get_wavelet_plan64(nil, nxl, nslices, wavelet, device):
    key = ('wavelet64', nil, nxl, str(wavelet), device)
    plan = _plans.get(key)
    if plan is None or plan.max_slices < nslices:
        if plan is not None:
            plan.close()
        plan = _ffi.WaveletPlan64(nil, nxl, max(nslices, 1), wavelet=wavelet, device=device)
        _plans[key] = plan
    return plan


def _shearlet_adaptive_tau_min(sumsq, shape2d):
    """1/3 median_s(log10(j_s + 1) sqrt(||ST_s||^2 / ST.size)) with j_s the scale of shearlet s (POCS.py:302-320)."""
    nsh = sumsq.shape[-1]
    nscales = get_number_scales(tuple(shape2d) + (nsh,))   # the reference passes the 3-D coefficient array
    j = np.hstack((np.array([0]), np.repeat(np.arange(1, nscales + 1), [2 ** (i + 2) for i in range(nscales)])))
    size = shape2d[0] * shape2d[1] * nsh
    return 1 / 3 * np.median(np.log10(j + 1) * np.sqrt(sumsq / size), axis=-1)


def _shearlet_schedule_from_stats(stats, shape2d, thresh_model, niter, p_max, p_min, kind):
    """stats (n, nsh, 5) from ``p3d_shearlet_stats`` -> tau (n, niter, nsh) (or (niter, 1) for kind='factors')."""
    peak = stats[..., 0] + 1j * stats[..., 1]
    if not np.any(stats[..., 1]):
        peak = peak.real
    tau_min = None
    inverse_prop = all(s in thresh_model for s in ['inverse', 'proportional'])
    if kind == 'values' and not inverse_prop and isinstance(p_min, str) and p_min == 'adaptive':
        tau_min = _shearlet_adaptive_tau_min(stats[..., 4], shape2d)[..., None]   # same value for every shearlet of a slice
    tau = np.asarray(_schedule(thresh_model, niter, p_max, p_min, kind, peak, stats[..., 2], stats[..., 3], None, None, 'SHEARLET', tau_min))
    if tau.ndim == 3:
        return np.moveaxis(tau, -1, 1)
    return np.reshape(tau, (niter, 1))


_shearlet_plans = {}


def _get_shearlet_plan(psi, nslices, device):
    """Plans hold the spectra on the device; they are cached per Psi array (identity + shape + a few samples)."""
    psi = np.asarray(psi)
    probe = psi[::max(psi.shape[0] // 7, 1), ::max(psi.shape[1] // 5, 1), ::max(psi.shape[2] // 3, 1)]
    key = (psi.shape, str(psi.dtype), float(np.sum(probe)), float(np.sum(np.abs(probe) ** 2)), device)
    plan = _shearlet_plans.get(key)
    if plan is None or plan.max_slices < nslices:
        if plan is not None:
            plan.close()
        plan = _ffi.ShearletPlan(psi, max_slices=max(nslices, 1), device=device)
        _shearlet_plans[key] = plan
    return plan


def _get_shearlet_plan64(psi, nslices, device):
    psi = np.asarray(psi)
    probe = psi[::max(psi.shape[0] // 7, 1), ::max(psi.shape[1] // 5, 1), ::max(psi.shape[2] // 3, 1)]
    key = ('double', psi.shape, str(psi.dtype), float(np.sum(probe)), float(np.sum(np.abs(probe) ** 2)), device)
    plan = _shearlet_plans.get(key)
    if plan is None or plan.max_slices < nslices:
        if plan is not None:
            plan.close()
        plan = _ffi.ShearletPlan64(psi, max_slices=max(nslices, 1), device=device)
        _shearlet_plans[key] = plan
    return plan


def release_plans():
    """Free the cached GPU plans (work buffers) of this process."""
    for w in _workers.values():
        w.close()
    _workers.clear()
    for bufs in _batch_bufs.values():
        for b in bufs:
            b.free()
    _batch_bufs.clear()
    for cache in (_plans, _shearlet_plans, _holders):
        for plan in cache.values():
            plan.close()
        cache.clear()


def _wavelet_name(transform, wavelet):
    """Wavelet of a ``partial(pywt.wavedec2, wavelet=..., mode='smooth')`` transform (cube_POCS_interpolation_3D.py:261-264)."""
    kw = getattr(transform, 'keywords', None) or {}
    if wavelet is None:
        wavelet = kw.get('wavelet', 'coif5')
    mode = kw.get('mode', 'smooth')
    if mode != 'smooth':
        raise NotImplementedError(f"wavelet signal extension mode {mode!r}: only 'smooth' (the workflow's) is implemented")
    return getattr(wavelet, 'name', wavelet)  # pywt.Wavelet objects carry .name


def _wavelet_schedule_from_stats(stats, thresh_model, niter, p_max, p_min, kind):
    """stats (n, nlev, 3, 4) from ``p3d_wavelet_stats`` -> tau (n, niter, nlev, 3)."""
    peak = stats[..., 0] + 1j * stats[..., 1]
    if not np.any(stats[..., 1]):
        peak = peak.real
    tau = _schedule(thresh_model, niter, p_max, p_min, kind, peak, stats[..., 2], stats[..., 3], None, None, 'WAVELET')
    tau = np.asarray(tau)
    if tau.ndim == 4:
        return np.moveaxis(tau, -1, 1)
    return np.broadcast_to(np.reshape(tau, (1, niter, 1, 1)), (stats.shape[0], niter) + stats.shape[1:3])


def _check_common(mask, transform_kind, thresh_op):
    if np.max(mask) > 1:
        raise ValueError(f'mask should be quasi-boolean (0 or 1) but has maximum of {np.max(mask)}')
    if transform_kind is None or transform_kind.upper() not in TRANSFORMS:
        raise ValueError(f'Unsupported transform. Please select one of: {TRANSFORMS}')
    kind = transform_kind.upper()
    if kind not in _HIP_TRANSFORMS:
        raise NotImplementedError(
            f'{kind} transform is not implemented by the HIP kernels yet (available: {_HIP_TRANSFORMS})')
    if thresh_op not in _THRESH_OPS:
        raise ValueError(f'Unknown threshold operator {thresh_op!r}. Please select one of: {_THRESH_OPS}')
    return kind


_copy_pool = None
_timeline = None     # a list while tools/e2e_timeline.py records the phases of the chunk workers (None: no recording)


def _slab_copy(dst, src):
    """dst[...] = src (same shape, any dtypes) in parallel slabs along axis 0: NumPy releases the GIL in the copy loops, one
    core moves ~10 GB/s, the PCIe link five times that."""
    global _copy_pool
    n = dst.shape[0]
    parts = min(_COPY_THREADS, n)
    if parts <= 1 or dst.nbytes < (32 << 20):
        np.copyto(dst, src, casting='unsafe')
        return
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _copy_pool = ThreadPoolExecutor(_COPY_THREADS)
    cuts = [n * i // parts for i in range(parts + 1)]
    list(_copy_pool.map(lambda i: np.copyto(dst[cuts[i]:cuts[i + 1]], src[cuts[i]:cuts[i + 1]], casting='unsafe'), range(parts)))


_COPY_THREADS = max(1, min(8, (os.cpu_count() or 2) // 2))
_holders = {}


def _holder(device):
    """A minimal FFT plan per device: owner of the device buffers and copies of the WAVELET / SHEARLET batches (``p3d_malloc`` / ``p3d_memcpy_*`` go
    through a plan's device and stream)."""
    h = _holders.get(device)
    if h is None or h.handle is None:
        h = _holders[device] = _ffi.Plan(4, 4, 1, device=device)
    return h


_batch_bufs = {}
_device_locks = {}
_device_locks_guard = __import__('threading').Lock()


def _device_lock(device):
    """One re-entrant lock per device: everything cached per device (plans, chunk workers, ``_batch_bufs``) is used under it."""
    with _device_locks_guard:
        lock = _device_locks.get(device)
        if lock is None:
            import threading
            lock = _device_locks[device] = threading.RLock()
        return lock


def _batch_buffers(device, cube_bytes, mask_bytes):
    """Device buffers (observed batch, result batch, mask) of the resident WAVELET / SHEARLET / double-precision batches, kept per device and grown
    on demand: a per-slice caller (``POCS_algorithm`` under ``xr.apply_ufunc``) does not pay three hipMalloc / hipFree pairs per call."""
    holder = _holder(device)
    have = _batch_bufs.get(device)
    if have is None or have[0].nbytes < cube_bytes or have[2].nbytes < mask_bytes or have[0].plan is not holder:
        if have is not None:
            for b in have:
                b.free()
        have = _batch_bufs[device] = (holder.alloc(cube_bytes), holder.alloc(cube_bytes), holder.alloc(mask_bytes))
    return have


def _active_slices(chunk):
    """``np.count_nonzero(x) != 0`` per slice (POCS.py:515-521), the slices spread over a few threads (NumPy's reductions release the GIL; one thread
    takes ~25 ms per 256 MiB)."""
    global _copy_pool
    n = chunk.shape[0]
    parts = max(1, min(_COPY_THREADS, n, chunk.nbytes >> 22))
    if parts == 1:
        return chunk.reshape(n, -1).any(axis=1)
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _copy_pool = ThreadPoolExecutor(_COPY_THREADS)
    cuts = [n * i // parts for i in range(parts + 1)]
    return np.concatenate(list(_copy_pool.map(lambda i: chunk[cuts[i]:cuts[i + 1]].reshape(cuts[i + 1] - cuts[i], -1).any(axis=1), range(parts))))
_PIN_MIN_BYTES = int(os.environ.get('P3D_PIN_MIN_MIB', 256)) << 20   # cubes from this size on are page-locked in place for the call (0 MiB: always; huge: never)


def _touch_pages(arr):
    """Write one byte into every 4-KiB page of a freshly allocated C-contiguous array from a few threads (its contents are about to
    be overwritten): the kernel hands out the pages in parallel instead of one by one under the first download."""
    global _copy_pool
    flat = arr.reshape(-1).view(np.uint8)
    parts = max(1, min(_COPY_THREADS, flat.size >> 22))
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _copy_pool = ThreadPoolExecutor(_COPY_THREADS)
    cuts = [flat.size * i // parts for i in range(parts + 1)]

    def touch(i):
        flat[cuts[i]:cuts[i + 1]:4096] = 0
    return [_copy_pool.submit(touch, i) for i in range(parts)]


class _HostPins:
    """The caller's cube and the result array of one ``pocs_cube`` call, page-locked in place (``_ffi.host_register``) for the
    duration of the call.  Both are registered in the background from the start of the call, CHUNK BY CHUNK in the order the workers
    take them (the cube at once; the result behind a few threads that touch its fresh pages): a worker waits for the slab it is about to
    move -- ``input_ready(i)`` / ``result_ready(i)`` -- not for the whole array.  That matters the first time memory is pinned:
    hipHostRegister of 4 GiB that never was takes ~90 ms and holds a lock of the runtime that every other HIP call of the process
    waits for; slab by slab the lock is held ~3 ms at a time and the uploads follow the registration front.  Everything is released
    at the end of the call.  A refused registration (memory the caller page-locked itself, a read-only mapping ...) is not an error:
    that slab simply stays pageable.  (Registering AND releasing chunk by chunk from the workers themselves was tried: it costs
    more than it hides -- 215 against 155 ms per call on BASELINE configs[2], profiles/r04_pcie_probe.txt.)"""

    def __init__(self, cube, out, starts, step):
        import threading
        self._regs = []              # (address, bytes) of the registered ranges
        self._lock = threading.Lock()
        self._slabs = [(lo, lo + step) for lo in starts]
        self._in_done = [threading.Event() for _ in self._slabs]
        self._out_done = [threading.Event() for _ in self._slabs]
        # (a caller's `out` that IS the cube -- an in-place call -- holds the input: its pages exist already and must not be written to)
        self._touching = [] if np.may_share_memory(cube, out) else _touch_pages(out)
        self._threads = [threading.Thread(target=self._pin, args=(cube, (), self._in_done), daemon=True),
                         threading.Thread(target=self._pin, args=(out, self._touching, self._out_done), daemon=True)]
        for t in self._threads:
            t.start()

    def _pin(self, arr, after, done):
        try:
            for f in after:
                f.result()
            # Every slab is registered as exactly the bytes a worker's copy moves.  (Page-aligned, non-overlapping ranges were tried in round 5: a
            # slab then starts in its predecessor's last page, and hipMemcpyAsync refuses a copy whose host range spans two registrations --
            # "invalid argument".  Neighbouring slabs sharing a boundary page is what the runtime accepts: the driver pins pages per
            # registration and counts them.)
            base, per = arr.ctypes.data, arr.strides[0]
            for (lo, hi), ev in zip(self._slabs, done):
                a, b = base + lo * per, base + min(hi, arr.shape[0]) * per
                if b > a and _ffi.host_register_range(a, b - a):
                    with self._lock:
                        self._regs.append((a, b - a))
                ev.set()
        finally:
            for ev in done:
                ev.set()

    def input_ready(self, i):
        self._in_done[i].wait()

    def result_ready(self, i):
        self._out_done[i].wait()

    def release(self):
        for t in self._threads:
            t.join()
        with self._lock:
            regs, self._regs = self._regs, []
        for addr, nbytes in regs:   # best effort: this runs in a `finally` -- one refused release must neither hide the caller's exception
            try:                    # nor keep the remaining slabs page-locked
                _ffi.host_unregister_range(addr)
            except _ffi.P3DError as exc:
                import warnings
                warnings.warn(f'could not release a page-locked slab of {nbytes} bytes: {exc}', RuntimeWarning, stacklevel=2)


_CHUNK_WORKERS = int(os.environ.get('P3D_CHUNK_WORKERS', 4))   # chunks in flight (slots 0.. of the plan cache; slot 15 belongs to the unchunked path)


class _FFTWorker:
    """One plan + device buffers for a chunk of slices; four of them alternate so that the PCIe transfers (and, for float64 /
    complex128 cubes, the dtype conversion through page-locked staging buffers) of one chunk run while another chunk iterates
    (ctypes and NumPy's copy loops release the GIL; each plan has its own non-blocking stream).  Workers are cached like the plans
    (per slice shape, device and slot): a per-slice caller (``POCS_algorithm`` under ``xr.apply_ufunc``) does not pay for
    allocations on every call."""

    def __init__(self, nil, nxl, step, device, slot):
        self.plan = _get_plan(nil, nxl, step, device, slot)
        self.capacity = self.plan.max_slices
        per = nil * nxl * 8
        self.x = self.plan.alloc(per * self.capacity)
        self.o = self.plan.alloc(per * self.capacity)
        self.m = self.plan.alloc(nil * nxl * 4)
        self.hx = self.ho = None     # page-locked staging, allocated when a cube needs a dtype conversion

    @classmethod
    def get(cls, nil, nxl, step, device, slot, maskf):
        key = (nil, nxl, device, slot)
        w = _workers.get(key)
        if w is None or w.capacity < step or w.plan.handle is None:
            if w is not None:
                w.close()
            w = _workers[key] = cls(nil, nxl, step, device, slot)
        w.m.upload(maskf)
        return w

    def close(self):
        for b in (self.x, self.o, self.m):   # p3d_malloc'ed: they outlive a closed plan, p3d_free takes a NULL plan then
            b.free()
        for b in (self.hx, self.ho):
            if b is not None:
                b.free()

    def run(self, chunk, dst, sched, niter, thresh_op, version, eps, alpha, before_upload=None, before_download=None):
        n = chunk.shape[0]
        t0 = time.perf_counter()
        marks = [('start', t0)] if _timeline is not None else None   # tools/e2e_timeline.py: where a chunk's wall time goes

        def mark(name):
            if marks is not None:
                marks.append((name, time.perf_counter()))
        if n > self.capacity or chunk.shape[1:] != (self.plan.nil, self.plan.nxl):
            raise ValueError(f'chunk {chunk.shape} does not fit the worker ({self.capacity}, {self.plan.nil}, {self.plan.nxl})')
        dt, dtype = (_ffi.P3D_C64, np.complex64) if np.iscomplexobj(chunk) else (_ffi.P3D_F32, np.float32)
        # cubes that already have the device dtype move straight between the caller's arrays and the device (the runtime reaches
        # the PCIe rate from pageable memory here); others are converted through page-locked staging buffers in parallel slabs
        direct = chunk.dtype == dtype and chunk.flags.c_contiguous and dst.dtype == dtype and dst.flags.c_contiguous
        if before_upload is not None:
            before_upload()                            # (the caller's cube is page-locked, or will not be)
        if direct:
            self.x.upload(chunk)
        else:
            if self.hx is None:
                per = self.plan.nil * self.plan.nxl * 8 * self.capacity
                self.hx, self.ho = _ffi.PinnedBuffer(per), _ffi.PinnedBuffer(per)
            xin = self.hx.view(chunk.shape, dtype)
            _slab_copy(xin, chunk)
            self.x.upload(xin)                         # one upload serves the statistics and the loop
        mark('h2d')
        stats = self.plan.prime_dev(self.x.ptr, dt, self.m.ptr, n)   # the statistics pass doubles as the first pass of the job
        mark('prime')
        active = ~(stats[:, 2] == 0)                   # max|fft2(x)| == 0 <=> np.count_nonzero(x) == 0 -> untouched (POCS.py:515-521)
        stats[~active] = 1.0                           # keep NaNs of empty slices out of the (unused) schedule rows
        tau = sched(stats)
        mark('schedule')
        done, sums, _ = self.plan.run_dev(self.x.ptr, dt, self.m.ptr, tau, niter, self.o.ptr, n, thresh_op=thresh_op,
                                          version=version, eps=eps, alpha=alpha, active=active, primed=True)
        mark('loop')
        if before_download is not None:
            before_download()                          # (the result array's pages are there and page-locked)
        if direct:
            self.o.download_into(dst)
        else:
            xout = self.ho.view(chunk.shape, dtype)
            self.o.download_into(xout)
            _slab_copy(dst, xout)                      # e.g. a float64 cube: results are cast on assignment, as before
        mark('d2h')
        if marks is not None:
            _timeline.append((id(self), marks))
        return done, sums, time.perf_counter() - t0


def _result_rows(done, sums, runtime):
    n = done.shape[0]
    with np.errstate(invalid='ignore', divide='ignore'):
        costs = ((sums[1:] - sums[:-1]) / sums[1:]) ** 2  # POCS.py:622
    rows = []
    for s in range(n):
        k = int(done[s])
        rows.append({
            'niterations': k,
            'runtime': round(runtime / n, 3) if k else 0,
            'cost': float(costs[k - 1, s]) if k else 0,
            'costs': [float(c) for c in costs[:k, s]] if k else [0],
        })
    return rows


def _get_plan64(nil, nxl, nslices, device):
    key = ('f64', nil, nxl, device)
    plan = _plans.get(key)
    if plan is None or plan.max_slices < nslices:
        if plan is not None:
            plan.close()
        plan = _ffi.Plan64(nil, nxl, max(nslices, 1), device)
        _plans[key] = plan
    return plan


def _pocs_cube_double(cube, mask, out, niter, thresh_op, thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, version, results, device,
                      batch_slices):
    """``pocs_cube`` through the double-precision loop (``_ffi.Plan64``): statistics of the double-precision ``fft2`` -> the schedule
    (host, complex128, as always) -> the iterations, batch by batch.  The data-driven schedule takes its picks from the float32
    plan's device sort (positions in the sorted spectrum; the values differ from a double sort's by float32 rounding)."""
    nslices, nil, nxl = cube.shape
    step = int(batch_slices) if batch_slices else max(1, min(nslices, (2 << 30) // (nil * nxl * 16)))   # work + staging: 48 B per point and slice
    step = min(step, 65535)   # (plan64's limit: cubes of tiny slices)
    plan = _get_plan64(nil, nxl, min(step, nslices), device)
    mask64 = np.ascontiguousarray(mask, dtype=np.float64)
    # one upload per batch into device buffers, statistics and loop on the resident copy, the result downloaded straight into `out` (whose fresh
    # pages a few threads touch meanwhile): a 2-GiB batch of complex128 slices used to be uploaded twice, downloaded into an array of the
    # wrapper's own and copied once more on the host -- more wall time than its loop
    cap = min(step, nslices)
    itemsize = cube.dtype.itemsize if cube.dtype in _ffi.Plan64._DT else (16 if np.iscomplexobj(cube) else 8)
    xd, od, md = _batch_buffers(device, nil * nxl * itemsize * cap, mask64.nbytes)
    touching = [] if (np.may_share_memory(cube, out) or not out.flags.c_contiguous) else _touch_pages(out)
    try:
        md.upload(mask64)
        for lo in range(0, nslices, step):
            chunk = cube[lo:lo + step]
            n = chunk.shape[0]
            t0 = time.perf_counter()
            xc, dt = plan._cube(chunk)
            xd.upload(xc)
            active = _active_slices(chunk)   # np.count_nonzero(x) == 0 -> untouched (POCS.py:515-521)
            if thresh_model == 'data-driven':
                narrow = chunk.astype(np.complex64 if np.iscomplexobj(chunk) else np.float32)
                tau = _data_driven_batch(_get_plan(nil, nxl, n, device, slot=15), narrow, active, niter, p_max, p_min)
            else:
                stats = plan.stats_dev(xd.ptr, dt, n)
                stats[~active] = 1.0
                tau = _schedule_from_stats(stats, nil * nxl, thresh_model, niter, p_max, p_min, decay_kind)
            if sqrt_decay:
                tau = np.sqrt(tau)
            done, sums, _ = plan.run_dev(xd.ptr, dt, md.ptr, tau, niter, od.ptr, n, thresh_op=thresh_op, version=version, eps=eps, alpha=alpha, active=active)
            for f in touching:
                f.result()
            touching = []
            dst = out[lo:lo + n]
            if dst.dtype == xc.dtype and dst.flags.c_contiguous:
                od.download_into(dst)
            else:
                dst[...] = od.download(xc.shape, xc.dtype)
            runtime = time.perf_counter() - t0
            if results is not None:
                results.extend(_result_rows(done, sums, runtime))
    finally:
        for f in touching:
            f.result()
    return out


def _pocs_cube_wavelet_double(cube, mask, out, wavelet, niter, thresh_op, thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, version, results, device,
                              batch_slices):
    """``pocs_cube`` for the WAVELET transform through the double-precision loop (``_ffi.WaveletPlan64``): statistics of the double-precision
    decomposition -> the schedule (host, as always) -> the iterations, batch by batch; complex64 / float32 cubes are widened on load and the
    result is cast back on store (the reference's final cast, cube_POCS_interpolation_3D.py:324)."""
    nslices, nil, nxl = cube.shape
    # coefficient vector, feed, row / column intermediates, approximations and reconstructions of every level, staging: ~7.5 slice-sized arrays of
    # 16-byte elements
    step = int(batch_slices) if batch_slices else max(1, min(nslices, (4 << 30) // (nil * nxl * 120)))
    step = min(step, 65535, nslices)
    plan = _get_wavelet_plan64(nil, nxl, step, wavelet, device)
    mask64 = np.ascontiguousarray(mask, dtype=np.float64)
    for lo in range(0, nslices, step):
        chunk = cube[lo:lo + step]
        n = chunk.shape[0]
        t0 = time.perf_counter()
        xc, dt = plan._cube(chunk)
        active = _active_slices(chunk)   # np.count_nonzero(x) == 0 -> untouched (POCS.py:515-521)
        stats = plan.stats_dev(xc.ctypes.data, dt, n)
        stats[~active] = 1.0       # keep NaNs of empty slices out of the (unused) schedule rows ...
        stats[~active, ..., 1] = 0.0   # ... without making them complex
        tau = _wavelet_schedule_from_stats(stats, thresh_model, niter, p_max, p_min, decay_kind)
        if sqrt_decay:
            tau = np.sqrt(tau)  # POCS.py:595
        dst = out[lo:lo + n]
        direct = dst.dtype == xc.dtype and dst.flags.c_contiguous
        res = dst if direct else np.empty_like(xc)
        done, sums, _ = plan.run_dev(xc.ctypes.data, dt, mask64.ctypes.data, tau, niter, res.ctypes.data, n, thresh_op=thresh_op, version=version, eps=eps,
                                     alpha=alpha, active=active)
        if not direct:
            dst[...] = res
        runtime = time.perf_counter() - t0
        if results is not None:
            results.extend(_result_rows(done, sums, runtime))
    return out


def _pocs_cube_shearlet_double(cube, mask, out, psi, niter, thresh_op, thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, version, results, device,
                               batch_slices):
    """``pocs_cube`` for the SHEARLET transform through the double-precision loop (``_ffi.ShearletPlan64``), batch by batch as
    ``_pocs_cube_wavelet_double``: statistics of the double-precision coefficients -> the schedule (host) -> the iterations; complex64 / float32
    cubes are widened on load and the result is cast back on store (the reference's final cast, cube_POCS_interpolation_3D.py:324)."""
    nslices, nil, nxl = cube.shape
    nsh = psi.shape[2]
    # coefficients of one slice are nsh full-size complex128 arrays on the device: bound the batch by memory (<= 16 GiB of coefficients)
    fit = max(1, min(int((16 << 30) // (nsh * nil * nxl * 16)), 65535 // nsh))
    step = min(int(batch_slices) if batch_slices else nslices, fit, nslices)
    plan = _get_shearlet_plan64(psi, step, device)
    mask64 = np.ascontiguousarray(mask, dtype=np.float64)
    for lo in range(0, nslices, step):
        chunk = cube[lo:lo + step]
        n = chunk.shape[0]
        t0 = time.perf_counter()
        xc, dt = plan._cube(chunk)
        active = _active_slices(chunk)   # np.count_nonzero(x) == 0 -> untouched (POCS.py:515-521)
        stats = plan.stats_dev(xc.ctypes.data, dt, n)
        stats[~active] = 1.0
        stats[~active, ..., 1] = 0.0
        tau = _shearlet_schedule_from_stats(stats, (nil, nxl), thresh_model, niter, p_max, p_min, decay_kind)
        if sqrt_decay:
            tau = np.sqrt(tau)  # POCS.py:595
        dst = out[lo:lo + n]
        direct = dst.dtype == xc.dtype and dst.flags.c_contiguous
        res = dst if direct else np.empty_like(xc)
        done, sums, _ = plan.run_dev(xc.ctypes.data, dt, mask64.ctypes.data, tau, niter, res.ctypes.data, n, thresh_op=thresh_op, version=version, eps=eps,
                                     alpha=alpha, active=active)
        if not direct:
            dst[...] = res
        runtime = time.perf_counter() - t0
        if results is not None:
            results.extend(_result_rows(done, sums, runtime))
    return out


def _check_cube_args(cube, mask, transform_kind, thresh_op, version, niter, eps, p_max, alpha, p_min):
    """Argument checks of the batched entry points (``pocs_cube``, ``sharding.pocs_block_on_device``), made BEFORE anything is
    uploaded: the kernels read ``nil * nxl`` mask entries whatever the caller handed over.  Returns the normalised
    ``(cube, mask, kind, niter, eps, p_max, alpha, p_min)`` (POCS.py:488-508 for the scalar conversions)."""
    cube = np.asarray(cube)
    if cube.ndim != 3:
        raise ValueError(f'cube must be (nslices, iline, xline), got shape {cube.shape}')
    mask = np.asarray(mask)
    if mask.shape != cube.shape[1:]:
        raise ValueError(f'mask shape {mask.shape} does not match slice shape {cube.shape[1:]}')
    kind = _check_common(mask, transform_kind, thresh_op)
    if version not in _ffi.P3D_VER:
        raise ValueError(f'Unknown POCS version {version!r}')
    niter, eps, p_max, alpha = int(niter), float(eps), float(p_max), float(alpha)
    if isinstance(p_min, str) and p_min != 'adaptive':
        p_min = float(p_min)  # YAML 1.1 reads 1e-4 as a string
    if niter < 1 and cube.shape[0] > 0:  # the reference's loop body never runs and it then fails on `iiter`; be explicit
        raise ValueError('niter must be >= 1')
    return cube, mask, kind, niter, eps, p_max, alpha, p_min


def pocs_cube(
    cube,
    mask,
    transform_kind='FFT',
    niter=50,
    thresh_op='hard',
    thresh_model='exponential',
    eps=1e-9,
    alpha=1.0,
    p_max=0.99,
    p_min=1e-5,
    sqrt_decay=False,
    decay_kind='values',
    version='regular',
    results=None,
    device=0,
    batch_slices=None,
    wavelet=None,
    auxiliary_data=None,
    precision=None,
    out=None,
    **ignored,
):
    """
    Interpolate every ``(iline, xline)`` slice of ``cube`` (shape ``(nslices, nil, nxl)``, slice-major
    like the netCDF cubes of the workflow) with the POCS iteration of ``POCS_algorithm`` -- all slices
    of a batch advance together on the GPU.

    Parameters are those of :func:`POCS_algorithm`; ``mask`` (``(nil, nxl)``) is shared by all slices
    (cube_POCS_interpolation_3D.py:242-244).  ``results`` (list, optional) receives one dict per
    slice with ``niterations``, ``runtime``, ``cost`` and ``costs``.  ``transform`` / ``itransform``
    callables in ``**ignored`` are accepted for signature compatibility and not called: the transform
    is selected by ``transform_kind``.

    ``precision`` (every transform; default: the environment variable ``P3D_PRECISION``, else ``None``) -- the arithmetic of the loop:
    ``None``: that of the cube -- float32 kernels for complex64 / float32 cubes, the double-precision loop for complex128 / float64
    cubes (the reference computes such cubes in double precision; POCS.py:371-656 never narrows its input); ``'reference'``: double
    precision also for complex64 / float32 cubes, the result cast back (what the reference itself executes for the soft / garrote
    operators, FPOCS and APOCS, and for every run under NumPy < 2 -- SURVEY appendix A.16); ``'float32'``: the float32 kernels whatever the
    cube (double cubes are converted on the way in, the result widened on the way out).  One exception to ``None``: SHEARLET on slice extents that are not
    powers of two but have plans on the double-precision register engine (7-smooth extents 96 ... 4096) runs the double-precision loop for single-precision
    cubes too -- it is the faster loop there (fused passes) as well as the reference's arithmetic.  The double-precision loops are
    precision paths (FFT: about a quarter of the float32 rate; WAVELET: per-axis kernels without LDS tiles; SHEARLET: three fused
    passes per iteration where both extents have a plan on the double-precision register engine, a tenth of the float32 rate), have the hard / soft /
    garrote operators, the FFT and SHEARLET ones slice extents up to 5120;
    a call that asks for (or implies) double precision outside that coverage runs the float32 kernels and says so with a ``RuntimeWarning``.

    ``out`` (optional): an array of the shape and dtype of ``cube`` to write the result into (e.g. a slab of the merged cube of the
    step-13 driver) instead of a new one.

    Returns an array with the shape and dtype of ``cube`` (``out`` when given).
    """
    cube, mask, kind, niter, eps, p_max, alpha, p_min = _check_cube_args(cube, mask, transform_kind, thresh_op, version, niter, eps,
                                                                         p_max, alpha, p_min)
    if out is None:
        out = np.empty_like(cube)
    elif not isinstance(out, np.ndarray) or out.shape != cube.shape or out.dtype != cube.dtype:
        raise ValueError(f'out must be a NumPy array of shape {cube.shape} and dtype {cube.dtype}')
    if cube.shape[0] == 0:
        return out
    # Plans, chunk workers and the batch buffers are cached per device and are not re-entrant (include/p3d.h: "a plan is bound to one
    # device + stream"): calls on ONE device take turns -- a threaded caller (dask's threaded scheduler under xr.apply_ufunc) gets the same
    # results as a serial one; calls on different devices run side by side.
    with _device_lock(device):
        return _pocs_cube_locked(cube, mask, kind, niter, thresh_op, thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, version,
                                 results, device, batch_slices, wavelet, auxiliary_data, precision, out, ignored)


def _double_loop_wanted(dtype, nil, nxl, kind, thresh_op, precision, wavelet=None, transform=None, auxiliary_data=None):
    """``(precision, want_double)``: the normalised ``precision`` argument (``P3D_PRECISION`` when it is None) and whether the call asks for one of
    the double-precision loops -- explicitly (``'reference'``), by the cube's dtype (complex128 / float64), or because the double-precision loop is the
    only / the faster one for the job: WAVELET banks of more than 64 taps (the float32 tile kernels hold db1 ... db32), SHEARLET on extents that are
    not powers of two but have plans on the double-precision register engine (fused passes there against the float32 loop's unfused ones).  One rule
    for ``pocs_cube`` and for ``sharding.pocs_block_on_device``."""
    if precision is None:
        precision = os.environ.get('P3D_PRECISION') or None
    if precision not in (None, 'reference', 'float32'):
        raise ValueError(f"precision must be None, 'reference' or 'float32', got {precision!r}")
    wide = np.dtype(dtype) in (np.dtype(np.complex128), np.dtype(np.float64))
    want_double = precision == 'reference' or (wide and precision is None)
    if precision is None and not wide and thresh_op in _WAVELET_OPS:
        if kind == 'WAVELET':
            bank = _wavelet_name(transform, wavelet)
            taps = len(bank[0]) if isinstance(bank, (tuple, list)) else len(_ffi.wavelet_filters(bank)[0])
            if taps > 64:   # the longer banks (db33-38, coif11-17 ...) run the double-precision loop
                want_double = True
        elif kind == 'SHEARLET' and auxiliary_data is not None and (nil & (nil - 1) or nxl & (nxl - 1)) and _ffi.shearlet64_fused_shape(nil, nxl):
            # Extents that are not powers of two: the float32 loop runs unfused passes there, the double-precision loop its three fused ones -- faster
            # (1000 x 1000: 1.0 against 1.6 ms per slice-iteration, tools/shearlet_probe.py) AND the reference's own arithmetic (np.fft inside FFST computes
            # in double whatever the cube's dtype): single-precision cubes take it by default on such grids; precision='float32' keeps the float32 kernels
            want_double = True
    return precision, want_double


def _pocs_cube_locked(cube, mask, kind, niter, thresh_op, thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, version, results, device,
                      batch_slices, wavelet, auxiliary_data, precision, out, ignored):
    """``pocs_cube`` behind its argument checks, with the device's lock held."""
    nslices, nil, nxl = cube.shape
    step = int(batch_slices) if batch_slices else nslices
    maskf = np.ascontiguousarray(mask, dtype=np.float32)
    precision, want_double = _double_loop_wanted(cube.dtype, nil, nxl, kind, thresh_op, precision, wavelet, ignored.get('transform'), auxiliary_data)
    if want_double and kind == 'WAVELET' and thresh_op in _WAVELET_OPS:
        if decay_kind == 'factors' and not all(s in thresh_model for s in ['inverse', 'proportional']):
            raise IndexError('list index out of range (decay_kind="factors" yields one tau per iteration, the WAVELET '
                             'thresholding needs one per level and detail)')   # (as the float32 path below: the reference fails here)
        return _pocs_cube_wavelet_double(cube, mask, out, _wavelet_name(ignored.get('transform'), wavelet), niter, thresh_op, thresh_model, eps, alpha, p_max,
                                         p_min, sqrt_decay, decay_kind, version, results, device, batch_slices)
    if want_double and kind == 'SHEARLET' and thresh_op in _WAVELET_OPS and auxiliary_data is not None and max(nil, nxl) <= 5120:
        psi = np.asarray(auxiliary_data)
        if psi.ndim != 3 or psi.shape[:2] != (nil, nxl):
            raise ValueError(f'Psi must be ({nil}, {nxl}, nshearlets), got shape {psi.shape}')
        return _pocs_cube_shearlet_double(cube, mask, out, psi, niter, thresh_op, thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, version,
                                          results, device, batch_slices)
    if want_double and not (kind == 'FFT' and thresh_op in _WAVELET_OPS and max(nil, nxl) <= 5120):
        import warnings
        why = (f'the {kind} transform' if kind != 'FFT' else f'thresh_op={thresh_op!r}' if thresh_op not in _WAVELET_OPS else f'slice extents above 5120 ({nil} x {nxl})')
        warnings.warn(f'double-precision arithmetic was {"requested" if precision == "reference" else "implied by the " + str(cube.dtype) + " cube"}, but the double-precision '
                      f'loop does not cover {why}: this call runs the float32 kernels (pass precision="float32" to say so explicitly)', RuntimeWarning, stacklevel=3)
    if want_double and kind == 'FFT' and thresh_op in _WAVELET_OPS and max(nil, nxl) <= 5120:
        return _pocs_cube_double(cube, mask, out, niter, thresh_op, thresh_model, eps, alpha, p_max, p_min, sqrt_decay, decay_kind, version, results,
                                 device, batch_slices)
    if kind == 'SHEARLET':
        if auxiliary_data is None:
            raise ValueError(f'{kind} requires pre-computed shearlets in Fourier domain (Psi)')
        if thresh_op not in _WAVELET_OPS:
            raise NotImplementedError(f'thresh_op {thresh_op!r} is not available for the SHEARLET transform')
        psi = np.asarray(auxiliary_data)
        if psi.ndim != 3 or psi.shape[:2] != (nil, nxl):
            raise ValueError(f'Psi must be ({nil}, {nxl}, nshearlets), got shape {psi.shape}')
        # coefficients of one slice are nsh full-size arrays on the device: bound the batch by memory (<= 8 GiB of coefficients)
        fit = max(1, min(int((8 << 30) // (psi.shape[2] * nil * nxl * 8)), 65535 // psi.shape[2]))
        step = min(step, fit)
        plan = _get_shearlet_plan(psi, min(step, nslices), device)
    elif kind == 'WAVELET':
        if thresh_op not in _WAVELET_OPS:  # threshold_wavelet (POCS.py:105-166) has no percentile variants
            raise NotImplementedError(f'thresh_op {thresh_op!r} is not available for the WAVELET transform')
        if decay_kind == 'factors' and not all(s in thresh_model for s in ['inverse', 'proportional']):
            # the reference builds a (niter, 1, 1) schedule here and threshold_wavelet (POCS.py:135-166) then indexes it per
            # level and detail: it fails for every decomposition; keep the failure instead of inventing a behaviour
            raise IndexError('list index out of range (decay_kind="factors" yields one tau per iteration, the WAVELET '
                             'thresholding needs one per level and detail)')
        plan = _get_wavelet_plan(nil, nxl, min(step, nslices), _wavelet_name(ignored.get('transform'), wavelet), device)
    elif thresh_model != 'data-driven' and out.flags.c_contiguous:
        # FFT, statistics-driven schedules: chunks go through device buffers, uploaded once each, two chunks in flight
        slice_bytes = nil * nxl * (8 if np.iscomplexobj(cube) else 4)
        if not batch_slices and nslices * slice_bytes >= (1 << 30):
            step = max(1, (int(os.environ.get('P3D_CHUNK_MIB', 128)) << 20) // slice_bytes)   # ~128 MiB per chunk, four in flight
        step = min(step, nslices)
        starts = list(range(0, nslices, step))
        nworkers = min(_CHUNK_WORKERS, len(starts))
        workers = [None] * nworkers      # (each lane creates / fetches its own worker: plans and device buffers of a first call are set up side by side)

        def sched(stats):
            tau = _schedule_from_stats(stats, nil * nxl, thresh_model, niter, p_max, p_min, decay_kind)
            return np.sqrt(tau) if sqrt_decay else tau  # POCS.py:595

        # Large cubes: page-lock the caller's cube and the result IN PLACE for the duration of the call, so that the chunk transfers are
        # DMA on the workers' own streams, uploads and downloads at the same time (profiles/r04_pcie_probe.txt: 96 GB/s both ways
        # together against 56 GB/s for pageable memory, whose copies take turns; a download into FRESH pages runs at 17 GB/s).  The
        # result array is brand new: its pages are touched by a few threads first (24 ms for 4 GiB; the driver alone takes 176 ms),
        # while the first chunks are already uploading.  Registration runs in the background, slab by slab ahead of the workers (_HostPins).
        pin = _HostPins(cube, out, starts, step) if (len(starts) > 1 and cube.nbytes >= _PIN_MIN_BYTES and cube.dtype == out.dtype
                                       and cube.dtype in (np.complex64, np.float32) and cube.flags.c_contiguous) else None

        def lane(w):
            t_lane = time.perf_counter()
            workers[w] = _FFTWorker.get(nil, nxl, step, device, w, maskf)
            if _timeline is not None:
                _timeline.append(('setup', [('lane starts', t_lane), ('worker ready', time.perf_counter())]))
            rows = []
            for i in range(w, len(starts), nworkers):
                lo = starts[i]
                rows.append((lo, workers[w].run(cube[lo:lo + step], out[lo:lo + step], sched, niter, thresh_op, version, eps, alpha,
                                                before_upload=None if pin is None else (lambda i=i: pin.input_ready(i)),
                                                before_download=None if pin is None else (lambda i=i: pin.result_ready(i)))))
            return rows

        try:
            if nworkers == 1:
                done_rows = lane(0)
            else:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(nworkers) as pool:
                    done_rows = [r for part in pool.map(lane, range(nworkers)) for r in part]
        finally:
            if pin is not None:
                pin.release()
        if results is not None:
            for _, (done, sums, runtime) in sorted(done_rows, key=lambda r: r[0]):
                results.extend(_result_rows(done, sums, runtime))
        return out
    else:
        plan = _get_plan(nil, nxl, min(step, nslices), device, slot=15)   # the low slots belong to the chunk workers

    if kind in ('WAVELET', 'SHEARLET'):
        # One upload per batch, statistics and loop on the device-resident copy, the result downloaded straight into `out` (whose fresh
        # pages a few threads touch while the batch iterates: a download into untouched pages runs at a third of the link's rate).
        # configs[3]'s cube: 0.10 -> 0.04 s per call (tools/wv_e2e.py); the entry points took host pointers before: two uploads,
        # a result array of their own and a copy of it.
        narrow = np.complex64 if np.iscomplexobj(cube) else np.float32
        per = nil * nxl * np.dtype(narrow).itemsize
        cap = min(step, nslices)
        xd, od, md = _batch_buffers(device, per * cap, maskf.nbytes)
        touching = [] if (np.may_share_memory(cube, out) or not out.flags.c_contiguous) else _touch_pages(out)
        try:
            md.upload(maskf)
            for lo in range(0, nslices, step):
                chunk = cube[lo:lo + step]
                n = chunk.shape[0]
                t0 = time.perf_counter()
                xc, dt = plan._cube(chunk)
                xd.upload(xc)
                active = _active_slices(chunk)  # np.count_nonzero(x) == 0 -> untouched (POCS.py:515-521)
                stats = plan.stats_dev(xd.ptr, dt, n)
                stats[~active] = 1.0      # keep NaNs of empty slices out of the (unused) schedule rows ...
                stats[~active, ..., 1] = 0.0  # ... without making them complex
                if kind == 'WAVELET':
                    tau = _wavelet_schedule_from_stats(stats, thresh_model, niter, p_max, p_min, decay_kind)
                else:
                    tau = _shearlet_schedule_from_stats(stats, (nil, nxl), thresh_model, niter, p_max, p_min, decay_kind)
                if sqrt_decay:
                    tau = np.sqrt(tau)  # POCS.py:595
                done, sums, _ = plan.run_dev(xd.ptr, dt, md.ptr, tau, niter, od.ptr, n, thresh_op=thresh_op, version=version, eps=eps, alpha=alpha,
                                             active=active)
                for f in touching:
                    f.result()
                touching = []
                dst = out[lo:lo + n]
                if dst.dtype == xc.dtype and dst.flags.c_contiguous:
                    od.download_into(dst)
                else:
                    dst[...] = od.download(xc.shape, xc.dtype)   # (a double-precision cube: the reference's final cast, the other way round)
                runtime = time.perf_counter() - t0
                if results is not None:
                    results.extend(_result_rows(done, sums, runtime))
        finally:
            for f in touching:
                f.result()
        return out

    # FFT, the schedules that need the sorted spectrum (or a result array that is not contiguous): one batch at a time, resident between one
    # upload and one download like the WAVELET / SHEARLET batches above
    narrow = np.complex64 if np.iscomplexobj(cube) else np.float32
    xd, od, md = _batch_buffers(device, nil * nxl * np.dtype(narrow).itemsize * min(step, nslices), maskf.nbytes)
    md.upload(maskf)
    for lo in range(0, nslices, step):
        chunk = cube[lo:lo + step]
        n = chunk.shape[0]
        t0 = time.perf_counter()
        xc, dt = plan._cube(chunk)
        xd.upload(xc)
        active = _active_slices(chunk)  # np.count_nonzero(x) == 0 -> untouched (POCS.py:515-521)
        if thresh_model == 'data-driven':
            tau = _data_driven_batch(plan, chunk, active, niter, p_max, p_min, x_dev=xd.ptr if dt == _ffi.P3D_C64 else None)
        else:
            stats = plan.stats_dev(xd.ptr, dt, n)
            stats[~active] = 1.0  # keep NaNs of empty slices out of the (unused) schedule rows
            tau = _schedule_from_stats(stats, nil * nxl, thresh_model, niter, p_max, p_min, decay_kind)
        if sqrt_decay:
            tau = np.sqrt(tau)  # POCS.py:595
        done, sums, _ = plan.run_dev(xd.ptr, dt, md.ptr, tau, niter, od.ptr, n, thresh_op=thresh_op, version=version, eps=eps, alpha=alpha, active=active)
        dst = out[lo:lo + n]
        if dst.dtype == xc.dtype and dst.flags.c_contiguous:
            od.download_into(dst)
        else:
            dst[...] = od.download(xc.shape, xc.dtype)
        runtime = time.perf_counter() - t0
        if results is not None:
            results.extend(_result_rows(done, sums, runtime))
    return out


# =================================================================================================
#                                      per-slice contract
# =================================================================================================
_FFT_MODULES = ('numpy.fft', 'scipy.fft', 'scipy.fftpack', 'pyfftw.interfaces', 'mkl_fft')
_FFT_BENIGN = {'s': (None,), 'axes': ((-2, -1), (0, 1), [-2, -1], [0, 1], None), 'norm': (None, 'backward')}


def _unwrap_partial(func):
    """(innermost callable, merged keywords) of nested functools.partial objects; keywords None when positional arguments are bound."""
    kw = {}
    while isinstance(func, partial):
        if func.args:
            return func.func, None
        kw = {**(func.keywords or {}), **kw}
        func = func.func
    return func, kw


def _check_transform_callables(kind, transform, itransform):
    """The reference CALLS ``transform`` / ``itransform`` (POCS.py:535, 592, 613); here the HIP kernels are the transform, chosen by
    ``transform_kind``.  A callable that is not the one the kind names would be ignored silently -- refuse it instead: FFT wants
    ``fft2`` / ``ifft2`` of numpy.fft (or a drop-in: scipy.fft, pyfftw.interfaces, mkl_fft) over the last two axes, WAVELET
    ``partial(pywt.wavedec2, ...)`` / ``partial(pywt.waverec2, ...)``, SHEARLET FFST's ``shearletTransformSpect`` /
    ``inverseShearletTransformSpect`` (cube_POCS_interpolation_3D.py:255-274)."""
    want = {'FFT': ('fft2', 'ifft2'), 'WAVELET': ('wavedec2', 'waverec2'),
            'SHEARLET': ('shearletTransformSpect', 'inverseShearletTransformSpect')}[kind]
    for func, name, role in ((transform, want[0], 'transform'), (itransform, want[1], 'itransform')):
        inner, kw = _unwrap_partial(func)
        fname = getattr(inner, '__name__', type(inner).__name__)
        ok = kw is not None
        if kind == 'FFT':
            ok = ok and fname == name and str(getattr(inner, '__module__', '')).startswith(_FFT_MODULES)
            ok = ok and all(k in _FFT_BENIGN and v in _FFT_BENIGN[k] for k, v in (kw or {}).items())
        else:   # third-party functions (pywt / FFST need not be installed here): recognised by name
            ok = ok and fname.endswith(name)
        if not ok:
            raise NotImplementedError(
                f'{role}={func!r} is not the {name} that transform_kind={kind!r} names: the HIP kernels execute the transform '
                f'themselves and cannot run an arbitrary callable (no CPU fallback)')

def POCS_algorithm(
    x,
    mask,
    auxiliary_data=None,
    transform=None,
    itransform=None,
    transform_kind: str = None,
    niter: int = 50,
    thresh_op: str = 'hard',
    thresh_model: str = 'exponential',
    eps: float = 1e-9,
    alpha: int = 1.0,
    p_max: float = 0.99,
    p_min: float = 1e-5,
    sqrt_decay: str = False,
    decay_kind: str = 'values',
    verbose: bool = False,
    version: str = 'regular',
    results_dict: dict = None,
    path_results: str = None,
):
    """
    Interpolate sparse input grid using Point Onto Convex Sets (POCS) algorithm -- drop-in for the
    reference's ``POCS_algorithm`` (POCS.py:371-656) with the loop executed by the HIP kernels.

    Parameters, error behaviour (``ValueError`` for a non-boolean mask, missing transforms, unknown
    transform kind, shearlet without Psi; POCS.py:488-503), side channels (``results_dict`` keys
    ``niterations`` / ``runtime`` / ``cost``; one ``niter;runtime;cost_1;..`` line appended to
    ``path_results``; POCS.py:644-651) and return value (complex in -> complex out, real in -> real
    part; POCS.py:653-656) follow the reference.  Differences: ``transform`` / ``itransform`` must be
    supplied and must be the functions ``transform_kind`` names (``np.fft.fft2`` / ``ifft2`` for ``'FFT'``; partials of
    ``pywt.wavedec2`` / ``waverec2`` for ``'WAVELET'``, the wavelet name read from ``transform.keywords['wavelet']`` as set up by
    the step-13 driver; FFST's ``shearletTransformSpect`` pair for ``'SHEARLET'`` with the spectra in ``auxiliary_data``) -- they
    are not called, the HIP kernels are the transform, and any other callable raises ``NotImplementedError`` instead of being
    ignored; arithmetic is float32 on the GPU.
    """
    if np.max(mask) > 1:
        raise ValueError(f'mask should be quasi-boolean (0 or 1) but has maximum of {np.max(mask)}')
    if any(v is None for v in [transform, itransform]):
        raise ValueError('Forward and inverse transform function have to be supplied')
    if transform_kind is None or transform_kind.upper() not in TRANSFORMS:
        raise ValueError(f'Unsupported transform. Please select one of: {TRANSFORMS}')
    transform_kind = transform_kind.upper()
    if transform_kind == 'SHEARLET' and auxiliary_data is None:
        raise ValueError(f'{transform_kind} requires pre-computed shearlets in Fourier domain (Psi)')

    if transform_kind in _HIP_TRANSFORMS:
        _check_transform_callables(transform_kind, transform, itransform)

    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError(f'x must be a 2D slice, got shape {x.shape}')
    results = []
    out = pocs_cube(
        x[None], mask, transform_kind=transform_kind, niter=niter, thresh_op=thresh_op,
        thresh_model=thresh_model, eps=eps, alpha=alpha, p_max=p_max, p_min=p_min, sqrt_decay=sqrt_decay,
        decay_kind=decay_kind, version=version, results=results, transform=transform, itransform=itransform,
        auxiliary_data=auxiliary_data,
    )[0]
    info = results[0]

    if verbose:
        print('\n' + '-' * 20)
        print(f'# iterations:  {info["niterations"]:4d}')
        print(f'cost function: {info["cost"]}')
        print(f'runtime:       {info["runtime"]:.3f} s')
        print('-' * 20)

    if isinstance(results_dict, dict):
        results_dict['niterations'] = info['niterations']
        results_dict['runtime'] = info['runtime']
        results_dict['cost'] = info['cost']

    if path_results is not None:
        with open(path_results, mode="a", newline='\n') as f:
            f.write(';'.join([str(i) for i in [info['niterations'], info['runtime']] + info['costs']]) + '\n')

    return out


POCS = partial(POCS_algorithm, version='regular')
FPOCS = partial(POCS_algorithm, version='fast')
APOCS = partial(POCS_algorithm, version='adaptive')
