"""
Steps 12 -> 13 -> 14 of the workflow on one GPU without leaving HBM in between.

The reference writes a netCDF cube after each step (cube_apply_FFT.py -> cube_POCS_interpolation_3D.py -> cube_apply_IFFT.py);
with the three drivers of this package that is still one host round trip per step.  :func:`interpolate_time_cube` chains the same
device code -- time -> frequency (``p3d_time2freq_dev``), the POCS loop on every frequency slice (``p3d_pocs_run_dev``), frequency ->
time (``p3d_freq2time_dev``) -- on device buffers: one upload of the time cube, one download of the result.
"""
import numpy as np

from . import _ffi
from .functions import POCS as P


def interpolate_time_cube(x, mask, dt, t0=0.0, nfft=None, real_only=True, window=None, device=0, batch_slices=None, niter=50,
                          thresh_op='hard', thresh_model='exponential', eps=1e-9, alpha=1.0, p_max=0.99, p_min=1e-5, sqrt_decay=False,
                          decay_kind='values', version='regular', results=None):
    """
    ``x``: ``(nt, nil, nxl)`` float32 time-domain cube with zeros at missing traces; ``mask``: ``(nil, nxl)`` (1 = observed).
    ``dt`` / ``t0``: sample spacing and time of the first sample (cube_apply_FFT.py:240-254); ``real_only``: keep the non-negative
    frequencies only (``--compute_real``); ``window``: optional per-frequency weights (frequency filter of step 12).  The POCS
    parameters are those of :func:`functions.POCS.pocs_cube` (FFT transform).  Returns the interpolated ``(nfft, nil, nxl)`` float32
    cube -- the same numbers as ``freq2time(pocs_cube(time2freq(x)))``.
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 3:
        raise ValueError(f'x must be (nt, iline, xline), got shape {x.shape}')
    nt, nil, nxl = x.shape
    mask = np.asarray(mask)
    if mask.shape != (nil, nxl):
        raise ValueError(f'mask shape {mask.shape} does not match slice shape {(nil, nxl)}')
    if np.max(mask) > 1:
        raise ValueError(f'mask should be quasi-boolean (0 or 1) but has maximum of {np.max(mask)}')
    if thresh_model == 'data-driven' or thresh_op.endswith('percentile'):
        raise NotImplementedError('the device-resident pipeline covers the statistics-driven schedules and hard / soft / garrote')
    nfft = int(nfft or nt)
    nfreq = nfft // 2 + 1 if real_only else nfft
    ntr = nil * nxl
    niter, eps, p_max, alpha = int(niter), float(eps), float(p_max), float(alpha)
    if isinstance(p_min, str) and p_min != 'adaptive':
        p_min = float(p_min)
    step = int(batch_slices) if batch_slices else max(1, min(nfreq, (512 << 20) // (ntr * 8)))
    plan = P._get_plan(nil, nxl, min(step, nfreq), device, slot=14)
    lib = _ffi.lib()
    maskf = np.ascontiguousarray(mask, dtype=np.float32)
    win = None if window is None else np.ascontiguousarray(window, dtype=np.float32)
    if win is not None and win.shape != (nfreq,):
        raise ValueError(f'window must have {nfreq} entries')

    bufs = []

    def alloc(nbytes):
        bufs.append(plan.alloc(nbytes))
        return bufs[-1]

    try:
        tbuf = alloc(4 * max(nt, nfft) * ntr).upload(x)
        fbuf = alloc(8 * nfreq * ntr)
        obuf = alloc(8 * nfreq * ntr)
        mbuf = alloc(maskf.nbytes).upload(maskf)
        _ffi.check(lib.p3d_time2freq_dev(int(device), tbuf.ptr, nt, ntr, float(dt), float(t0), nfft, int(bool(real_only)),
                                         None if win is None else _ffi._ptr(win), fbuf.ptr))
        for lo in range(0, nfreq, step):
            n = min(step, nfreq - lo)
            off = lo * ntr * 8
            stats = plan.prime_dev(fbuf.ptr + off, _ffi.P3D_C64, mbuf.ptr, n)   # statistics = first pass of the job
            active = stats[:, 2] > 0                     # max |X0| = 0 <=> all-zero slice (POCS.py:515-521)
            stats[~active] = 1.0
            tau = P._schedule_from_stats(stats, ntr, thresh_model, niter, p_max, p_min, decay_kind)
            if sqrt_decay:
                tau = np.sqrt(tau)
            done, sums, _ = plan.run_dev(fbuf.ptr + off, _ffi.P3D_C64, mbuf.ptr, tau, niter, obuf.ptr + off, n, thresh_op=thresh_op,
                                         version=version, eps=eps, alpha=alpha, active=active, primed=True)
            if results is not None:
                results.extend(P._result_rows(done, sums, 0.0))
        kidx = np.arange(nfreq, dtype=np.int32)
        _ffi.check(lib.p3d_freq2time_dev(int(device), obuf.ptr, nfreq, _ffi._ptr(kidx), ntr, float(dt), float(t0), nfft,
                                         int(bool(real_only)), tbuf.ptr))
        return tbuf.download((nfft, nil, nxl), np.float32)
    finally:
        for b in bufs:
            b.free()
