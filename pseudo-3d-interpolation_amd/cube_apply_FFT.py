"""
Step 12 -- forward FFT along the time axis of a (pseudo-)3D cube, on the GPU.

Mirror of ``pseudo_3D_interpolation/cube_apply_FFT.py``: same command line (``define_input_args`` :24-45), same filter
window builders (:49-181, pure NumPy), same naming and attribute bookkeeping (:296-313).  The arithmetic that the
reference delegates to ``xrft.fft(..., shift=False, true_phase=True, true_amplitude=True, shape=...)`` (:240-254) runs
in ``p3d_time2freq`` (include/p3d.h).  Cubes are ``.npz`` files (see cube_io.py) or netCDF when xarray is installed.
"""
import argparse
import datetime
import os
import sys
import warnings

import numpy as np
import yaml

from . import _ffi
from .cube_io import Cube, open_cube, save_cube
from .functions.utils import xprint


# fmt: off
def define_input_args():  # noqa
    parser = argparse.ArgumentParser(
        description='Apply FFT along time axis of (pseudo-)3D cube.')
    parser.add_argument('path_cube', type=str,
                        help='Input path of 3D cube')
    parser.add_argument('--params_netcdf', type=str, required=True,
                        help='Path of netCDF parameter file (YAML format).')
    parser.add_argument('--prefix', type=str, default='freq',
                        help='Prefix for new netCDF variable and coordinate.')
    parser.add_argument('--compute_real', action='store_true',
                        help='Compute FFT assuming real input and thus discarting redundant negative frequencies.')
    parser.add_argument('--upsampling-factor', type=int, default=1,
                        help='Increase resolution of FFT by `upsampling-factor`.')
    # filter options
    parser.add_argument('--filter', type=str, default=None, choices=['lowpass', 'highpass', 'bandpass'],
                        help='Optional filter to apply prior to FFT computation.')
    parser.add_argument('--filter_freqs', type=int, nargs='+', help='Filter corner frequencies (in Hz).')
    parser.add_argument('--drop-filtered-freq', action='store_true', help='Drop filtered frequency samples.')
    #
    parser.add_argument('--verbose', '-V', type=int, nargs='?', default=0, const=1, choices=[0, 1, 2],
                        help='Level of output verbosity (default: 0)')
    return parser
# fmt: on


def _get_stopband(nstopband: int, kind: str):
    """Half of a Hann window with an odd number of ``2*nstopband (+1)`` points: rising flank for `highpass`, falling for
    `lowpass` (cube_apply_FFT.py:49-58)."""
    size = nstopband * 2
    size += 1 if size % 2 == 0 else 0
    flank = slice(1, size // 2 + 1) if kind == 'highpass' else slice(size // 2, -1)
    return np.hanning(size)[flank]


def _get_const_values(kind: str):
    """Padding values left / right of the taper (cube_apply_FFT.py:61-69)."""
    return {'highpass': (0, 1), 'lowpass': (1, 0), 'bandpass': (0, 0)}[kind]


def get_freq_filter_win(filter_freqs: list, frequencies, dim: str = 'freq_twt', filter_type: str = 'lowpass'):
    """
    Filter window (`highpass`, `lowpass` or `bandpass`) in the frequency domain, values in [0, 1]
    (cube_apply_FFT.py:72-143).  ``frequencies`` is a 1-D array (or anything with ``.data``); returns a NumPy array of
    the same length (the reference wraps it in an ``xr.DataArray``).
    """
    freqs = np.asarray(getattr(frequencies, 'data', frequencies))
    if filter_type in ['lowpass', 'highpass']:
        fmin, fmax = min(filter_freqs), max(filter_freqs)
        pad_values = _get_const_values(kind=filter_type)
        n_lower = np.count_nonzero(freqs < fmin)
        n_stop = np.count_nonzero((freqs >= fmin) & (freqs <= fmax))
        n_higher = np.count_nonzero(freqs > fmax)
        taper = _get_stopband(n_stop, kind=filter_type)
    elif filter_type == 'bandpass':
        filter_freqs.sort()
        f1, f2, f3, f4 = filter_freqs
        pad_values = _get_const_values(filter_type)
        n_lower = np.count_nonzero(freqs < f1)
        n_low = np.count_nonzero((freqs >= f1) & (freqs <= f2))
        n_pass = np.count_nonzero((freqs > f2) & (freqs < f3))
        n_high = np.count_nonzero((freqs >= f3) & (freqs <= f4))
        n_higher = np.count_nonzero(freqs > f4)
        taper = np.hstack((_get_stopband(n_low, kind='highpass'), np.ones((n_pass,)), _get_stopband(n_high, kind='lowpass')))
    else:
        raise ValueError(f'unknown filter type {filter_type!r}')
    return np.pad(taper, pad_width=(n_lower, n_higher), mode='constant', constant_values=(pad_values,))


def get_freq_filter_mask(frequencies, dim: str = 'freq_twt', freqs: list = None, filter_type: str = 'lowpass'):
    """Boolean mask of the frequency samples a filter keeps (cube_apply_FFT.py:146-181)."""
    f = np.asarray(getattr(frequencies, 'data', frequencies))
    corner = sorted(freqs)
    if filter_type == 'lowpass':
        assert len(freqs) == 2, 'Please provide filter frequencies as [fmin, fmax]'
        return f <= corner[-1]
    if filter_type == 'highpass':
        assert len(freqs) == 2, 'Please provide filter frequencies as [fmin, fmax]'
        return f >= corner[0]
    if filter_type == 'bandpass':
        assert len(freqs) == 4, 'Please provide filter frequencies as [f1, f2, f3, f4]'
        return np.logical_and(f >= corner[0], f <= corner[-1])
    raise ValueError(f'unknown filter type {filter_type!r}')


def main(argv=sys.argv, return_dataset=False):  # noqa
    """Apply FFT along _time_ axis wrapper function."""
    TODAY = datetime.date.today().strftime('%Y-%m-%d')
    SCRIPT = os.path.splitext(os.path.basename(__file__))[0]
    args = define_input_args().parse_args(argv[1:])

    dir_work, filename = os.path.split(args.path_cube)
    basename, suffix = os.path.splitext(filename)
    fout = basename.replace('twt', f'{args.prefix}')
    fout += f'_up-{args.upsampling_factor}' if args.upsampling_factor > 1 else ''
    fout += '-trunc' if args.drop_filtered_freq else ''
    path_cube_freq = os.path.join(dir_work, fout + suffix)
    prefix = f'{args.prefix}_'

    with open(args.params_netcdf, 'r') as f_attrs:
        kwargs_nc = yaml.safe_load(f_attrs)

    cube = open_cube(args.path_cube)
    dim = cube.slice_dim()
    var = [v for v in cube.data_vars if v not in ['fold', 'amp_ref']][0]
    var_new, dim_new = f'{prefix}{var}', f'{prefix}{dim}'
    data = np.asarray(cube.data_vars[var])
    if cube.dims[var][0] != dim:  # slice-major layout expected (cube_binning_3D.py:1313-1351)
        data = np.moveaxis(data, cube.dims[var].index(dim), 0)
    t = np.asarray(cube.coords[dim], dtype=np.float64)

    xprint('Compute FFT along time axis', kind='info', verbosity=args.verbose)
    if t.size % 2 != 0:
        warnings.warn(f'Selected dim `{dim}` has odd length ({t.size}), which causes issues for inverse FFT. '
                      'Last slice will be removed!')
        data, t = data[:-1], t[:-1]
    nt = t.size
    dt = float(t[1] - t[0]) if nt > 1 else 1.0
    nfft = args.upsampling_factor * nt
    history_reso = f' FACTOR x{args.upsampling_factor}' if args.upsampling_factor > 1 else ''
    freqs = np.fft.rfftfreq(nfft, dt) if args.compute_real else np.fft.fftfreq(nfft, dt)

    window = None
    attrs_var, history_filter = {}, ''
    if args.filter is not None:
        if args.filter_freqs is None:
            raise ValueError('Filter frequencies must be specified!')
        units = cube.coord_attrs.get(dim, {}).get('units')
        divisor = 1000 if units == 'ms' else 1
        filter_freqs = [f / divisor for f in args.filter_freqs]
        xprint(f'Apply > {args.filter} < filter ({"/".join([str(round(f * divisor)) for f in filter_freqs])} Hz) in frequency domain',
               kind='info', verbosity=args.verbose)
        window = get_freq_filter_win(filter_freqs, frequencies=freqs, dim=dim_new, filter_type=args.filter)
        _filter_freq_str = '/'.join(str(f) for f in args.filter_freqs)
        attrs_var = {'filter': args.filter, 'filter_freq_Hz': _filter_freq_str}
        history_filter = f' {args.filter.upper()} ({_filter_freq_str} Hz)'

    spec = _ffi.time2freq(data, dt, float(t[0]), nfft=nfft, real_only=args.compute_real, window=window)

    out = cube.copy_meta()
    out.coords.pop(dim, None)
    out.coord_attrs.pop(dim, None)
    out.coords[dim_new] = freqs
    out.coord_attrs[dim_new] = {'direct_lag': float(t[0]), 'spacing': float(freqs[1] - freqs[0]) if freqs.size > 1 else 0.0,
                                'dt': dt}
    if args.filter is not None and args.drop_filtered_freq and args.filter == 'lowpass':
        out.coord_attrs[dim_new]['nfft'] = int(freqs.size)  # original size, needed for the inverse transform
        keep = get_freq_filter_mask(freqs, dim_new, freqs=filter_freqs, filter_type=args.filter)
        spec, out.coords[dim_new] = spec[keep], freqs[keep]
        out.coord_attrs[dim_new]['kidx'] = np.flatnonzero(keep).tolist()
    elif args.drop_filtered_freq and args.filter != 'lowpass':
        warnings.warn(f'Filter type `{args.filter}` does not support dropping of frequency slices')
    out.coord_attrs[dim_new]['nfft_time'] = int(nfft)
    out.coord_attrs[dim_new]['real_only'] = bool(args.compute_real)

    out.data_vars[var_new] = spec.astype(np.complex64, copy=False)
    out.dims[var_new] = (dim_new,) + tuple(d for d in cube.dims[var] if d != dim)
    if 'fold' in cube.data_vars:
        out.data_vars['fold'], out.dims['fold'] = cube.data_vars['fold'], cube.dims['fold']

    out.attrs.update({
        'long_name': cube.attrs.get('long_name', '') + ' (frequency domain)',
        'description': cube.attrs.get('description', '') + ' (frequency domain)',
        'history': cube.attrs.get('history', '') + f'{SCRIPT}: FFT({var}){history_reso}{history_filter};',
        'text': cube.attrs.get('text', '') + f'\n{TODAY}: FFT(TIME){history_reso}{history_filter}',
    })
    out.var_attrs[var_new] = dict(cube.var_attrs.get(var, {}), original_var=var)
    if kwargs_nc is not None:
        out.var_attrs[var_new].update(kwargs_nc.get('attrs_freq', {}).get('data', {}) or {})
        out.var_attrs[var_new].update(attrs_var)
        out.coord_attrs[dim_new].update(kwargs_nc.get('attrs_freq', {}).get('new_dim', {}) or {})

    save_cube(out, path_cube_freq)
    if return_dataset:
        return out


if __name__ == '__main__':
    main()
