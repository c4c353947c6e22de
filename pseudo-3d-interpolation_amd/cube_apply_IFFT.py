"""
Step 14 -- inverse FFT along the frequency axis of a (pseudo-)3D cube, on the GPU.

Mirror of ``pseudo_3D_interpolation/cube_apply_IFFT.py``: same command line (:20-32); rebuilds the complex variable from
its ``.real`` / ``.imag`` parts (:74-79); the arithmetic of ``xrft.ifft(..., true_phase=True, true_amplitude=True)``
(:83-94) runs in ``p3d_freq2time`` (include/p3d.h); optional clip + global min-max rescale (:121-140); output name
``<name with prefix -> 'twt'>_interp-freq[_rescale-env]`` (:143-147).
"""
import argparse
import datetime
import os
import sys

import numpy as np
import yaml

from . import _ffi
from .cube_io import open_cube, save_cube
from .functions.utils import rescale_dask, xprint


# fmt: off
def define_input_args():  # noqa
    parser = argparse.ArgumentParser(
        description='Apply inverse FFT along frequency axis of (pseudo-)3D cube.')
    parser.add_argument('path_cube', type=str,
                        help='Input path of 3D cube.')
    parser.add_argument('--params_netcdf', type=str, required=True,
                        help='Path of netCDF parameter file (*.yaml).')
    parser.add_argument('--compute_real', action='store_true',
                        help='Compute IFFT assuming real input was used for previously applied FFT.')
    parser.add_argument('--rescale-envelope', action='store_true', help='Rescale envelope data to [0-1].')
    parser.add_argument('--verbose', '-V', type=int, nargs='?', default=0, const=1, choices=[0, 1, 2],
                        help='Level of output verbosity (default: 0)')
    return parser
# fmt: on


def main(argv=sys.argv, return_dataset=False):  # noqa
    """Apply inverse FFT along _frequency_ axis wrapper function."""
    TODAY = datetime.date.today().strftime('%Y-%m-%d')
    SCRIPT = os.path.splitext(os.path.basename(__file__))[0]
    args = define_input_args().parse_args(argv[1:])

    dir_work, file = os.path.split(args.path_cube)
    with open(args.params_netcdf, 'r') as f_attrs:
        kwargs_nc = yaml.safe_load(f_attrs)

    cube = open_cube(args.path_cube)
    dim = cube.slice_dim()
    prefix = dim.split('_')[0]
    names = list(cube.data_vars)
    var_freq = [v for v in names if prefix in v][0]
    var = cube.var_attrs.get(var_freq, {}).get('original_var', '_'.join(var_freq.split('.')[0].split('_')[1:]))

    # restore the complex array from the float parts written by step 13 (cube_POCS_interpolation_3D.py:160-164)
    var_real = [v for v in names if 'real' in v]
    var_imag = [v for v in names if 'imag' in v]
    if var_real and var_imag:
        spec = cube.data_vars[var_real[0]] + 1j * cube.data_vars[var_imag[0]]
        dims = cube.dims[var_real[0]]
    else:
        spec, dims = cube.data_vars[var_freq], cube.dims[var_freq]
    if dims[0] != dim:
        spec = np.moveaxis(spec, dims.index(dim), 0)
        dims = (dim,) + tuple(d for d in dims if d != dim)

    cattrs = cube.coord_attrs.get(dim, {})
    freqs = np.asarray(cube.coords[dim], dtype=np.float64)
    real_only = bool(args.compute_real or cattrs.get('real_only', False))
    nstored = int(cattrs.get('nfft', freqs.size))            # frequency samples before --drop-filtered-freq
    nfft = int(cattrs.get('nfft_time', 2 * (nstored - 1) if real_only else nstored))
    dt = float(cattrs.get('dt', 1.0 / (nfft * (freqs[1] - freqs[0])) if freqs.size > 1 else 1.0))
    t0 = float(cattrs.get('direct_lag', 0.0))
    kidx = cattrs.get('kidx')

    xprint('Compute inverse FFT along time axis', kind='info', verbosity=args.verbose)
    data = _ffi.freq2time(spec, dt, t0, nfft=nfft, real_only=real_only, kidx=kidx)

    out = cube.copy_meta()
    out.coords.pop(dim, None)
    out.coord_attrs.pop(dim, None)
    out.coords['twt'] = (t0 + dt * np.arange(nfft)).astype(np.float32)   # "fix rounding errors.." (:101-103)
    out.coord_attrs['twt'] = {}
    out.data_vars[var] = data.astype(np.float32, copy=False)
    out.dims[var] = ('twt',) + tuple(dims[1:])
    if 'fold' in cube.data_vars:
        out.data_vars['fold'], out.dims['fold'] = cube.data_vars['fold'], cube.dims['fold']

    out.attrs.update({
        'long_name': cube.attrs.get('long_name', '').split(' (')[0] + ' (interpolated)',
        'history': cube.attrs.get('history', '') + f'{SCRIPT}: IFFT({var});',
        'text': cube.attrs.get('text', '') + f'\n{TODAY}: INVERSE FFT(FREQ -> TIME)',
    })
    if kwargs_nc is not None:
        out.var_attrs[var] = dict(kwargs_nc.get('attrs_time', {}).get(var.split('_')[0], {}) or {})
        out.coord_attrs['twt'].update(kwargs_nc.get('attrs_time', {}).get('twt', {}) or {})
        if 'spacing' in out.coord_attrs['twt']:
            out.coord_attrs['twt']['dt'] = float(f"{out.coord_attrs['twt'].pop('spacing'):g}")

    if args.rescale_envelope:  # clip < 0, then global min-max rescale to [0, 1]
        clipped = np.where(out.data_vars[var] < 0, 0, out.data_vars[var])
        amin, amax = clipped.min(), clipped.max()
        xprint(f'amin: {amin}', kind='debug', verbosity=args.verbose)
        xprint(f'amax: {amax}', kind='debug', verbosity=args.verbose)
        out.data_vars[var] = rescale_dask(clipped, amin=amin, amax=amax).astype(np.float32)

    tsuffix = '_rescale-env' if args.rescale_envelope else ''
    basename, fsuffix = os.path.splitext(file)
    path_out = os.path.join(dir_work, basename.replace(prefix, 'twt') + f'_interp-freq{tsuffix}{fsuffix}')
    save_cube(out, path_out)
    if return_dataset:
        return out


if __name__ == '__main__':
    main()
