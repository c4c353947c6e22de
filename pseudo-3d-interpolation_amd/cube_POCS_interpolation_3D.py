"""
Step 13 -- interpolating a sparse 3D volume with the iterative POCS algorithm, on MI355X GPUs.

Mirror of ``pseudo_3D_interpolation/cube_POCS_interpolation_3D.py``: same command line (``define_input_args`` :68-84),
same YAML file (top-level ``dim, var, batch_chunk, output_runtime_results`` and the ``metadata`` block that is splatted
into ``POCS_algorithm``; ``n_workers / processes / threads_per_worker / memory_limit`` are accepted and ignored -- the
dask ``LocalCluster`` slice farm (:291-340) is replaced by whole batches of slices on the GPU), same mask rule
``fold <= 1 ? fold : 1`` (:242-244), same output naming (:146-157, :223-228), ``.real`` / ``.imag`` split (:160-164),
per-batch ``slice-XXXX-YYYY.out`` runtime files merged into ``runtimes_<prefix>.txt`` (:177-195), same attributes
(:346-367).  Cubes are ``.npz`` files (cube_io.py) or netCDF when xarray is installed.  ``transform_kind``: FFT, WAVELET (with the ``wavelet`` key, default
coif5, :261) or SHEARLET (spectra from functions/shearlets.py instead of FFST, :269-272); the other kinds raise as the reference does when their third-party package is missing (:287-288).
"""
import argparse
import datetime
import glob
import os
import sys
from functools import partial

import numpy as np
import yaml

from .cube_io import open_cube, save_cube
from .functions.POCS import APOCS, FPOCS, POCS, pocs_cube  # noqa: F401
from .functions.utils import xprint

POCS_VERSIONS = {'POCS': POCS, 'FPOCS': FPOCS, 'APOCS': APOCS}
ffloat = partial(np.format_float_positional, trim='-')


def define_input_args():  # noqa
    parser = argparse.ArgumentParser(description='Interpolate sparse 3D cube using POCS algorithm.')
    parser.add_argument(
        'path_cube', type=str, help='Input path of 3D cube'
    )
    parser.add_argument(
        '--path_pocs_parameter', type=str, required=True,
        help='Path of netCDF parameter file (YAML format).'
    )
    parser.add_argument(
        '--path_output_dir', type=str, help='Output directory for interpolated slices.'
    )
    parser.add_argument(
        "--verbose", "-V", type=int, nargs="?", default=0, choices=[0, 1, 2],
        help="Level of output verbosity (default: 0)",
    )
    return parser


def create_file_path(coord, prefix=None, root_path='.', suffix='.nc'):
    """Batch file name from the first / last coordinate of the batch (cube_POCS_interpolation_3D.py:146-157)."""
    if prefix is None:
        prefix = datetime.datetime.today().strftime('%Y-%m-%d')
    coord = np.atleast_1d(coord)
    return os.path.join(root_path, f'{prefix}_{coord[0]:06.3f}_{coord[-1]:06.3f}{suffix}')


def split_complex_variable(cube, var):
    """Replace a complex variable by ``<var>.real`` and ``<var>.imag`` (cube_POCS_interpolation_3D.py:160-164)."""
    data = cube.data_vars.pop(var)
    dims = cube.dims.pop(var)
    attrs = cube.var_attrs.pop(var, {})
    for part, values in (('real', data.real), ('imag', data.imag)):
        cube.data_vars[f'{var}.{part}'] = np.ascontiguousarray(values)
        cube.dims[f'{var}.{part}'] = dims
        cube.var_attrs[f'{var}.{part}'] = dict(attrs)
    return cube


def combine_runtime_results(dir_files: str, prefix: str = 'combined', fsuffix: str = 'out') -> None:
    """Concatenate the per-batch runtime files (cube_POCS_interpolation_3D.py:177-195)."""
    files = glob.glob(os.path.join(dir_files, f'*.{fsuffix}'))
    with open(os.path.join(dir_files, f'runtimes_{prefix}.txt'), mode='w', newline='\n') as fout:
        for file in files:
            with open(file, mode='r') as f:
                fout.write(f.read())


def main(argv=sys.argv, return_dataset=False):
    """Interpolate sparse 3D cube."""
    SCRIPT = os.path.basename(__file__)
    TODAY = datetime.date.today().strftime('%Y-%m-%d')
    args = define_input_args().parse_args(argv[1:])
    verbose = args.verbose

    xprint("Load POCS parameter from config file", kind="info", verbosity=verbose)
    with open(args.path_pocs_parameter, mode="r") as f:
        cfg = yaml.safe_load(f)
        cfg['metadata']['transform_kind'] = cfg['metadata']['transform_kind'].upper()
    metadata = cfg['metadata']
    TRANSFORM = metadata['transform_kind']

    path_cube = args.path_cube
    dir_work, file = os.path.split(path_cube)
    filename, suffix = os.path.splitext(file)
    prefix = f"{filename}_{TRANSFORM}_{metadata['thresh_op']}_niter-{metadata['niter']}"
    out_path = args.path_output_dir if args.path_output_dir is not None else os.path.join(dir_work, prefix)
    if not os.path.isdir(out_path):
        os.mkdir(out_path)

    cube = open_cube(path_cube)
    dim = cfg['dim']
    var = cfg.get('var', [v for v in list(cube.data_vars) if v != 'fold'][0])
    fold = np.asarray(cube.data_vars['fold'])
    mask = np.where(fold <= 1, fold, 1)                      # cube_POCS_interpolation_3D.py:242-244
    data = np.asarray(cube.data_vars[var])
    if cube.dims[var][0] != dim:
        data = np.moveaxis(data, cube.dims[var].index(dim), 0)
    COMPLEX = np.iscomplexobj(data)

    with open(os.path.join(out_path, f'parameter_{prefix}.yml'), mode='w', newline='\n') as f:
        yaml.safe_dump(metadata, f)

    # the GPU path selects the transform by `transform_kind` (+ `wavelet`); the callables are kept for signature compatibility
    if TRANSFORM == 'FFT':
        metadata['transform'] = np.fft.fft2
        metadata['itransform'] = np.fft.ifft2
    elif TRANSFORM == 'WAVELET':                             # cube_POCS_interpolation_3D.py:260-266
        wavelet = metadata.get('wavelet', 'coif5')
        wavelet_mode = 'smooth'
        metadata['wavelet'] = wavelet
        metadata['transform'] = metadata['itransform'] = None
        prefix += f'_{wavelet}-{wavelet_mode}'
    elif TRANSFORM == 'SHEARLET':                            # cube_POCS_interpolation_3D.py:269-274
        from .functions.shearlets import scalesShearsAndSpectra
        Psi = scalesShearsAndSpectra(data.shape[1:], numOfScales=None, realCoefficients=True, fftshift_spectra=True, dtype=np.float32)
        metadata['transform'] = metadata['itransform'] = None
    else:
        raise ValueError(f'Transform < {metadata["transform_kind"]} > is not supported.')

    coord = np.asarray(cube.coords[dim])
    step = int(cfg['batch_chunk'])
    indices = list(range(0, coord.size + step, step))
    batches = [slice(a, min(b, coord.size)) for a, b in zip(indices[:-1], indices[1:]) if a < coord.size]
    exclude_keys = ['transform', 'itransform', 'results_dict', 'path_results']
    attrs_domain = '(frequency domain)' if 'freq' in dim else '(time domain)'
    kwargs = {k: v for k, v in metadata.items() if k not in exclude_keys}
    dims_out = (dim,) + tuple(d for d in cube.dims[var] if d != dim)

    def wrap(block, sl):
        ds = cube.copy_meta()
        ds.coords[dim] = coord[sl]
        ds.data_vars[f'{var}_interp'] = block
        ds.dims[f'{var}_interp'] = dims_out
        ds.var_attrs[f'{var}_interp'] = dict(cube.var_attrs.get(var, {}))
        ds.data_vars['fold'], ds.dims['fold'] = cube.data_vars['fold'], cube.dims['fold']
        if COMPLEX:
            split_complex_variable(ds, f'{var}_interp')
        ds.attrs.update({
            'description': f'Interpolated pseudo-3D cube using {metadata["transform_kind"]} transform created from TOPAS profiles '
                           + attrs_domain,
            'interp_params_keys': ';'.join([k for k in metadata if k not in exclude_keys]),
            'interp_params_vals': ';'.join([str(metadata[k]) for k in metadata if k not in exclude_keys]),
            'history': cube.attrs.get('history', '') + f'{SCRIPT}:{metadata["transform_kind"]} {attrs_domain};',
            'text': cube.attrs.get('text', '') + f'\n{TODAY}: {metadata["transform_kind"]} {attrs_domain.upper()}',
        })
        return ds

    merged = np.empty_like(data)
    # `batch_chunk` is the unit of the OUTPUT (one file and one runtime record per batch, as the reference writes them); the GPU is handed several
    # consecutive batches per call -- a 20-slice batch of the documented example is a fraction of what keeps the device and the host link busy
    # (pocs_cube overlaps upload, loop and download of ~128-MiB portions on several plans inside ONE call, not across calls)
    try:
        group_gib = float(os.environ.get('P3D_CLI_GROUP_GIB', '4'))
    except ValueError:
        raise ValueError(f"P3D_CLI_GROUP_GIB must be a number of GiB, got {os.environ['P3D_CLI_GROUP_GIB']!r}") from None
    if not group_gib >= 0:
        raise ValueError(f'P3D_CLI_GROUP_GIB must be >= 0, got {group_gib}')
    group_bytes = group_gib * 2.0 ** 30
    per_slice = data[0].nbytes if len(data) else 1
    groups, cur, cur_bytes = [], [], 0
    for sl in batches:
        nb = (sl.stop - sl.start) * per_slice
        if cur and cur_bytes + nb > group_bytes:
            groups.append(cur)
            cur, cur_bytes = [], 0
        cur.append(sl)
        cur_bytes += nb
    if cur:
        groups.append(cur)
    for group in groups:
        span = slice(group[0].start, group[-1].stop)
        results = []
        aux = Psi if TRANSFORM == 'SHEARLET' else None      # :307
        # (straight into the merged cube: no second group-sized result array.  `runtime` of a slice's record is its GROUP's wall time divided by
        # the group's slices -- the reference records a batch's time / its slices; INTEGRATION.md says so)
        pocs_cube(data[span], mask, results=results, auxiliary_data=aux, out=merged[span], **kwargs)
        for sl in group:
            block = merged[sl]
            save_cube(wrap(block, sl), create_file_path(coord[sl], prefix=prefix, root_path=out_path, suffix=suffix))
            if cfg.get('output_runtime_results'):
                with open(os.path.join(out_path, f"slice-{sl.start:04d}-{sl.stop:04d}.out"), mode='a', newline='\n') as f:
                    for info in results[sl.start - span.start:sl.stop - span.start]:   # one line per slice: niter;runtime;cost_1;...  (POCS.py:649-651)
                        f.write(';'.join([str(i) for i in [info['niterations'], info['runtime']] + info['costs']]) + '\n')
            xprint(f'batch {sl.start}-{sl.stop} done', kind='info', verbosity=verbose)

    if cfg.get('output_runtime_results'):
        combine_runtime_results(out_path, prefix=prefix)

    xprint('Write combinded netCDF file to disk', kind='info', verbosity=verbose)
    full = wrap(merged, slice(0, coord.size))
    save_cube(full, f'{out_path}{suffix}')
    if return_dataset:
        return full


if __name__ == '__main__':
    main()
