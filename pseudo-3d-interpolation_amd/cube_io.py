"""Minimal cube container for the step 12-14 drivers.

The reference reads and writes netCDF through xarray/h5netcdf (``xr.open_dataset(..., engine='h5netcdf')``,
cube_POCS_interpolation_3D.py:231-233).  Neither is installed in the build image, so the drivers here work on a
small in-memory :class:`Cube` (variables + coordinates + attributes, the subset of ``xr.Dataset`` they need) that is
stored natively as ``.npz`` and converted from / to netCDF when xarray is importable."""
import json
import os

import numpy as np

from .functions.backends import xarray_enabled


class Cube:
    def __init__(self, data_vars=None, dims=None, coords=None, attrs=None, var_attrs=None, coord_attrs=None):
        self.data_vars = dict(data_vars or {})      # name -> ndarray
        self.dims = dict(dims or {})                # name -> tuple of dimension names
        self.coords = dict(coords or {})            # dimension name -> 1-D ndarray
        self.attrs = dict(attrs or {})
        self.var_attrs = {k: dict(v) for k, v in (var_attrs or {}).items()}
        self.coord_attrs = {k: dict(v) for k, v in (coord_attrs or {}).items()}

    def slice_dim(self):
        """The dimension that is neither iline nor xline (cube_apply_FFT.py:210)."""
        for name, dims in self.dims.items():
            for d in dims:
                if d not in ('iline', 'xline'):
                    return d
        raise ValueError('cube has no slice dimension')

    def copy_meta(self):
        return Cube({}, {}, dict(self.coords), dict(self.attrs), {}, {k: dict(v) for k, v in self.coord_attrs.items()})


def _jsonable(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, (np.generic,)):
            v = v.item()
        elif isinstance(v, np.ndarray):
            v = v.tolist()
        out[k] = v
    return out


def save_cube(cube, path):
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npz':
        meta = dict(dims=cube.dims, attrs=_jsonable(cube.attrs), var_attrs={k: _jsonable(v) for k, v in cube.var_attrs.items()},
                    coord_attrs={k: _jsonable(v) for k, v in cube.coord_attrs.items()})
        arrays = {f'v/{k}': v for k, v in cube.data_vars.items()}
        arrays.update({f'c/{k}': v for k, v in cube.coords.items()})
        np.savez(path, __meta__=np.array(json.dumps(meta)), **arrays)
        return path
    if ext == '.nc':
        if not xarray_enabled:
            raise ImportError('writing netCDF needs xarray + h5netcdf; use a .npz path in this environment')
        import xarray as xr
        ds = xr.Dataset({k: (cube.dims[k], v, cube.var_attrs.get(k, {})) for k, v in cube.data_vars.items()},
                        coords={k: (k, v, cube.coord_attrs.get(k, {})) for k, v in cube.coords.items()}, attrs=cube.attrs)
        ds.to_netcdf(path, engine='h5netcdf', invalid_netcdf=True)
        return path
    raise ValueError(f'unsupported cube file type {ext!r} (use .npz or .nc)')


def open_cube(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npz':
        with np.load(path, allow_pickle=False) as z:
            meta = json.loads(str(z['__meta__']))
            data = {k[2:]: z[k] for k in z.files if k.startswith('v/')}
            coords = {k[2:]: z[k] for k in z.files if k.startswith('c/')}
        return Cube(data, {k: tuple(v) for k, v in meta['dims'].items()}, coords, meta.get('attrs'), meta.get('var_attrs'),
                    meta.get('coord_attrs'))
    if ext == '.nc':
        if not xarray_enabled:
            raise ImportError('reading netCDF needs xarray + h5netcdf; convert the cube to .npz in this environment')
        import xarray as xr
        ds = xr.open_dataset(path, engine='h5netcdf').load()
        return Cube({k: ds[k].values for k in ds.data_vars}, {k: tuple(ds[k].dims) for k in ds.data_vars},
                    {k: ds[k].values for k in ds.coords}, dict(ds.attrs), {k: dict(ds[k].attrs) for k in ds.data_vars},
                    {k: dict(ds[k].attrs) for k in ds.coords})
    raise ValueError(f'unsupported cube file type {ext!r} (use .npz or .nc)')
