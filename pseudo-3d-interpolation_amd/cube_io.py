"""Minimal cube container for the step 12-14 drivers.

The reference reads and writes netCDF through xarray/h5netcdf (``xr.open_dataset(..., engine='h5netcdf')``,
cube_POCS_interpolation_3D.py:231-233).  Neither is installed in the build image, so the drivers here work on a
small in-memory :class:`Cube` (variables + coordinates + attributes, the subset of ``xr.Dataset`` they need) that is
stored natively as ``.npz``.  netCDF (``.nc``) goes through xarray + h5netcdf where they are installed and otherwise through
h5py alone: the files the workflow exchanges are netCDF-4 = HDF5 with dimension scales, and the reader / writer below speak
exactly that layout (what ``xr.Dataset.to_netcdf(engine='h5netcdf')`` produces and ``xr.open_dataset(engine='h5netcdf')``
accepts: cube_binning_3D.py:1313-1351, cube_apply_FFT.py:319, cube_POCS_interpolation_3D.py:231-244, 342-376, 392-405):

* every dimension is a coordinate dataset turned into an HDF5 dimension scale (``CLASS = DIMENSION_SCALE``, ``NAME``,
  ``_Netcdf4Dimid`` = position in the file's dimension order);
* every variable is a dataset whose axes carry ``DIMENSION_LIST`` references to those scales; its attributes are HDF5
  attributes, strings as variable-length UTF-8, ``_FillValue`` = NaN on floating-point variables (xarray's default encoding);
* complex variables (the frequency-domain cube of step 12, written with ``invalid_netcdf=True``) are HDF5 compounds
  ``{r, i}`` -- h5py's native mapping of NumPy complex types;
* CF packing on read: ``scale_factor`` / ``add_offset`` are applied, ``_FillValue`` / ``missing_value`` become NaN in
  floating-point data."""
import json
import os

import numpy as np

from .functions.backends import h5py_enabled, xarray_enabled

# attributes that belong to the HDF5 / netCDF-4 machinery, not to the user
_NC_INTERNAL = {'CLASS', 'NAME', 'DIMENSION_LIST', 'REFERENCE_LIST', '_Netcdf4Dimid', '_Netcdf4Coordinates', '_NCProperties',
                '_nc3_strict', '_FillValue', 'missing_value', 'scale_factor', 'add_offset'}
# NAME of a dimension scale that carries no coordinate values; netCDF-C (nc4hdf.c) and h5netcdf write this text FOLLOWED by the
# dimension's length formatted '%10d'
_NOT_A_VARIABLE = 'This is a netCDF dimension but not a netCDF variable.'


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
            if h5py_enabled:
                return _save_nc_h5py(cube, path)
            raise ImportError('writing netCDF needs xarray + h5netcdf, or h5py; use a .npz path in this environment')
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
            if h5py_enabled:
                return _open_nc_h5py(path)
            raise ImportError('reading netCDF needs xarray + h5netcdf, or h5py; convert the cube to .npz in this environment')
        import xarray as xr
        ds = xr.open_dataset(path, engine='h5netcdf').load()
        return Cube({k: ds[k].values for k in ds.data_vars}, {k: tuple(ds[k].dims) for k in ds.data_vars},
                    {k: ds[k].values for k in ds.coords}, dict(ds.attrs), {k: dict(ds[k].attrs) for k in ds.data_vars},
                    {k: dict(ds[k].attrs) for k in ds.coords})
    raise ValueError(f'unsupported cube file type {ext!r} (use .npz or .nc)')


# ---- netCDF-4 through h5py alone ---------------------------------------------------------------------------------------------
def _attr_out(v):
    """Attribute value as netCDF stores it: strings as they are, numbers as 1-element arrays of their NumPy type."""
    if isinstance(v, (str, bytes)):
        return v
    if isinstance(v, bool):
        return np.array([int(v)], np.int8)
    a = np.asarray(v)
    if a.dtype.kind in 'US':
        return str(v) if a.ndim == 0 else ';'.join(str(x) for x in a.ravel())
    if a.dtype.kind == 'O':
        return str(v)
    return a.reshape(-1) if a.ndim == 0 else a


def _attr_in(v):
    if isinstance(v, bytes):
        return v.decode('utf-8', 'replace')
    if isinstance(v, np.ndarray):
        if v.dtype.kind == 'S':
            v = v.astype('U')
        if v.dtype.kind == 'O':
            v = np.array([x.decode('utf-8', 'replace') if isinstance(x, bytes) else x for x in v.ravel()]).reshape(v.shape)
        if v.size == 1:
            return v.reshape(-1)[0].item() if v.dtype.kind != 'U' else str(v.reshape(-1)[0])
    if isinstance(v, np.generic):
        return v.item()
    return v


def _save_nc_h5py(cube, path):
    import h5py
    dim_order = list(cube.coords)
    for name, dims in cube.dims.items():
        for d in dims:
            if d not in dim_order:
                dim_order.append(d)
    with h5py.File(path, 'w') as f:
        f.attrs['_NCProperties'] = f'version=2,h5py={h5py.__version__},hdf5={h5py.version.hdf5_version}'
        for i, d in enumerate(dim_order):
            if d in cube.coords:
                ds = f.create_dataset(d, data=np.asarray(cube.coords[d]))
            else:   # a dimension without coordinate values: an empty scale of the right length, as netCDF-4 writes it
                n = next(np.shape(cube.data_vars[k])[cube.dims[k].index(d)] for k in cube.dims if d in cube.dims[k])
                ds = f.create_dataset(d, shape=(n,), dtype='f4')
            # netCDF-C / h5netcdf append the length, formatted '%10d', to the NAME of a coordinate-less dimension
            ds.make_scale(d if d in cube.coords else _NOT_A_VARIABLE + '%10d' % ds.shape[0])
            ds.attrs['_Netcdf4Dimid'] = np.int32(i)
            if ds.dtype.kind == 'f' and d in cube.coords:
                ds.attrs['_FillValue'] = np.array([np.nan], ds.dtype)
            for k, v in cube.coord_attrs.get(d, {}).items():
                ds.attrs[k] = _attr_out(v)
        for name, arr in cube.data_vars.items():
            arr = np.asarray(arr)
            dims = cube.dims[name]
            if arr.ndim != len(dims):
                raise ValueError(f'variable {name!r}: {arr.ndim} axes but dimensions {dims}')
            ds = f.create_dataset(name, data=arr)   # complex -> compound {r, i}
            for ax, d in enumerate(dims):
                ds.dims[ax].attach_scale(f[d])
            if arr.dtype.kind == 'f':
                ds.attrs['_FillValue'] = np.array([np.nan], arr.dtype)
            for k, v in cube.var_attrs.get(name, {}).items():
                ds.attrs[k] = _attr_out(v)
        for k, v in cube.attrs.items():
            f.attrs[k] = _attr_out(v)
    return path


def _open_nc_h5py(path):
    import h5py
    data, dims, coords, var_attrs, coord_attrs = {}, {}, {}, {}, {}
    with h5py.File(path, 'r') as f:
        scales = [name for name, ds in f.items()
                  if isinstance(ds, h5py.Dataset) and _attr_in(ds.attrs.get('CLASS', b'')) == 'DIMENSION_SCALE']
        order = sorted((int(np.ravel(f[n].attrs.get('_Netcdf4Dimid', [1 << 30]))[0]), n) for n in scales)

        def decode(ds):
            a = ds[()]
            at = ds.attrs
            if a.dtype.kind in 'iuf' and ('scale_factor' in at or 'add_offset' in at):
                fill = at.get('_FillValue', at.get('missing_value'))
                bad = None if fill is None else a == np.ravel(fill)[0]
                a = a * np.ravel(at.get('scale_factor', [1.0]))[0] + np.ravel(at.get('add_offset', [0.0]))[0]
                if bad is not None:
                    a = np.where(bad, np.nan, a)
            elif a.dtype.kind == 'f':
                for key in ('_FillValue', 'missing_value'):
                    if key in at and np.isfinite(np.ravel(at[key])[0]):
                        a = np.where(a == np.ravel(at[key])[0], np.nan, a).astype(a.dtype)
            return a

        for _, name in order:
            ds = f[name]
            is_var = not str(_attr_in(ds.attrs.get('NAME', b''))).startswith(_NOT_A_VARIABLE)
            if is_var:
                coords[name] = decode(ds)
                coord_attrs[name] = {k: _attr_in(v) for k, v in ds.attrs.items() if k not in _NC_INTERNAL}
        for name, ds in f.items():
            if not isinstance(ds, h5py.Dataset) or name in scales:
                continue
            names = []
            for ax in range(ds.ndim):
                if len(ds.dims[ax]) > 0:
                    names.append(ds.dims[ax][0].name.lstrip('/'))
                elif ds.dims[ax].label:
                    names.append(ds.dims[ax].label)
                else:
                    names.append(f'phony_dim_{ax}')
            data[name] = decode(ds)
            dims[name] = tuple(names)
            var_attrs[name] = {k: _attr_in(v) for k, v in ds.attrs.items() if k not in _NC_INTERNAL}
        attrs = {k: _attr_in(v) for k, v in f.attrs.items() if k not in _NC_INTERNAL}
    return Cube(data, dims, coords, attrs, var_attrs, coord_attrs)
