"""Run by tests/test_cube_io_netcdf.py under an interpreter that has h5py (the image's conda Python 3.9; the test interpreter has
none): the h5py-only netCDF-4 reader / writer of cube_io.py -- round trip, HDF5 structure as netCDF-4 / h5netcdf lay it out, and a
file written independently in that layout (packed variable, fill values, dimension without coordinate variable, byte strings)."""
import os
import sys

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pseudo_3d_interpolation_amd import cube_io  # noqa: E402
from pseudo_3d_interpolation_amd.functions import backends  # noqa: E402

assert backends.h5py_enabled and not backends.xarray_enabled
tmp = sys.argv[1]
rng = np.random.default_rng(0)
nf, ni, nx = 5, 6, 7
cube = cube_io.Cube(
    {'freq_env': (rng.standard_normal((nf, ni, nx)) + 1j * rng.standard_normal((nf, ni, nx))).astype(np.complex64),
     'env.real': rng.standard_normal((nf, ni, nx)).astype(np.float32),
     'fold': rng.integers(0, 3, (ni, nx)).astype(np.uint8)},
    {'freq_env': ('freq_twt', 'iline', 'xline'), 'env.real': ('freq_twt', 'iline', 'xline'), 'fold': ('iline', 'xline')},
    {'freq_twt': np.arange(nf) * 2.5, 'iline': np.arange(ni) + 100, 'xline': np.arange(nx) + 200},
    {'history': 'a;b;', 'text': 'line1\nline2', 'bin_size': np.float64(2.5), 'n': 3, 'interp_params_keys': 'niter;eps'},
    {'freq_env': {'units': 'amplitude', 'long_name': 'Envelope'}}, {'iline': {'long_name': 'inline'}, 'freq_twt': {'units': 'kHz'}})
path = os.path.join(tmp, 'cube.nc')
cube_io.save_cube(cube, path)

# ---- structure: what netCDF-4 / h5netcdf expect of an HDF5 file ----
with h5py.File(path, 'r') as f:
    assert set(f) == {'freq_env', 'env.real', 'fold', 'freq_twt', 'iline', 'xline'}
    for i, d in enumerate(('freq_twt', 'iline', 'xline')):
        s = f[d]
        assert s.attrs['CLASS'] == b'DIMENSION_SCALE' and s.attrs['NAME'] == d.encode() and int(s.attrs['_Netcdf4Dimid']) == i
        assert h5py.h5ds.is_scale(s.id)
    v = f['freq_env']
    assert v.dtype == np.complex64 and v.dtype.names is None   # h5py maps the {r, i} compound back to complex
    assert v.id.get_type().get_member_name(0) == b'r' and v.id.get_type().get_member_name(1) == b'i'
    assert [v.dims[a][0].name for a in range(3)] == ['/freq_twt', '/iline', '/xline']
    assert [f['fold'].dims[a][0].name for a in range(2)] == ['/iline', '/xline'] and f['fold'].dtype == np.uint8
    assert np.isnan(f['env.real'].attrs['_FillValue'][0]) and '_FillValue' not in f['fold'].attrs
    assert f.attrs['history'] == 'a;b;' and f.attrs['bin_size'].shape == (1,) and f['freq_env'].attrs['units'] == 'amplitude'

# ---- round trip ----
back = cube_io.open_cube(path)
assert back.dims == cube.dims and set(back.coords) == set(cube.coords)
for k in cube.data_vars:
    assert back.data_vars[k].dtype == cube.data_vars[k].dtype and np.array_equal(back.data_vars[k], cube.data_vars[k]), k
for k in cube.coords:
    assert np.array_equal(back.coords[k], cube.coords[k])
assert back.attrs == {'history': 'a;b;', 'text': 'line1\nline2', 'bin_size': 2.5, 'n': 3, 'interp_params_keys': 'niter;eps'}
assert back.var_attrs['freq_env'] == {'units': 'amplitude', 'long_name': 'Envelope'} and back.coord_attrs['iline'] == {'long_name': 'inline'}
assert back.slice_dim() == 'freq_twt'

# ---- a file somebody else wrote in the netCDF-4 layout ----
other = os.path.join(tmp, 'foreign.nc')
with h5py.File(other, 'w') as f:
    f.attrs['_NCProperties'] = b'version=2,netcdf=4.8.1,hdf5=1.12.1'
    f.attrs['title'] = np.bytes_('binned cube')                       # fixed-length byte string, as the C library writes them
    tw = f.create_dataset('twt', data=np.linspace(0.0, 1.0, 4)); tw.make_scale('twt'); tw.attrs['_Netcdf4Dimid'] = np.int32(0)
    il = f.create_dataset('iline', data=np.arange(3, dtype=np.int32)); il.make_scale('iline'); il.attrs['_Netcdf4Dimid'] = np.int32(1)
    xl = f.create_dataset('xline', shape=(2,), dtype='f4')            # dimension without coordinate values
    xl.make_scale('This is a netCDF dimension but not a netCDF variable.%10d' % 2); xl.attrs['_Netcdf4Dimid'] = np.int32(2)   # as netCDF-C / h5netcdf write it
    raw = np.array([[[10, 20], [30, -999], [50, 60]]] * 4, dtype=np.int16)
    env = f.create_dataset('env', data=raw)
    for a, s in enumerate((tw, il, xl)):
        env.dims[a].attach_scale(s)
    env.attrs['scale_factor'] = np.array([0.5]); env.attrs['add_offset'] = np.array([1.0]); env.attrs['_FillValue'] = np.array([-999], np.int16)
    env.attrs['units'] = np.array([b'mV'], dtype='S2')
    fl = f.create_dataset('amp', data=np.array([[1.0, 9.96921e36], [2.0, 3.0], [4.0, 5.0]], np.float32))
    fl.dims[0].attach_scale(il); fl.dims[1].attach_scale(xl); fl.attrs['_FillValue'] = np.array([9.96921e36], np.float32)
got = cube_io.open_cube(other)
assert got.dims == {'env': ('twt', 'iline', 'xline'), 'amp': ('iline', 'xline')} and set(got.coords) == {'twt', 'iline'}
assert got.attrs == {'title': 'binned cube'} and got.var_attrs['env'] == {'units': 'mV'}
want = raw * 0.5 + 1.0
want[:, 1, 1] = np.nan
assert np.array_equal(got.data_vars['env'], want, equal_nan=True)
assert np.isnan(got.data_vars['amp'][0, 1]) and got.data_vars['amp'].dtype == np.float32 and got.data_vars['amp'][2, 1] == 5.0

# ---- a cube with a coordinate-less dimension: written with the length suffix, read back as a dimension (not as fill values) ----
nocoord = cube_io.Cube({'amp': np.arange(6, dtype=np.float32).reshape(2, 3)}, {'amp': ('iline', 'xline')}, {'iline': np.arange(2)}, {}, {}, {})
p3 = os.path.join(tmp, 'nocoord.nc')
cube_io.save_cube(nocoord, p3)
with h5py.File(p3, 'r') as f:
    assert f['xline'].attrs['NAME'] == b'This is a netCDF dimension but not a netCDF variable.         3'
again = cube_io.open_cube(p3)
assert set(again.coords) == {'iline'} and again.dims['amp'] == ('iline', 'xline') and np.array_equal(again.data_vars['amp'], nocoord.data_vars['amp'])
with h5py.File(p3, 'r+') as f:     # the bare text (what round 2 wrote) is still recognised
    f['xline'].attrs['NAME'] = np.bytes_('This is a netCDF dimension but not a netCDF variable.')
assert set(cube_io.open_cube(p3).coords) == {'iline'}
print('NC CHECKS OK')
