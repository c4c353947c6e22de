"""Availability probes.  Mirrors pseudo_3D_interpolation/functions/backends.py:5-11 (same flag
names, same ``find_spec`` semantics) and adds ``hip_enabled`` for the HIP library of this package."""
import os
from importlib import util

scipy_enabled = util.find_spec('scipy') is not None
numba_enabled = util.find_spec('numba') is not None
pywt_enabled = util.find_spec('pywt') is not None
FFST_enabled = util.find_spec('FFST') is not None  # PyShearlets
curvelops_enabled = util.find_spec('curvelops') is not None
geopandas_enabled = util.find_spec('geopandas') is not None
tpxo_tide_prediction_enabled = util.find_spec('tpxo_tide_prediction') is not None
xarray_enabled = util.find_spec('xarray') is not None
h5py_enabled = util.find_spec('h5py') is not None   # netCDF-4 files without xarray: cube_io reads / writes the HDF5 layout itself

#: the compiled HIP library sits next to the package sources (built by csrc/Makefile)
hip_library_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'libp3d_hip.so')
hip_enabled = os.path.isfile(hip_library_path)
