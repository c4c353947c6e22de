"""Run by tests/test_gpu_cli.py under the interpreter that has h5py: steps 12 -> 13 -> 14 of the workflow on netCDF files.
argv: work directory (holding cube_twt.npz, netcdf.yml, pocs.yml written by the test)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as step13  # noqa: E402
from pseudo_3d_interpolation_amd import cube_apply_FFT as step12  # noqa: E402
from pseudo_3d_interpolation_amd import cube_apply_IFFT as step14  # noqa: E402
from pseudo_3d_interpolation_amd.cube_io import open_cube, save_cube  # noqa: E402

work, prefix = sys.argv[1], sys.argv[2]
nc = os.path.join(work, 'nc')
os.makedirs(nc, exist_ok=True)
save_cube(open_cube(os.path.join(work, 'cube_twt.npz')), os.path.join(nc, 'cube_twt.nc'))          # the workflow's input format
step12.main(['12_cube_apply_FFT', os.path.join(nc, 'cube_twt.nc'), '--params_netcdf', os.path.join(work, 'netcdf.yml'), '--compute_real'])
step13.main(['13_cube_interpolate_POCS', os.path.join(nc, 'cube_freq.nc'), '--path_pocs_parameter', os.path.join(work, 'pocs.yml')])
step14.main(['14_cube_apply_IFFT', os.path.join(nc, f'{prefix}.nc'), '--params_netcdf', os.path.join(work, 'netcdf.yml'), '--compute_real'])
# hand the results back in the format the calling interpreter can read
for name in ('cube_freq', prefix, f'{prefix.replace("freq", "twt")}_interp-freq'):
    save_cube(open_cube(os.path.join(nc, name + '.nc')), os.path.join(nc, name + '.back.npz'))
print('NC PIPELINE OK', sorted(os.listdir(os.path.join(nc, prefix))))
