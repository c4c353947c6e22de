"""the on-disk format of the workflow's cubes.  The reference reads and writes netCDF-4 through
xarray + h5netcdf (cube_POCS_interpolation_3D.py:231-244, 342-376, 392-405); neither exists in this image, but its conda Python
has h5py, and cube_io.py speaks the netCDF-4 HDF5 layout through h5py alone.  The checks run in that interpreter."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = "/opt/conda/bin/python3.9"


def conda_has_h5py():
    if not os.path.isfile(CONDA):
        return False
    return subprocess.run([CONDA, "-c", "import h5py, numpy, yaml"], capture_output=True).returncode == 0


@pytest.mark.skipif(not conda_has_h5py(), reason="no interpreter with h5py in this image")
def test_netcdf4_layout_round_trip_and_foreign_file(tmp_path):
    res = subprocess.run([CONDA, os.path.join(ROOT, "tests", "helpers", "nc_checks.py"), str(tmp_path)], capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0 and "NC CHECKS OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
