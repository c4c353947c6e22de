"""`pip install .` yields the package under its importable name and the three console scripts of the reference
(/root/reference/setup.cfg:93-95) with the reference's flags (cube_POCS_interpolation_3D.py:68-84, cube_apply_FFT.py:24-45,
cube_apply_IFFT.py:20-32).  CPU only: nothing is computed, the library is loaded and its ABI version read."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

FLAGS = {
    "12_cube_apply_FFT": ["path_cube", "--params_netcdf", "--prefix", "--compute_real", "--upsampling-factor", "--filter", "--filter_freqs",
                          "--drop-filtered-freq", "--verbose"],
    "13_cube_interpolate_POCS": ["path_cube", "--path_pocs_parameter", "--path_output_dir", "--verbose"],
    "14_cube_apply_IFFT": ["path_cube", "--params_netcdf", "--compute_real", "--rescale-envelope", "--verbose"],
}


@pytest.fixture(scope="module")
def prefix(tmp_path_factory):
    if not os.path.isfile(os.path.join(ROOT, "pseudo-3d-interpolation_amd", "libp3d_hip.so")):
        pytest.skip("libp3d_hip.so not built")
    dest = tmp_path_factory.mktemp("prefix")
    made = [p for p in ("build", "pseudo_3d_interpolation_amd.egg-info") if not os.path.exists(os.path.join(ROOT, p))]
    res = subprocess.run([sys.executable, "-m", "pip", "install", "--no-deps", "--no-build-isolation", "--no-index", "--prefix", str(dest), "."],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    for p in made:     # pip builds in the tree: leave it as it was
        shutil.rmtree(os.path.join(ROOT, p), ignore_errors=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    site = [os.path.dirname(p) for p in glob.glob(os.path.join(str(dest), "**", "pseudo_3d_interpolation_amd"), recursive=True)
            if os.path.isdir(p)]
    assert len(site) == 1, site
    bindir = [os.path.dirname(p) for p in glob.glob(os.path.join(str(dest), "**", "13_cube_interpolate_POCS"), recursive=True)]
    assert len(bindir) == 1, bindir
    return dest, site[0], bindir[0]


def test_installed_package_holds_the_library_and_the_wavelet_table(prefix):
    dest, site, _ = prefix
    pkg = os.path.join(site, "pseudo_3d_interpolation_amd")
    for name in ("libp3d_hip.so", "wavelets.json", "_ffi.py", os.path.join("functions", "POCS.py"), "cube_POCS_interpolation_3D.py"):
        assert os.path.isfile(os.path.join(pkg, name)), name
    code = ("import pseudo_3d_interpolation_amd as p, os; from pseudo_3d_interpolation_amd import _ffi; "
            "from pseudo_3d_interpolation_amd.functions.POCS import POCS, FPOCS, APOCS, POCS_algorithm, pocs_cube; "
            "assert os.path.dirname(_ffi.LIB_PATH) == os.path.dirname(p.__file__); print(_ffi.lib().p3d_abi_version())")
    res = subprocess.run([sys.executable, "-c", code], cwd=str(dest), env=dict(os.environ, PYTHONPATH=site), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip() == "1", res.stderr[-2000:]


@pytest.mark.parametrize("script", sorted(FLAGS))
def test_console_scripts_show_the_flags_of_the_reference(prefix, script):
    dest, site, bindir = prefix
    res = subprocess.run([os.path.join(bindir, script), "--help"], cwd=str(dest), env=dict(os.environ, PYTHONPATH=site), capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    for flag in FLAGS[script]:
        assert flag in res.stdout, (script, flag, res.stdout)
