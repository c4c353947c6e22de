"""Installable form of the package (the reference ships setup.py + pyproject.toml + setup.cfg the same way; its console scripts
`12_cube_apply_FFT`, `13_cube_interpolate_POCS`, `14_cube_apply_IFFT` -- /root/reference/setup.cfg:93-95 -- keep their names).

The product directory is `pseudo-3d-interpolation_amd/` (not an importable name): `package_dir` installs it as
`pseudo_3d_interpolation_amd`.  `libp3d_hip.so` (built by `make -C pseudo-3d-interpolation_amd/csrc`, hipcc for gfx950) and
`wavelets.json` travel as package data; `build_py` runs that make first when hipcc is there."""
import os
import shutil
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

PKG = "pseudo_3d_interpolation_amd"
SRC = "pseudo-3d-interpolation_amd"
HERE = os.path.dirname(os.path.abspath(__file__))


class build_py_with_hip(build_py):
    def run(self):
        csrc = os.path.join(HERE, SRC, "csrc")
        lib = os.path.join(HERE, SRC, "libp3d_hip.so")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if os.path.isfile(hipcc) or shutil.which("hipcc"):
            subprocess.check_call(["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1))])
        elif not os.path.isfile(lib):
            raise RuntimeError(f"{lib} is missing and hipcc was not found: build it on a ROCm machine (make -C {csrc}); there is no CPU fallback")
        super().run()


setup(
    packages=[PKG, PKG + ".functions"],
    package_dir={PKG: SRC},
    package_data={PKG: ["libp3d_hip.so", "wavelets.json"]},
    include_package_data=False,
    zip_safe=False,
    cmdclass={"build_py": build_py_with_hip},
)
