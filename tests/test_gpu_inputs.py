"""Unusual host inputs of ``pocs_cube`` (GPU): Fortran-ordered, strided, read-only and memory-mapped cubes, a memory-mapped result array, bool / uint8 /
float64 masks -- through the page-locking chunk pipeline (forced onto a small cube by environment switches that are read at import: hence a child
process, tools/odd_inputs.py) -- give the bits of the plain call."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unusual_host_inputs_give_the_bits_of_the_plain_call():
    env = dict(os.environ, P3D_PIN_MIN_MIB="1", P3D_CHUNK_MIB="2")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "odd_inputs.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "DIFFERS" not in res.stdout and res.stdout.count("same bits") >= 10, res.stdout
