import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """Which HIP runtime the library bound in THIS process (the test log records it: compiled vs. runtime version and the file; -q keeps the line)."""
    try:
        from pseudo_3d_interpolation_amd import _ffi
        terminalreporter.write_line(f"hip_runtime: {_ffi.runtime_info()}")
    except Exception as exc:  # noqa: BLE001 -- not built yet
        terminalreporter.write_line(f"hip_runtime: unavailable ({type(exc).__name__})")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def parse_params(arr):
    """'key=repr' strings stored by make_golden.py -> dict."""
    import ast

    out = {}
    for item in arr:
        k, v = str(item).split("=", 1)
        out[k] = ast.literal_eval(v)
    return out


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / nb) if nb > 0 else float(np.linalg.norm(a - b))


@pytest.fixture(scope="session")
def golden_pocs():
    return load_golden("pocs.npz")
