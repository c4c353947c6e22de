#!/opt/conda/bin/python3.9
"""Golden vectors for the pure-NumPy helpers either side of the POCS path, produced by the REFERENCE's own functions.

Run in the build container only (the reference does not travel):

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_helpers.py

Why this needs a trick, and what the trick may and may not do.  ``cube_postprocessing_3D.py``, ``cube_apply_FFT.py`` and
``cube_POCS_interpolation_3D.py`` import ``xarray`` (and ``xrft``) at module level; no interpreter of this image has them, so the
modules cannot be imported as they are -- although the helpers recorded here take and return plain ``numpy.ndarray``.  The conda
interpreter has everything else they import (NumPy 1.26, SciPy 1.7.1, dask, PyYAML).  This script therefore registers INERT
stand-in modules under the two missing names before importing the reference modules:

  * a stand-in module has NO attributes except the class names the reference mentions in def-time annotations
    (``xr.DataArray``, ``xr.Dataset``) and ``set_options``, which the reference calls once at import time with a metadata
    switch (accepted and dropped): any other attribute access raises ``AttributeError``;
  * those names are bound to ``Inert``, a class that cannot be instantiated, called, indexed or used in arithmetic: every
    such use raises ``InertUse``.  So no stand-in object ever exists, let alone takes part in a computation; every number
    written below was computed by the reference's own statements on NumPy arrays with NumPy / SciPy.

One reference helper, ``get_freq_filter_win`` (cube_apply_FFT.py:72-143), computes its window with NumPy and then wraps it
into an ``xr.DataArray`` in its last statement.  That constructor call raises ``InertUse`` here; the window is read from the
local variable ``filter_window`` of the reference function's own frame (the traceback holds it) -- the value the reference had
computed, untouched by this script.

Fixtures are data: inputs, keyword arguments and outputs.  No source text of the reference is stored.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class InertUse(RuntimeError):
    pass


class _InertMeta(type):
    def __call__(cls, *a, **k):
        raise InertUse(f"stand-in {cls.__name__} used: the class exists for def-time annotations only")

    def __getitem__(cls, item):
        raise InertUse("stand-in indexed")

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise InertUse(f"stand-in attribute {name!r} used")


class Inert(metaclass=_InertMeta):
    pass


def _discard_options(**kwargs):
    """``xr.set_options(keep_attrs=True)`` at the top of cube_apply_FFT.py / cube_apply_IFFT.py (a display / metadata setting of a package that
    is not there): accepted and dropped -- returns nothing, touches nothing."""
    return None


class _InertModule(types.ModuleType):
    _ALLOWED = {"DataArray": Inert, "Dataset": Inert, "set_options": _discard_options}

    def __getattr__(self, name):
        if name in self._ALLOWED:
            return self._ALLOWED[name]
        raise AttributeError(f"inert stand-in for {self.__name__!r}: attribute {name!r} does not exist (the package is not installed)")


def install_stand_ins():
    for name in ("xarray", "xrft"):
        if name in sys.modules:
            raise SystemExit(f"{name} is importable here: import the reference directly instead of using stand-ins")
        try:
            __import__(name)
        except ImportError:
            sys.modules[name] = _InertModule(name)
        else:
            raise SystemExit(f"{name} is installed: no stand-in needed")


def frame_local(exc, func_name, var):
    """Value of local variable `var` in the frame of reference function `func_name` on the traceback of `exc`."""
    tb = exc.__traceback__
    while tb is not None:
        if tb.tb_frame.f_code.co_name == func_name and var in tb.tb_frame.f_locals:
            return tb.tb_frame.f_locals[var]
        tb = tb.tb_next
    raise KeyError(f"{func_name}.{var} not on the traceback")


def main():
    install_stand_ins()
    from pseudo_3D_interpolation import cube_apply_FFT as ref_fft
    from pseudo_3D_interpolation import cube_POCS_interpolation_3D as ref_pocs
    from pseudo_3D_interpolation import cube_postprocessing_3D as ref_post
    from pseudo_3D_interpolation.functions import utils as ref_utils

    out = {}
    meta = {"numpy": np.__version__, "python": sys.version.split()[0], "cases": {}}
    import scipy
    meta["scipy"] = scipy.__version__

    def put(key, arr):
        out[key] = np.asarray(arr)

    rng = np.random.default_rng(20241005)

    # ---- functions/utils.py:413-441 rescale -------------------------------------------------------------------------------
    cases = []
    for i, (shape, kw) in enumerate([((7, 5), {}), ((16,), dict(vmin=-2.0, vmax=3.5)), ((6, 6), dict(vmin=None, vmax=None)),
                                     ((4, 3), dict(vmin=1e-3, vmax=1))]):
        a = rng.standard_normal(shape)
        if i == 0:
            a[1, 2] = np.nan   # nanmin / nanmax
        put(f"rescale_{i}_in", a)
        put(f"rescale_{i}_out", ref_utils.rescale(a, **kw))
        cases.append({"kwargs": {k: (None if v is None else float(v)) for k, v in kw.items()}})
    const = np.full((3, 3), 2.5)
    put("rescale_const_in", const)
    put("rescale_const_out", ref_utils.rescale(const))
    meta["cases"]["rescale"] = cases

    # ---- cube_postprocessing_3D.py:127-176 gaussian_kernel_2d -------------------------------------------------------------
    cases = []
    for i, kw in enumerate([dict(), dict(sigma=3), dict(sigma=2, n=9), dict(sigma=2, n=(6, 11)), dict(sigma=4, normalized=False),
                            dict(sigma=3, orientation="iline"), dict(sigma=3, orientation="xline")]):
        put(f"gk2d_{i}", ref_post.gaussian_kernel_2d(**kw))
        cases.append({"kwargs": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}})
    meta["cases"]["gaussian_kernel_2d"] = cases

    # ---- cube_postprocessing_3D.py:179-260 remove_acquisition_footprint ---------------------------------------------------
    cases = []
    for i, (shape, kw) in enumerate([((64, 48), dict()), ((50, 70), dict(sigma=3, direction="iline")),
                                     ((40, 40), dict(sigma=2, direction="xline", buffer_center=0.3, buffer_filter=2)),
                                     ((33, 57), dict(sigma=3, direction="twt")), ((57, 33), dict(sigma=3, direction="twt")),
                                     ((48, 64), dict(sigma=4, direction="iline", dims=("xline", "iline")))]):
        data = rng.standard_normal(shape).astype(np.float32)
        filt, ffilter = ref_post.remove_acquisition_footprint(data, return_filter=True, verbose=0, **kw)
        put(f"footprint_{i}_in", data)
        put(f"footprint_{i}_out", filt)
        put(f"footprint_{i}_filter", ffilter)
        cases.append({"kwargs": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}})
    meta["cases"]["remove_acquisition_footprint"] = cases

    # ---- cube_postprocessing_3D.py:263-347 spatial_antialiasing -------------------------------------------------------------
    cases = []
    for i, (shape, direction, fac, kw) in enumerate([((96, 64), "iline", {"iline": 4, "xline": 1}, dict(sigma=3)),
                                                      ((64, 96), "xline", {"iline": 1, "xline": 2}, dict(sigma=2)),
                                                      ((80, 80), "iline", {"iline": 3, "xline": 1}, dict())]):
        data = rng.standard_normal(shape).astype(np.float32)
        filt, ffilter = ref_post.spatial_antialiasing(data, direction, dict(fac), return_filter=True, verbose=0, **kw)
        put(f"antialias_{i}_in", data)
        put(f"antialias_{i}_out", filt)
        put(f"antialias_{i}_filter", ffilter)
        cases.append({"direction": direction, "factors": fac, "kwargs": kw})
    try:
        ref_post.spatial_antialiasing(np.zeros((8, 8)), "iline", {"a": 1, "b": 2}, verbose=0)
        meta["antialias_bad_keys"] = None
    except ValueError as exc:
        meta["antialias_bad_keys"] = type(exc).__name__
    meta["cases"]["spatial_antialiasing"] = cases

    # ---- cube_postprocessing_3D.py:88-124 smoothing_filter ------------------------------------------------------------------
    cases = []
    for i, (shape, name, kwf, resc, kwr) in enumerate([
            ((40, 36), "gaussian", dict(sigma=1), False, None), ((40, 36), "gaussian", dict(sigma=2.5, truncate=3.0), False, None),
            ((37, 41), "median", dict(size=3), False, None), ((37, 41), "median", dict(size=5), False, None),
            ((24, 30), "median", dict(size=7), False, None),
            ((40, 36), "gaussian", dict(sigma=1.5), True, dict(vminmax=[2, 98])), ((30, 30), "median", dict(size=3), True, dict(vminmax=[95, 5]))]):
        x = rng.standard_normal(shape).astype(np.float32)
        put(f"smooth_{i}_in", x)
        put(f"smooth_{i}_out", ref_post.smoothing_filter(x, name, dict(kwf), resc, None if kwr is None else dict(kwr)))
        cases.append({"filter_name": name, "kwargs_filter": kwf, "rescale_slice": resc, "kwargs_rescale": kwr})
    meta["cases"]["smoothing_filter"] = cases

    # ---- cube_apply_FFT.py:49-69 _get_stopband / _get_const_values ----------------------------------------------------------
    cases = []
    for n in (0, 1, 2, 3, 7, 8, 25):
        for kind in ("highpass", "lowpass"):
            put(f"stopband_{n}_{kind}", ref_fft._get_stopband(n, kind))
            cases.append({"nstopband": n, "kind": kind})
    meta["cases"]["_get_stopband"] = cases
    meta["const_values"] = {k: list(ref_fft._get_const_values(k)) for k in ("highpass", "lowpass", "bandpass")}

    # ---- cube_apply_FFT.py:72-143 get_freq_filter_win (window harvested from the reference's own frame, see the docstring) ---
    cases = []
    for i, (nf, df, ftype, freqs) in enumerate([(129, 0.05, "lowpass", [2.0, 3.0]), (129, 0.05, "highpass", [0.5, 1.5]),
                                                (200, 0.025, "bandpass", [0.4, 0.9, 3.0, 4.0]), (64, 0.1, "lowpass", [3.05, 1.95]),
                                                (101, 0.05, "bandpass", [4.0, 0.25, 3.0, 1.0]), (128, 0.05, "highpass", [-1.0, 0.4])]):
        frequencies = np.arange(nf) * df if ftype != "highpass" or i != 5 else np.fft.fftfreq(nf, 1.0 / (nf * df))
        try:
            ref_fft.get_freq_filter_win(list(freqs), frequencies, dim="freq_twt", filter_type=ftype)
        except InertUse as exc:
            win = frame_local(exc, "get_freq_filter_win", "filter_window")
        else:
            raise SystemExit("get_freq_filter_win returned without touching xarray?")
        put(f"fwin_{i}_freqs", frequencies)
        put(f"fwin_{i}_win", win)
        cases.append({"filter_freqs": list(freqs), "filter_type": ftype})
    meta["cases"]["get_freq_filter_win"] = cases

    # ---- cube_apply_FFT.py:146-181 get_freq_filter_mask (a mapping of plain arrays stands for the DataArray: `da[dim]` only) ---
    cases = []
    freq_axis = np.arange(64) * 0.1
    put("fmask_freqs", freq_axis)
    for i, (ftype, freqs) in enumerate([("lowpass", [2.0, 3.0]), ("highpass", [1.5, 0.5]), ("bandpass", [0.4, 0.9, 3.0, 4.0]),
                                        ("bandpass", [4.0, 3.0, 0.9, 0.4])]):
        put(f"fmask_{i}", ref_fft.get_freq_filter_mask({"freq_twt": freq_axis}, "freq_twt", list(freqs), ftype))
        cases.append({"filter_type": ftype, "freqs": list(freqs)})
    meta["cases"]["get_freq_filter_mask"] = cases

    # ---- cube_POCS_interpolation_3D.py:146-157 create_file_path -------------------------------------------------------------
    cases = []
    for coord, kw in [([0.25, 0.5, 0.75], dict(prefix="pocs", root_path="/tmp/out")), ([12.0], dict(prefix="x", root_path=".")),
                      (3.125, dict(prefix="a_b", root_path="rel/dir")), ([100.5, 250.25], dict(prefix="p", root_path="/r", dim="freq_twt"))]:
        dim = kw.get("dim", "twt")
        ds = {dim: types.SimpleNamespace(data=np.asarray(coord, dtype=np.float64))}
        cases.append({"coord": coord, "kwargs": kw, "path": ref_pocs.create_file_path(ds, **kw)})
    meta["cases"]["create_file_path"] = cases

    # ---- cube_POCS_interpolation_3D.py:177-195 combine_runtime_results ------------------------------------------------------
    with tempfile.TemporaryDirectory() as tmp:
        lines = {"slice-0000.out": "3;0.012;1.0;0.5;0.25\n", "slice-0001.out": "4;0.015;1.0;0.5;0.25;0.125\n", "other.txt": "not me\n"}
        for name, text in lines.items():
            with open(os.path.join(tmp, name), "w", newline="\n") as fh:
                fh.write(text)
        ref_pocs.combine_runtime_results(tmp, prefix="pre", fsuffix="out")
        produced = sorted(f for f in os.listdir(tmp) if f not in lines)
        with open(os.path.join(tmp, produced[0])) as fh:
            combined = fh.read()
    meta["combine_runtime_results"] = {"inputs": lines, "prefix": "pre", "fsuffix": "out", "created": produced,
                                       "content_lines_sorted": sorted(combined.splitlines())}

    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
    with open(os.path.join(HERE, "helpers.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print(f"wrote {len(out)} arrays to helpers.npz; cases:", {k: len(v) for k, v in meta["cases"].items()})


if __name__ == "__main__":
    main()
