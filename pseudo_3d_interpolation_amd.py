"""Import shim: the product package lives in the directory ``pseudo-3d-interpolation_amd/`` (a name
Python cannot import directly).  Giving this module a ``__path__`` makes it a package whose
submodules are loaded from that directory, so

    import pseudo_3d_interpolation_amd
    from pseudo_3d_interpolation_amd.functions.POCS import POCS, FPOCS, APOCS

work from the repository root exactly like ``pseudo_3D_interpolation.functions.POCS`` does for the
reference."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pseudo-3d-interpolation_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f
