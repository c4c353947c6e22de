"""MI355X-native POCS hot path of fwrnke/pseudo-3D-interpolation (hand-written HIP behind a C ABI).

Layout
------
csrc/                           HIP kernels + C ABI (include/p3d.h)  ->  libp3d_hip.so
_ffi.py                         ctypes binding of the C ABI
functions/POCS.py               mirror of pseudo_3D_interpolation/functions/POCS.py (same names / kwargs)
functions/backends.py           feature probes (reference: functions/backends.py) + ``hip_enabled``
cube_POCS_interpolation_3D.py   mirror of the step-13 driver / CLI
sharding.py                     slice sharding across one-process-per-GPU ranks
"""
__version__ = "0.1.0"
