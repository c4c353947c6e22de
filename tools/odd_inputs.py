#!/usr/bin/env python3
"""Unusual host inputs of pocs_cube against the plain call (GPU box): Fortran-ordered, strided and read-only cubes, memory-mapped files, bool / uint8 masks --
on a cube large enough for the page-locking chunk pipeline (P3D_PIN_MIN_MIB=1 forces it on the small cube used here)."""
import os, sys, tempfile
os.environ.setdefault("P3D_PIN_MIN_MIB", "1")
os.environ.setdefault("P3D_CHUNK_MIB", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
nil, nxl, n = 128, 256, 24
_, mask, obs = orc.synthetic_cube(nil, nxl, n, 0.6)
obs = np.ascontiguousarray(obs.astype(np.complex64))
kw = dict(niter=6, thresh_op="hard", eps=0.0, p_min=1e-2)
want = P.pocs_cube(obs, mask, batch_slices=4, **kw)
ok = True
def check(name, got):
    global ok
    same = np.array_equal(got, want)
    ok &= same
    print(f"{name:46s} {'same bits' if same else 'DIFFERS'}", flush=True)
check("plain, unchunked", P.pocs_cube(obs, mask, **kw))
check("Fortran-ordered cube", P.pocs_cube(np.asfortranarray(obs), mask, batch_slices=4, **kw))
big = np.zeros((2 * n, nil, nxl), np.complex64); big[::2] = obs
check("every other slice of a larger cube", P.pocs_cube(big[::2], mask, batch_slices=4, **kw))
ro = obs.copy(); ro.flags.writeable = False
check("read-only cube", P.pocs_cube(ro, mask, batch_slices=4, **kw))
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "cube.npy"); np.save(f, obs)
    mm = np.load(f, mmap_mode="r")
    check("memory-mapped file (read-only)", P.pocs_cube(mm, mask, batch_slices=4, **kw))
    mo = np.lib.format.open_memmap(os.path.join(d, "out.npy"), mode="w+", dtype=np.complex64, shape=obs.shape)
    check("result into a memory-mapped file", P.pocs_cube(obs, mask, out=mo, batch_slices=4, **kw))
check("bool mask", P.pocs_cube(obs, mask.astype(bool), batch_slices=4, **kw))
check("uint8 mask", P.pocs_cube(obs, mask.astype(np.uint8), batch_slices=4, **kw))
check("float64 mask", P.pocs_cube(obs, mask.astype(np.float64), batch_slices=4, **kw))
w = P.pocs_cube(obs.real.astype(np.float32), mask, transform_kind="WAVELET", wavelet="db2", niter=4, thresh_op="soft", eps=0.0, p_min=0.05)
w2 = P.pocs_cube(np.asfortranarray(obs.real.astype(np.float32)), mask.astype(bool), transform_kind="WAVELET", wavelet="db2", niter=4, thresh_op="soft", eps=0.0, p_min=0.05)
print(f"{'WAVELET: Fortran cube, bool mask':46s} {'same bits' if np.array_equal(w, w2) else 'DIFFERS'}"); ok &= np.array_equal(w, w2)
sys.exit(0 if ok else 1)
