"""Development check of the mixed-radix register engine (p3d_mix.hpp): fft2 / ifft2 hooks against NumPy for every planned length as the row
and as the column axis, then short POCS jobs against the oracle.  Run on the GPU box:  python tools/mix_check.py [n ...]"""
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pseudo_3d_interpolation_amd import _ffi  # noqa: E402
from pseudo_3d_interpolation_amd.functions import POCS as P  # noqa: E402

inc = open(os.path.join(ROOT, "pseudo-3d-interpolation_amd", "csrc", "p3d_mix_plans.inc")).read()
lengths = [int(a) for a in sys.argv[1:]] or sorted(int(m) for m in re.findall(r"^X\((\d+),", inc, re.M))
rng = np.random.default_rng(3)
bad = 0
for n in lengths:
    for shape in ((n, 40), (24, n)):
        x = (rng.standard_normal((3,) + shape) + 1j * rng.standard_normal((3,) + shape)).astype(np.complex64)
        plan = _ffi.Plan(shape[0], shape[1], 3)
        f = plan.fft2(x)
        b = plan.fft2(f, inverse=True)
        ref = np.fft.fft2(x.astype(np.complex128))
        e1 = np.linalg.norm(f - ref) / np.linalg.norm(ref)
        e2 = np.linalg.norm(b - x) / np.linalg.norm(x)
        flag = "" if (e1 < 3e-6 and e2 < 3e-6) else "   <-- BAD"
        bad += bool(flag)
        print(f"{shape}: fft2 {e1:.2e}  round trip {e2:.2e}{flag}", flush=True)
        plan.close()
if "--pocs" in sys.argv or True:
    from oracle import pocs_oracle as orc
    for shape in [(lengths[0], 36), (20, lengths[-1])] + ([(lengths[len(lengths) // 2], lengths[len(lengths) // 3])] if len(lengths) > 2 else []):
        mask = orc.synthetic_mask(shape[0], shape[1], 0.5)
        cube = (np.stack([orc.synthetic_slice(shape[0], shape[1], 7 + s) for s in range(3)]) * mask).astype(np.complex64)
        for kw in (dict(niter=8, thresh_op="soft", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-2),
                   dict(niter=30, thresh_op="soft", thresh_model="exponential", eps=1e-6, p_max=0.99, p_min=1e-2, alpha=0.8, version="adaptive")):
            res, infos = [], []
            got = P.pocs_cube(cube, mask, results=res, **kw)
            want = orc.pocs_cube(cube.astype(np.complex128), mask, infos=infos, **kw)
            err = max(np.linalg.norm(got[s] - want[s]) / np.linalg.norm(want[s]) for s in range(3))
            same = [r["niterations"] for r in res] == [i["niterations"] for i in infos]
            flag = "" if (err < 1e-5 and same) else "   <-- BAD"
            bad += bool(flag)
            print(f"pocs {shape} {kw['version'] if 'version' in kw else 'regular'}: rel-L2 {err:.2e} niter equal {same}{flag}", flush=True)
print("FAILED" if bad else "all good")
sys.exit(1 if bad else 0)
