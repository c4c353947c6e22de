import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions import POCS as P
nil = nxl = 1024; n = 512
mask = orc.synthetic_mask(nil, nxl, 0.8)
base = np.stack([orc.synthetic_slice(nil, nxl, s) for s in range(8)]) * mask
x = torch.from_numpy(np.ascontiguousarray(np.tile(base, (n // 8, 1, 1)))).to("cuda:0")
host = x.cpu().numpy()
print("host flags", host.flags.c_contiguous, host.ctypes.data % 4096, "registered now?", _ffi.host_register(host)); 
try: _ffi.host_unregister(host)
except Exception as e: print("unreg", e)
kw = dict(niter=20, thresh_op="hard", thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
P.pocs_cube(host[:64], mask, **dict(kw, niter=2))
for rep in range(3):
    P._timeline = []
    t0 = time.perf_counter(); out = P.pocs_cube(host, mask, **kw); dt = time.perf_counter() - t0
    tl, P._timeline = P._timeline, None
    tl = [(w, m) for w, m in tl if w != 'setup']
    tot = {}
    for wid, marks in tl:
        for (n0, a), (n1, b) in zip(marks[:-1], marks[1:]): tot[n1] = tot.get(n1, 0.0) + (b - a)
    print(f"torch-made host array: {dt*1e3:.1f} ms", {k: round(v*1e3,1) for k,v in tot.items()}, flush=True)
    del out
host2 = np.array(host)   # a NumPy-owned copy
for rep in range(2):
    t0 = time.perf_counter(); out = P.pocs_cube(host2, mask, **kw); dt = time.perf_counter() - t0
    print(f"numpy-owned copy: {dt*1e3:.1f} ms", flush=True)
    del out
