"""Temporal blocking experiment: run all K iterations on groups of G slices (working set G x 8 MiB) instead of sweeping the
whole cube once per iteration, so that the work buffer of a group stays in the 256 MiB Infinity Cache between iterations."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd import _ffi
from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats
from bench import torch_slices

nil = nxl = 1024
ns = int(os.environ.get("NS", 512))
K = int(os.environ.get("K", 100))
dev = torch.device("cuda", 0)
mask = orc.synthetic_mask(nil, nxl, 0.8)
mask_t = torch.from_numpy(mask.astype(np.float32)).to(dev)
x = torch_slices(torch, nil, nxl, 0, ns, dev)
x *= mask_t
out = torch.empty_like(x)
torch.cuda.synchronize()
per = nil * nxl * 8
for G in [int(g) for g in os.environ.get("GROUPS", "512,64,32,24,16,12,8,4").split(",")]:
    plan = _ffi.Plan(nil, nxl, G, device=0)
    stats = np.concatenate([plan.stats_dev(x.data_ptr() + g * per, _ffi.P3D_C64, min(G, ns - g)) for g in range(0, ns, G)])
    tau = _schedule_from_stats(stats, nil * nxl, "exponential", K, 0.99, 1e-3, "values")
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in range(0, ns, G):
            n = min(G, ns - g)
            plan.run_dev(x.data_ptr() + g * per, _ffi.P3D_C64, mask_t.data_ptr(), tau[g:g + n], K, out.data_ptr() + g * per, n,
                         thresh_op="hard", eps=0.0, want_sums=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"G={G:4d}  {K / dt:8.1f} it/s   {dt * 1e3 / K:7.3f} ms/iteration", flush=True)
    plan.close()
