#!/usr/bin/env python3
"""Host-side probe (GPU box): what do transparent huge pages do for the first touch and the page-locking of a fresh 4-GiB result array?"""
import ctypes, mmap, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for f in ("enabled", "defrag", "shmem_enabled"):
    try:
        print(f, open(f"/sys/kernel/mm/transparent_hugepage/{f}").read().strip())
    except OSError as e:
        print(f, e)
libc = ctypes.CDLL("libc.so.6", use_errno=True)
MADV_HUGEPAGE = 14
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(8)
def touch(a, step=4096):
    flat = a.reshape(-1).view(np.uint8)
    cuts = [flat.size * i // 8 for i in range(9)]
    list(pool.map(lambda i: flat[cuts[i]:cuts[i + 1]:step].__setitem__(slice(None), 0), range(8)))
n = 4 << 30
from pseudo_3d_interpolation_amd import _ffi
_ffi.lib()
for mode in ("plain", "madvise hugepage", "plain", "madvise hugepage"):
    a = np.empty(n // 8, np.complex64)
    addr = a.ctypes.data
    if mode != "plain":
        lo = (addr + (2 << 20) - 1) & ~((2 << 20) - 1)
        rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t((addr + n - lo) & ~((2 << 20) - 1)), MADV_HUGEPAGE)
        if rc != 0: print("madvise failed", ctypes.get_errno())
    t0 = time.perf_counter(); touch(a); t1 = time.perf_counter()
    ok = _ffi.host_register(a); t2 = time.perf_counter()
    if ok: _ffi.host_unregister(a)
    anon = [l for l in open("/proc/self/smaps_rollup") if "AnonHuge" in l]
    print(f"{mode:18s}: first touch (8 threads) {1e3*(t1-t0):7.1f} ms, hipHostRegister {1e3*(t2-t1):7.1f} ms ({ok}); {anon[0].strip() if anon else ''}", flush=True)
    del a
