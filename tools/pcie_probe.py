#!/usr/bin/env python3
"""What the PCIe link gives a process that moves a 4-GiB NumPy cube to the GPU and another one back (GPU box), by HOW it is asked:
pageable hipMemcpy in chunks from one thread, from several threads at once, uploads and downloads at the same time, registered
(hipHostRegister) memory, page-locked staging with a parallel host copy, a fresh result array against a touched one.  Feeds the
design of functions.POCS.pocs_cube's chunk pipeline."""
import ctypes as C
import os, sys, threading, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_3d_interpolation_amd import _ffi
_ffi.lib()
import importlib.util
_spec = importlib.util.find_spec("torch")
hip = C.CDLL(os.path.join(os.path.dirname(_spec.origin), "lib", "libamdhip64.so"), mode=C.RTLD_GLOBAL)   # (already mapped by _ffi: the same runtime)
for name, args in (("hipMalloc", [C.POINTER(C.c_void_p), C.c_size_t]), ("hipMemcpy", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
                   ("hipHostRegister", [C.c_void_p, C.c_size_t, C.c_uint]), ("hipHostUnregister", [C.c_void_p]),
                   ("hipHostMalloc", [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]), ("hipDeviceSynchronize", []),
                   ("hipMemcpyAsync", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
                   ("hipStreamCreateWithFlags", [C.POINTER(C.c_void_p), C.c_uint]), ("hipStreamSynchronize", [C.c_void_p])):
    f = getattr(hip, name); f.argtypes = args; f.restype = C.c_int
H2D, D2H = 1, 2
def ck(e):
    if e != 0: raise RuntimeError(f"hip error {e}")

GIB = 1 << 30
total = 4 * GIB
chunk = 128 << 20
nch = total // chunk
dbuf = C.c_void_p(); ck(hip.hipMalloc(C.byref(dbuf), total))
dbuf2 = C.c_void_p(); ck(hip.hipMalloc(C.byref(dbuf2), total))
src = np.ones(total // 8, np.complex64)          # touched pages
def ptr(a, off=0): return C.c_void_p(a.ctypes.data + off)
def dptr(d, off=0): return C.c_void_p(d.value + off)

def timed(label, fn, nbytes):
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{label:86s} {dt*1e3:8.1f} ms  {nbytes/dt/1e9:6.1f} GB/s", flush=True)
    return dt

def up_chunks(lo, hi, arr=None):
    a = src if arr is None else arr
    for i in range(lo, hi): ck(hip.hipMemcpy(dptr(dbuf, i * chunk), ptr(a, i * chunk), chunk, H2D))
def down_chunks(lo, hi, dst):
    for i in range(lo, hi): ck(hip.hipMemcpy(ptr(dst, i * chunk), dptr(dbuf2, i * chunk), chunk, D2H))
def par(fns):
    th = [threading.Thread(target=f) for f in fns]
    [t.start() for t in th]; [t.join() for t in th]

up_chunks(0, 2)
timed("H2D pageable, one thread, 32 chunks of 128 MiB", lambda: up_chunks(0, nch), total)
timed("H2D pageable, one thread, again (same pages)", lambda: up_chunks(0, nch), total)
timed("H2D pageable, one call of 4 GiB", lambda: ck(hip.hipMemcpy(dbuf, ptr(src), total, H2D)), total)
timed("H2D pageable, 2 threads x 16 chunks", lambda: par([lambda: up_chunks(0, nch // 2), lambda: up_chunks(nch // 2, nch)]), total)
timed("H2D pageable, 4 threads x 8 chunks", lambda: par([(lambda k=k: up_chunks(k * nch // 4, (k + 1) * nch // 4)) for k in range(4)]), total)
fresh = np.empty(total // 8, np.complex64)
timed("D2H into a FRESH np.empty, one thread, 32 chunks", lambda: down_chunks(0, nch, fresh), total)
timed("D2H into the same array again (touched)", lambda: down_chunks(0, nch, fresh), total)
timed("D2H touched, 4 threads x 8 chunks", lambda: par([(lambda k=k: down_chunks(k * nch // 4, (k + 1) * nch // 4, fresh)) for k in range(4)]), total)
timed("H2D + D2H at the same time, one thread each (touched)", lambda: par([lambda: up_chunks(0, nch), lambda: down_chunks(0, nch, fresh)]), 2 * total)
fresh2 = np.empty(total // 8, np.complex64)
timed("H2D + D2H at the same time, D2H into a FRESH array", lambda: par([lambda: up_chunks(0, nch), lambda: down_chunks(0, nch, fresh2)]), 2 * total)
del fresh2
# pre-faulting a fresh array with several threads
fresh3 = np.empty(total // 8, np.complex64)
def touch(a, k, n):
    v = a.view(np.uint8).reshape(-1)
    lo, hi = k * v.size // n, (k + 1) * v.size // n
    v[lo:hi:4096] = 0
timed("pre-fault a fresh 4-GiB array, 8 threads, one byte per 4-KiB page", lambda: par([(lambda k=k: touch(fresh3, k, 8)) for k in range(8)]), total)
timed("D2H into the pre-faulted array, one thread", lambda: down_chunks(0, nch, fresh3), total)
del fresh3
# registered memory
t = timed("hipHostRegister of the 4-GiB source", lambda: ck(hip.hipHostRegister(ptr(src), total, 0)), total)
timed("H2D from the registered source, one thread, 32 chunks", lambda: up_chunks(0, nch), total)
ck(hip.hipHostUnregister(ptr(src)))
# page-locked staging: parallel host copy into a pinned chunk buffer, async H2D on a stream, double buffered
st = C.c_void_p(); ck(hip.hipStreamCreateWithFlags(C.byref(st), 1))
pins = []
for _ in range(2):
    p = C.c_void_p(); ck(hip.hipHostMalloc(C.byref(p), chunk, 0)); pins.append(p)
pin_np = [np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(chunk,)) for p in pins]
srcb = src.view(np.uint8).reshape(-1)
def staged_up(threads):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as pool:
        for i in range(nch):
            b = i & 1
            if i >= 2: ck(hip.hipStreamSynchronize(st))     # (coarse: waits for everything issued so far)
            cuts = [chunk * k // threads for k in range(threads + 1)]
            list(pool.map(lambda k: np.copyto(pin_np[b][cuts[k]:cuts[k + 1]], srcb[i * chunk + cuts[k]: i * chunk + cuts[k + 1]]), range(threads)))
            ck(hip.hipMemcpyAsync(dptr(dbuf, i * chunk), pins[b], chunk, H2D, st))
        ck(hip.hipStreamSynchronize(st))
timed("H2D through 2 page-locked 128-MiB buffers, 8-thread host copy", lambda: staged_up(8), total)
timed("H2D through 2 page-locked 128-MiB buffers, 4-thread host copy", lambda: staged_up(4), total)
