#!/usr/bin/env python3
"""Second PCIe probe (GPU box): is the link full duplex for this process, and what does registering (page-locking in place) the
caller's arrays cost?  See tools/pcie_probe.py."""
import ctypes as C
import importlib.util, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_3d_interpolation_amd import _ffi
_ffi.lib()
_spec = importlib.util.find_spec("torch")
hip = C.CDLL(os.path.join(os.path.dirname(_spec.origin), "lib", "libamdhip64.so"), mode=C.RTLD_GLOBAL)
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
total = 4 * GIB; chunk = 128 << 20; nch = total // chunk
dbuf = C.c_void_p(); ck(hip.hipMalloc(C.byref(dbuf), total))
dbuf2 = C.c_void_p(); ck(hip.hipMalloc(C.byref(dbuf2), total))
src = np.ones(total // 8, np.complex64)
def ptr(a, off=0): return C.c_void_p(a.ctypes.data + off)
def dptr(d, off=0): return C.c_void_p(d.value + off)
def timed(label, fn, nbytes):
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{label:96s} {dt*1e3:8.1f} ms  {nbytes/dt/1e9:6.1f} GB/s", flush=True)
def par(fns):
    th = [threading.Thread(target=f) for f in fns]
    [t.start() for t in th]; [t.join() for t in th]
s1 = C.c_void_p(); ck(hip.hipStreamCreateWithFlags(C.byref(s1), 1))
s2 = C.c_void_p(); ck(hip.hipStreamCreateWithFlags(C.byref(s2), 1))
def up_async(a, st):
    for i in range(nch): ck(hip.hipMemcpyAsync(dptr(dbuf, i * chunk), ptr(a, i * chunk), chunk, H2D, st))
    ck(hip.hipStreamSynchronize(st))
def down_async(a, st):
    for i in range(nch): ck(hip.hipMemcpyAsync(ptr(a, i * chunk), dptr(dbuf2, i * chunk), chunk, D2H, st))
    ck(hip.hipStreamSynchronize(st))
ck(hip.hipMemcpy(dbuf, ptr(src), chunk, H2D))
dst = np.empty(total // 8, np.complex64)
timed("hipHostRegister of a touched 4-GiB array", lambda: ck(hip.hipHostRegister(ptr(src), total, 0)), total)
timed("hipHostRegister of a FRESH 4-GiB np.empty", lambda: ck(hip.hipHostRegister(ptr(dst), total, 0)), total)
timed("H2D async from registered memory, one stream", lambda: up_async(src, s1), total)
timed("D2H async into registered memory, one stream", lambda: down_async(dst, s2), total)
timed("H2D + D2H async, registered both, two streams, two threads", lambda: par([lambda: up_async(src, s1), lambda: down_async(dst, s2)]), 2 * total)
def both_one_thread():
    for i in range(nch):
        ck(hip.hipMemcpyAsync(dptr(dbuf, i * chunk), ptr(src, i * chunk), chunk, H2D, s1))
        ck(hip.hipMemcpyAsync(ptr(dst, i * chunk), dptr(dbuf2, i * chunk), chunk, D2H, s2))
    ck(hip.hipStreamSynchronize(s1)); ck(hip.hipStreamSynchronize(s2))
timed("H2D + D2H async, registered both, two streams, one thread issuing", both_one_thread, 2 * total)
timed("hipHostUnregister (touched source)", lambda: ck(hip.hipHostUnregister(ptr(src))), total)
timed("hipHostUnregister (result)", lambda: ck(hip.hipHostUnregister(ptr(dst))), total)
# register in 8 pieces from 8 threads (does it parallelise?)
dst2 = np.empty(total // 8, np.complex64)
def reg_piece(a, k, n): ck(hip.hipHostRegister(ptr(a, k * total // n), total // n, 0))
timed("hipHostRegister of a FRESH 4-GiB array in 8 pieces, 8 threads", lambda: par([(lambda k=k: reg_piece(dst2, k, 8)) for k in range(8)]), total)
for k in range(8): ck(hip.hipHostUnregister(ptr(dst2, k * total // 8)))
# page-locked allocations
p = C.c_void_p()
timed("hipHostMalloc of 4 GiB", lambda: ck(hip.hipHostMalloc(C.byref(p), total, 0)), total)
pin = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(total,))
timed("D2H async into hipHostMalloc memory", lambda: [ck(hip.hipMemcpyAsync(C.c_void_p(p.value + i * chunk), dptr(dbuf2, i * chunk), chunk, D2H, s2)) for i in range(nch)] and ck(hip.hipStreamSynchronize(s2)), total)
def up_and_down_pinned():
    for i in range(nch):
        ck(hip.hipMemcpyAsync(dptr(dbuf, i * chunk), ptr(src, i * chunk), chunk, H2D, s1))
        ck(hip.hipMemcpyAsync(C.c_void_p(p.value + i * chunk), dptr(dbuf2, i * chunk), chunk, D2H, s2))
    ck(hip.hipStreamSynchronize(s1)); ck(hip.hipStreamSynchronize(s2))
ck(hip.hipHostRegister(ptr(src), total, 0))
timed("H2D (registered) + D2H (hipHostMalloc) async on two streams", up_and_down_pinned, 2 * total)
