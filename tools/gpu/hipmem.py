"""Device memory, events and the profiler buffer for the torch-free harnesses (tools/gpu/fwd_ab.py): ctypes over libamdhip64.

`import torch` costs 1-2 minutes on a fresh GPU box (the image pages in) — more than a whole kernel A/B.  The engine's C ABI takes
plain device pointers, so the tools that only drive kernels need no more than hipMalloc / hipMemcpy / hipEvent.  Test and bench
plumbing only: the product's device memory stays torch's (engine.py).
"""
import ctypes as C
import os

import numpy as np

_hip = None
for _name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):     # the soname libsdmi.so links: one runtime instance
    try:
        _hip = C.CDLL(_name, mode=getattr(os, "RTLD_GLOBAL", 0))
        break
    except OSError:
        continue
if _hip is None:
    raise ImportError("libamdhip64.so not found")

_hip.hipGetErrorString.restype = C.c_char_p
_hip.hipGetErrorString.argtypes = [C.c_int]
_hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_hip.hipFree.argtypes = [C.c_void_p]
_hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
_hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
_hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
_hip.hipEventSynchronize.argtypes = [C.c_void_p]
_hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
_hip.hipSetDevice.argtypes = [C.c_int]

H2D, D2H, D2D = 1, 2, 3


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: HIP error {rc} ({(_hip.hipGetErrorString(rc) or b'').decode()})")


def set_device(i=0):
    _ck(_hip.hipSetDevice(int(i)), "hipSetDevice")


def sync():
    _ck(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


class DevBuf:
    """One hipMalloc'ed buffer; `.ptr` is what the C ABI takes."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _ck(_hip.hipMalloc(C.byref(p), max(self.nbytes, 16)), f"hipMalloc({self.nbytes})")
        self.ptr = C.c_void_p(p.value)

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        _ck(_hip.hipMemcpy(b.ptr, C.c_void_p(a.ctypes.data), a.nbytes, H2D), "hipMemcpy H2D")
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        sync()
        _ck(_hip.hipMemcpy(C.c_void_p(out.ctypes.data), self.ptr, out.nbytes, D2H), "hipMemcpy D2H")
        return out

    def zero(self):
        _ck(_hip.hipMemset(self.ptr, 0, self.nbytes), "hipMemset")

    def free(self):
        if self.ptr is not None and self.ptr.value:
            _hip.hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self):
        e = C.c_void_p()
        _ck(_hip.hipEventCreate(C.byref(e)), "hipEventCreate")
        self.e = C.c_void_p(e.value)

    def record(self, stream=None):
        _ck(_hip.hipEventRecord(self.e, stream), "hipEventRecord")

    def ms_since(self, start):
        _ck(_hip.hipEventSynchronize(self.e), "hipEventSynchronize")
        ms = C.c_float()
        _ck(_hip.hipEventElapsedTime(C.byref(ms), start.e, self.e), "hipEventElapsedTime")
        return float(ms.value)
