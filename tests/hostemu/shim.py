"""TEST INFRASTRUCTURE: lets the `-m gpu` parity tests run on a machine WITHOUT a GPU against the host-emulated build of libsdmi
(tests/hostemu/: the library's own sources compiled as C++, "device" memory = host memory).

Activated by tests/conftest.py when SDMI_HOSTEMU=1 (and SDMI_LIB points at the emulated library, which stable-diffusion-webui_amd/_lib.py
then loads instead of lib/libsdmi.so).  The product package is not touched: this module only changes what the TEST PROCESS sees of torch —
  * a TorchFunctionMode that rewrites every `device=cuda...` argument to the CPU, so the host code's torch.empty(..., device=cuda) and
    x.to(cuda) hand the library host pointers;
  * torch.cuda.* stand-ins (current_stream().cuda_stream = 0, synchronize = nothing, Event = wall clock).
Nothing here computes: every number still comes out of the kernels' source, run by tests/hostemu/hip/hip_runtime.h.
"""
import contextlib
import ctypes as C
import os
import time
import types

import torch
from torch.overrides import TorchFunctionMode


def _is_cuda(d):
    if isinstance(d, torch.device):
        return d.type == "cuda"
    return isinstance(d, str) and d.startswith("cuda")


class _CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if func is torch.device:                               # torch.device("cuda", 0) itself
            return torch.device("cpu") if (args and _is_cuda(args[0])) or _is_cuda(kwargs.get("type")) else func(*args, **kwargs)
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = torch.device("cpu")
        if any(_is_cuda(a) for a in args):
            args = tuple(torch.device("cpu") if _is_cuda(a) else a for a in args)
        name = getattr(func, "__name__", "")
        if name == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        if name == "pin_memory" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        return func(*args, **kwargs)


class _Event:
    def __init__(self, enable_timing=False, **kw):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def wait(self, stream=None):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _Stream:
    cuda_stream = 0
    device = torch.device("cpu")

    def __init__(self, *a, **kw):
        pass

    def synchronize(self):
        pass

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def record_event(self, e=None):
        e = e or _Event()
        e.record()
        return e


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    path = os.environ.get("SDMI_LIB")
    if not path or not os.path.exists(path):
        raise RuntimeError("SDMI_HOSTEMU=1 needs SDMI_LIB=<the host-emulated library built by tests/conftest.py>")
    emu = C.CDLL(path)
    emu.emu_set_threaded(2)                                    # auto: sequential until a kernel synchronises, then fibers
    stream = _Stream()
    cu = torch.cuda
    cu.is_available = lambda: True
    cu.device_count = lambda: 1
    cu.current_device = lambda: 0
    cu.set_device = lambda d: None
    cu.synchronize = lambda d=None: None
    cu.current_stream = lambda d=None: stream
    cu.default_stream = lambda d=None: stream
    cu.empty_cache = lambda: None
    cu.Event = _Event
    cu.Stream = _Stream
    cu.stream = lambda s: contextlib.nullcontext()
    cu.mem_get_info = lambda d=None: (1 << 36, 1 << 37)
    cu.max_memory_allocated = lambda d=None: 0
    cu.memory_allocated = lambda d=None: 0
    cu.reset_peak_memory_stats = lambda d=None: None
    cu.manual_seed_all = lambda s: None
    cu.get_device_properties = lambda d=None: types.SimpleNamespace(name="hostemu", gcnArchName="gfx950:hostemu", total_memory=1 << 37,
                                                                    multi_processor_count=256)
    cu.get_device_name = lambda d=None: "hostemu"
    _CudaToCpu().__enter__()                                   # for the life of the process
