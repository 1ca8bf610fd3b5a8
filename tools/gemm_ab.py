#!/usr/bin/env python3
"""A/B driver for the GEMM kernel variants (run plain for TFLOP/s, or under `rocprofv3 --pmc ...` for SQ counters).

    python tools/gemm_ab.py [iters]

Interleaves the two-stage (gemm_pipe=0) and ping-pong (gemm_pipe=3) kernels on a few shapes of the workload, several
rounds each, so that clock / thermal drift hits both arms alike."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = importlib.import_module("stable-diffusion-webui_amd._lib")
ops = importlib.import_module("stable-diffusion-webui_amd.ops")
L = lib.lib

SHAPES = [
    ("conv3x3 320->320 @64^2 B16", dict(B=16, H=64, W=64, cin=320, cout=320, taps=9), 5),
    ("conv3x3 (640+320)->320 @64^2", dict(B=16, H=64, W=64, cin=960, cout=320, taps=9), 5),
    ("vae conv3x3 512->512 @128^2 B2", dict(B=2, H=128, W=128, cin=512, cout=512, taps=9), 4),
    ("linear ff2 1280->320 tok65536", dict(B=16, H=64, W=64, cin=1280, cout=320, taps=1), 5),
    ("linear 320->320 tok65536", dict(B=16, H=64, W=64, cin=320, cout=320, taps=1), 5),
    ("gemm 8192^3", dict(B=1, H=8192, W=1, cin=8192, cout=8192, taps=1), 4),
    ("conv3x3 640->640 @32^2 B16", dict(B=16, H=32, W=32, cin=640, cout=640, taps=9), 8),
    ("conv3x3 1280->1280 @16^2 B16", dict(B=16, H=16, W=16, cin=1280, cout=1280, taps=9), 8),
    ("linear 640->640 tok16384", dict(B=16, H=32, W=32, cin=640, cout=640, taps=1), 8),
    ("linear ff2 2560->640 tok16384", dict(B=16, H=32, W=32, cin=2560, cout=640, taps=1), 8),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    lib.require_device()
    dev = torch.device("cuda")
    only = os.environ.get("AB_ONLY")
    for name, kw, cfg in SHAPES:
        if only and only not in name:
            continue
        k = 3 if kw["taps"] == 9 else 1
        x = torch.randn(kw["B"], kw["H"], kw["W"], kw["cin"], device=dev).half()
        w = (torch.randn(kw["cout"], kw["cin"], k, k, device=dev) * (kw["cin"] * k * k) ** -0.5).half()
        if os.environ.get("AB_ZERO"):
            x.zero_(); w.zero_()
        wp = ops.pack_conv_weight(w)
        flops = 2.0 * kw["B"] * kw["H"] * kw["W"] * kw["cout"] * kw["cin"] * k * k
        lib.check(L.sdmi_debug_set(b"gemm_cfg", cfg))
        pp = 4 if cfg == 8 else 3
        arms = [(0, 0), (pp, 0)] + [(pp, int(f, 16)) for f in os.environ.get("AB_FLAGS", "").split(",") if f]
        res = {a: [] for a in arms}
        for rnd in range(3):
            for arm in arms:
                pv, fl = arm
                lib.check(L.sdmi_debug_set(b"gemm_pipe", pv)); lib.check(L.sdmi_debug_set(b"gemm_dbgflags", fl))
                for _ in range(3):
                    ops.conv_gemm(x, wp, taps=kw["taps"])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.conv_gemm(x, wp, taps=kw["taps"])
                e1.record()
                torch.cuda.synchronize()
                res[arm].append(flops / (e0.elapsed_time(e1) / iters * 1e-3) / 1e12)
        lib.check(L.sdmi_debug_set(b"gemm_pipe", -1)); lib.check(L.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(L.sdmi_debug_set(b"gemm_dbgflags", 0))
        fmt = lambda v: "/".join(f"{t:5.0f}" for t in v)
        extra = "".join(f" | pp flags {a[1]:#x} {fmt(res[a])}" for a in arms[2:])
        print(f"{name:34s} cfg{cfg} | two-stage {fmt(res[(0, 0)])} | ping-pong {fmt(res[(pp, 0)])}{extra} TF", flush=True)


if __name__ == "__main__":
    main()
