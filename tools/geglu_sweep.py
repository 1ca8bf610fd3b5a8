#!/usr/bin/env python3
"""Interleaved tile-config sweep for the short-K layers (GEGLU feed-forward and 1x1 projections): TFLOP/s per config."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk
lib, L = bk.lib, bk.L
SHAPES = [
    ("ff1 geglu 320->2560 tok65536", dict(B=16, H=64, W=64, cin=320, cout=2560, taps=1, geglu=True)),
    ("ff1 geglu 640->5120 tok16384", dict(B=16, H=32, W=32, cin=640, cout=5120, taps=1, geglu=True)),
    ("ff1 geglu 1280->10240 tok4096", dict(B=16, H=16, W=16, cin=1280, cout=10240, taps=1, geglu=True)),
    ("linear 320->320 tok65536", dict(B=16, H=64, W=64, cin=320, cout=320, taps=1)),
    ("linear 640->640 tok16384", dict(B=16, H=32, W=32, cin=640, cout=640, taps=1)),
    ("linear 1280->1280 tok4096", dict(B=16, H=16, W=16, cin=1280, cout=1280, taps=1)),
    ("linear qk 320->640 tok65536", dict(B=16, H=64, W=64, cin=320, cout=640, taps=1)),
]
CFGS = [(-1, "auto"), (0, "128x128"), (3, "128x128k32"), (7, "128x64"), (2, "64x64"), (4, "256x256pp"), (5, "256x320pp"), (8, "128x320")]
bn = {0: 128, 3: 128, 7: 64, 2: 64, 4: 256, 5: 320, 8: 320}
print("shape".ljust(34) + " | " + " | ".join(n.rjust(10) for _, n in CFGS))
for name, kw in SHAPES:
    cells = []
    for cfg, _ in CFGS:
        if cfg >= 0 and (kw["cout"] % bn[cfg] or (kw.get("geglu") and cfg in (5, 8))):
            cells.append("-".rjust(10)); continue
        lib.check(L.sdmi_debug_set(b"gemm_cfg", cfg))
        best = 0.0
        for _ in range(3):
            ms, tf = bk.bench_conv(iters=20, **kw)
            best = max(best, tf)
        cells.append(f"{best:10.1f}")
    lib.check(L.sdmi_debug_set(b"gemm_cfg", -1))
    print(name.ljust(34) + " | " + " | ".join(cells), flush=True)
