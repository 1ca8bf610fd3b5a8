#!/usr/bin/env python3
"""Torch-free check + timing of the fused feed-forward chain (csrc/rowchain.hip) at the C1 level-0 shape: 16 images x 4096 tokens x 320.

    python tools/gpu/rowchain_check.py [--rows 65536] [--iters 20] [--out gpurun_out/rowchain_check.json]

Per chain: the C-ABI launch on seeded numpy operands, compared on a sample of rows with a float64 numpy evaluation of the graph it
replaces (LayerNorm -> to_q -> softmax(q K^T d^-1/2) V -> to_out -> + x | LayerNorm -> GEGLU proj -> Linear -> + x), then an isolated
timing loop (HIP events).  Device memory through tools/gpu/hipmem.py (no `import torch`: a fresh box pays 1-2 minutes for it).
(Does not import oracle/: the numpy graph below is this tool's own.)
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
from math import erf

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

PKG = "stable-diffusion-webui_amd"


def ln(x, g, b, eps=1e-5):
    m = x.mean(-1, keepdims=True)
    v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + eps) * g + b


_verf = np.vectorize(erf)


def gelu(x):
    return 0.5 * x * (1 + _verf(x / np.sqrt(2)))


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--rows-per-image", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rowchain_check.json"))
    args = ap.parse_args()
    import hipmem
    _lib = importlib.import_module(f"{PKG}._lib")
    _lib.require_device()
    lib = _lib.lib
    hipmem.set_device(0)
    rng = np.random.default_rng(5)
    Cw, hidden, H, D, L, Lpad = 320, 1280, 8, 40, 77, 128
    rows, rpi = args.rows, args.rows_per_image
    B = rows // rpi
    sample = np.sort(rng.choice(rows, size=256, replace=False))
    x = rng.standard_normal((rows, Cw), dtype=np.float32).astype(np.float16)
    g = (1 + 0.1 * rng.standard_normal(Cw)).astype(np.float32)
    b = (0.1 * rng.standard_normal(Cw)).astype(np.float32)
    dx, dg, db = hipmem.DevBuf.from_numpy(x), hipmem.DevBuf.from_numpy(g), hipmem.DevBuf.from_numpy(b)
    dout = hipmem.DevBuf(rows * Cw * 2)
    e0, e1 = hipmem.Event(), hipmem.Event()
    res = {"rows": rows, "C": Cw}

    def timed(fn):
        fn(); hipmem.sync()
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        return e1.ms_since(e0) / args.iters * 1e3

    # ---- feed-forward
    w1 = (rng.standard_normal((2 * hidden, Cw)) / np.sqrt(Cw)).astype(np.float16)
    b1 = (0.1 * rng.standard_normal(2 * hidden)).astype(np.float32)
    w2 = (rng.standard_normal((Cw, hidden)) / np.sqrt(hidden)).astype(np.float16)
    b2 = (0.1 * rng.standard_normal(Cw)).astype(np.float32)
    dw1, db1, dw2, db2 = (hipmem.DevBuf.from_numpy(a) for a in (w1, b1, w2, b2))
    dpk = hipmem.DevBuf(lib.sdmi_rowchain_ff_pack_bytes(Cw, hidden))
    _lib.check(lib.sdmi_rowchain_ff_pack(dw1.ptr, db1.ptr, dw2.ptr, dpk.ptr, Cw, hidden, None), "ff_pack")

    def ff():
        _lib.check(lib.sdmi_rowchain_ff(dx.ptr, dout.ptr, dg.ptr, db.ptr, dpk.ptr, db2.ptr, rows, Cw, hidden, 1e-5, None), "rowchain_ff")
    us = timed(ff)
    out = dout.to_numpy(np.float16, (rows, Cw))[sample].astype(np.float64)
    xs = x[sample].astype(np.float64)
    n = ln(xs, g, b).astype(np.float16).astype(np.float64)
    h = n @ w1.astype(np.float64).T + b1
    G = (h[:, :hidden] * gelu(h[:, hidden:])).astype(np.float16).astype(np.float64)
    ref = xs + G @ w2.astype(np.float64).T + b2
    flops = 2.0 * rows * Cw * 3 * hidden
    res["ff"] = {"us": round(us, 1), "tflops": round(flops / us / 1e6, 1), "rel_l2": rel(out, ref), "rel_l2_delta": rel(out - xs, ref - xs),
                 "finite": bool(np.isfinite(out).all())}
    print("rowchain_ff   ", res["ff"], flush=True)

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    ok = res["ff"]["rel_l2"] < 5e-4 and res["ff"]["rel_l2_delta"] < 2e-3
    print("ok" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
