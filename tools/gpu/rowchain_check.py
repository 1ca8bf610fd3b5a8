#!/usr/bin/env python3
"""Torch-free check + timing of the fused row-local chains (csrc/rowchain.hip) at the C1 level-0 shape: 16 images x 4096 tokens x 320.

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
    if os.environ.get("SDMI_RC_VAR") == "10":                 # section timers of the 4-wave kernel (SDMI_RC_PARTS builds)
        nw = rows // 128 * 4
        dbg = hipmem.DevBuf(nw * 4 * 8)
        hipmem._hip.hipMemset(dbg.ptr, 0, dbg.nbytes)
        addr = dbg.ptr.value
        _lib.check(lib.sdmi_debug_set(b"rc_dbg_lo", C.c_int(addr & 0xFFFFFFFF if addr & 0xFFFFFFFF < 2 ** 31 else (addr & 0xFFFFFFFF) - 2 ** 32)), "rc_dbg_lo")
        _lib.check(lib.sdmi_debug_set(b"rc_dbg_hi", C.c_int(addr >> 32)), "rc_dbg_hi")
        ff(); hipmem.sync()
        t = dbg.to_numpy(np.int64, (nw, 4)).astype(np.float64)
        it = t[:, 3].mean()
        res["ff_sections_cycles_per_iteration"] = {"sync": float(t[:, 0].mean() / it), "stage1_geglu": float(t[:, 1].mean() / it),
                                                   "stage2": float(t[:, 2].mean() / it), "iterations": float(it),
                                                   "us_per_launch": us, "implied_GHz": float((t[:, :3].sum(1).mean()) / (us * 1e3) * (rows / 128 / 256))}
        print("sections", res["ff_sections_cycles_per_iteration"], flush=True)
        _lib.check(lib.sdmi_debug_set(b"rc_dbg_lo", 0), "rc_dbg_lo"); _lib.check(lib.sdmi_debug_set(b"rc_dbg_hi", 0), "rc_dbg_hi")
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

    # ---- cross-attention
    k = rng.standard_normal((B * L, Cw), dtype=np.float32).astype(np.float16)
    v = rng.standard_normal((B, L, Cw), dtype=np.float32).astype(np.float16)
    vt = np.zeros((B, Cw, Lpad), np.float16)
    vt[:, :, :L] = v.transpose(0, 2, 1)
    wq = (rng.standard_normal((Cw, Cw)) / np.sqrt(Cw)).astype(np.float16)
    wo = (rng.standard_normal((Cw, Cw)) / np.sqrt(Cw)).astype(np.float16)
    bo = (0.1 * rng.standard_normal(Cw)).astype(np.float32)
    dk, dvt, dwq, dwo, dbo = (hipmem.DevBuf.from_numpy(a) for a in (k, vt, wq, wo, bo))
    dxp = hipmem.DevBuf(lib.sdmi_rowchain_xattn_pack_bytes(Cw, B, H))
    scale = D ** -0.5

    def pack():
        _lib.check(lib.sdmi_rowchain_xattn_pack(dk.ptr, dvt.ptr, dwq.ptr, dwo.ptr, dxp.ptr, Cw, B, L, Lpad, H, scale, None), "xattn_pack")
    us_pack = timed(pack)

    def xa():
        _lib.check(lib.sdmi_rowchain_xattn(dx.ptr, dout.ptr, dg.ptr, db.ptr, dxp.ptr, dbo.ptr, rows, rpi, Cw, H, 1e-5, None), "rowchain_xattn")
    us = timed(xa)
    out = dout.to_numpy(np.float16, (rows, Cw))[sample].astype(np.float64)
    q = ln(xs, g, b) @ wq.astype(np.float64).T
    ref = np.zeros_like(xs)
    for i, r in enumerate(sample):
        bi = r // rpi
        kk, vv = k[bi * L:(bi + 1) * L].astype(np.float64), v[bi].astype(np.float64)
        o = np.zeros(Cw)
        for hh in range(H):
            s = kk[:, hh * D:(hh + 1) * D] @ q[i, hh * D:(hh + 1) * D] * scale
            p = np.exp(s - s.max())
            p /= p.sum()
            o[hh * D:(hh + 1) * D] = p @ vv[:, hh * D:(hh + 1) * D]
        ref[i] = o @ wo.astype(np.float64).T
    ref = xs + ref + bo
    flops_unfused = 2.0 * rows * Cw * (2 * Cw) + 4.0 * rows * L * Cw
    res["xattn"] = {"us": round(us, 1), "pack_us": round(us_pack, 1), "tflops_as_run": round(2.0 * rows * Cw * 2 * 96 * H / us / 1e6, 1),
                    "tflops_of_the_graph_it_replaces": round(flops_unfused / us / 1e6, 1), "rel_l2": rel(out, ref),
                    "rel_l2_delta": rel(out - xs, ref - xs), "finite": bool(np.isfinite(out).all())}
    print("rowchain_xattn", res["xattn"], flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    ok = res["ff"]["rel_l2"] < 5e-4 and res["xattn"]["rel_l2"] < 5e-4 and res["ff"]["rel_l2_delta"] < 2e-3 and res["xattn"]["rel_l2_delta"] < 3e-3
    print("ok" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
