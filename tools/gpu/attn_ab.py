#!/usr/bin/env python3
"""Torch-free twin of attn_time.py: the flash-attention kernel forms at the self- and cross-attention shapes of the benchmarked jobs, one
process, settings interleaved round by round, outputs compared with the first setting (bit-identity flag + rel-L2).

    python tools/gpu/attn_ab.py base attn_occ=5 attn_kvt=96 ... [--reps 5] [--iters 10] [--shapes c1|all]

Settings are comma-separated sdmi_debug_set knobs (attn_occ, attn_kvt, attn_lds_pad, attn_fold_min_m ...); "base" = the defaults.
Device memory and events through tools/gpu/hipmem.py (ctypes over libamdhip64), numpy inputs, the C ABI entry sdmi_attention_vt — the
one the engine's UNet calls.  Isolated loops (K / V cache-resident): in-job times are 5-15 % higher.  (Does not import oracle/.)
"""
import argparse
import ctypes as C
import importlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

PKG = "stable-diffusion-webui_amd"
DEFAULTS = {"attn_occ": 15, "attn_kvt": 0, "attn_lds_pad": 0, "attn_tau": 8, "attn_fold_min_m": 1024}
# name, images B, heads H, queries N, keys M, head size D
SHAPES = {
    "c1": [("c1 self level 0", 16, 8, 4096, 4096, 40), ("c1 self level 1", 16, 8, 1024, 1024, 80), ("c1 self level 2", 16, 8, 256, 256, 160),
           ("c1 cross level 0", 16, 8, 4096, 77, 40), ("c1 cross level 1", 16, 8, 1024, 77, 80), ("c1 cross level 2", 16, 8, 256, 77, 160)],
    "all": [("c4a hires self level 0", 2, 8, 16384, 16384, 40), ("c3 sdxl self level 1", 8, 10, 4096, 4096, 64),
            ("c3 sdxl self level 2", 8, 20, 1024, 1024, 64), ("c3 sdxl cross level 1", 8, 10, 4096, 77, 64)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--shapes", default="c1", choices=("c1", "all"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "attn_ab.json"))
    args = ap.parse_args()
    import hipmem
    _lib = importlib.import_module(f"{PKG}._lib")
    _lib.require_device()
    lib = _lib.lib
    hipmem.set_device(0)
    shapes = SHAPES["c1"] + (SHAPES["all"] if args.shapes == "all" else [])
    rng = np.random.default_rng(1)

    def apply(setting):
        vals = dict(DEFAULTS)
        if setting != "base":
            for kv in setting.split(","):
                k, v = kv.split("=")
                vals[k] = int(v)
        for k, v in vals.items():
            rc = lib.sdmi_debug_set(k.encode(), int(v))
            if rc and v == DEFAULTS.get(k):
                continue
            _lib.check(rc, k)

    res = {}
    e0, e1 = hipmem.Event(), hipmem.Event()
    for name, b, h, n, m, d in shapes:
        mpad = (m + 63) // 64 * 64
        q = rng.standard_normal((b, n, h * d), dtype=np.float32).astype(np.float16)
        k = rng.standard_normal((b, m, h * d), dtype=np.float32).astype(np.float16)
        vt = np.zeros((b, h * d, mpad), dtype=np.float16)
        vt[:, :, :m] = rng.standard_normal((b, h * d, m), dtype=np.float32).astype(np.float16)
        dq, dk, dvt = hipmem.DevBuf.from_numpy(q), hipmem.DevBuf.from_numpy(k), hipmem.DevBuf.from_numpy(vt)
        dout = hipmem.DevBuf(q.nbytes)
        scale = 1.0 / float(np.sqrt(d))

        def launch():
            _lib.check(lib.sdmi_attention_vt(dq.ptr, dk.ptr, dvt.ptr, dout.ptr, b, h, n, m, d, h * d, h * d, mpad, h * d, C.c_float(scale), 0, None),
                       "attention_vt")
        flops = 4.0 * b * h * n * m * d
        times = {s: [] for s in args.settings}
        outs, failed = {}, {}
        for rep in range(args.reps):
            for s in args.settings:
                if s in failed:
                    continue
                try:
                    apply(s)
                    launch()
                    hipmem.sync()
                    if s not in outs:
                        outs[s] = dout.to_numpy(np.float16, q.shape).astype(np.float32)
                    e0.record()
                    for _ in range(args.iters):
                        launch()
                    e1.record()
                    times[s].append(e1.ms_since(e0) / args.iters * 1e3)
                except (_lib.SdmiError, RuntimeError) as ex:
                    failed[s] = str(ex)
        base = next((s for s in args.settings if s not in failed), None)
        res[name] = {}
        print(f"{name}  (B {b} H {h} N {n} M {m} D {d})")
        for s in args.settings:
            if s in failed:
                res[name][s] = {"error": failed[s]}
                print(f"    {s:36s} FAILED: {failed[s]}")
                continue
            us = min(times[s])
            a, bb = outs[s].astype(np.float64), outs[base].astype(np.float64)
            rel = float(np.linalg.norm(a - bb) / max(np.linalg.norm(bb), 1e-30))
            same = bool(np.array_equal(outs[s], outs[base]))
            res[name][s] = {"us_min": round(us, 2), "us_median": round(statistics.median(times[s]), 2), "tflops": round(flops / us / 1e6, 1),
                            "rel_l2_vs_first": rel, "identical_to_first": same, "finite": bool(np.isfinite(outs[s]).all())}
            print(f"    {s:36s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  rel-L2 vs {base}: {rel:.2e}{'  (bit-identical)' if same else ''}", flush=True)
        for buf in (dq, dk, dvt, dout):
            buf.free()
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    apply("base")


if __name__ == "__main__":
    main()
