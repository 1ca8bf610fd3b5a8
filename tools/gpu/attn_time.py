#!/usr/bin/env python3
"""Standalone timing of the flash-attention kernel forms at the self-attention shapes of the benchmarked jobs (same box, same process,
forms interleaved round by round):   python tools/gpu/attn_time.py [--out gpurun_out/attn_time.json]"""
import argparse
import importlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "attn_time.json"))
    ap.add_argument("--variants", default="15,20,21")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--lds-pads", default="", help="occupancy probe: comma-separated extra LDS bytes per workgroup for variant 15 (e.g. 0,24000,56000,100000 = 4,2..3,1 workgroups per CU)")
    args = ap.parse_args()
    lib = importlib.import_module(f"{PKG}._lib")
    ops = importlib.import_module(f"{PKG}.ops")
    lib.require_device()
    variants = [int(v) for v in args.variants.split(",")]
    pads = [int(v) for v in args.lds_pads.split(",")] if args.lds_pads else []
    shapes = [("c1 level 0", 16, 8, 4096, 40), ("c4a hires level 0", 2, 8, 16384, 40), ("c1 level 1", 16, 8, 1024, 80),
              ("c3 sdxl level 1", 8, 10, 4096, 64), ("c3 sdxl level 2", 8, 20, 1024, 64)]
    out = {}
    for name, b, h, n, d in shapes:
        g = torch.Generator().manual_seed(1)
        q, k = torch.randn(b, n, h * d, generator=g).half().cuda(), torch.randn(b, n, h * d, generator=g).half().cuda()
        vt = torch.randn(b, h * d, n, generator=g).half().cuda()
        flops = 4.0 * b * h * n * n * d
        times = {v: [] for v in variants}
        for rep in range(args.reps):
            for v in variants:
                lib.check(lib.lib.sdmi_debug_set(b"attn_occ", v))
                ops.attention_vt(q, k, vt, h, n)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.attention_vt(q, k, vt, h, n)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 10 * 1e3)
        lib.check(lib.lib.sdmi_debug_set(b"attn_occ", 15))
        pad_times = {pd: [] for pd in pads}
        for rep in range(args.reps if pads else 0):
            for pd in pads:
                lib.check(lib.lib.sdmi_debug_set(b"attn_lds_pad", pd))
                ops.attention_vt(q, k, vt, h, n)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.attention_vt(q, k, vt, h, n)
                e1.record()
                torch.cuda.synchronize()
                pad_times[pd].append(e0.elapsed_time(e1) / 10 * 1e3)
        lib.check(lib.lib.sdmi_debug_set(b"attn_lds_pad", 0))
        out[name] = {**{f"occ15_ldspad{pd}": {"us_min": round(min(t), 1), "workgroups_per_cu_by_lds": min(160 * 1024 // (32768 + pd), 8)} for pd, t in pad_times.items()},
                     "B": b, "H": h, "N": n, "D": d,
                     **{f"occ{v}": {"us_min": round(min(t), 1), "us_median": round(statistics.median(t), 1), "tflops_at_min": round(flops / min(t) / 1e6, 1)}
                        for v, t in times.items()}}
        print(name, out[name], flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
