#!/usr/bin/env python3
"""Tuning aid: time the level-0 self-attention launch (B 16, H 8, N = M = 4096, d 40) for the production kernel, its experiment
variants and — when the library was built with -DSDMI_ATTN_PARTS — the component-removal variants 10..14 (wrong results, timing
only).  Also the d = 80 / 160 launches of the other levels.  (Does not import oracle/.)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
ops, lib = importlib.import_module(f"{PKG}.ops"), importlib.import_module(f"{PKG}._lib")


def bench(B, H, N, D, variants, iters=30):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, N, H * D, generator=g).half().cuda()
    k = torch.randn(B, N, H * D, generator=g).half().cuda()
    vt = torch.randn(B, H * D, N, generator=g).half().cuda()
    flops = 4.0 * B * H * N * N * D
    for var in variants:
        lib.check(lib.lib.sdmi_debug_set(b"attn_occ", var))
        for _ in range(3):
            ops.attention_vt(q, k, vt, H, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.attention_vt(q, k, vt, H, N)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"d={D:3d} N={N:5d} variant {var:2d}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
    lib.check(lib.lib.sdmi_debug_set(b"attn_occ", 15))


def sections(B=16, H=8, N=4096, D=40):
    """variant 18 (SDMI_ATTN_PARTS builds): s_memtime stamps around the sections of a KV-tile iteration, mean over all waves."""
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, N, H * D, generator=g).half().cuda()
    k = torch.randn(B, N, H * D, generator=g).half().cuda()
    vt = torch.randn(B, H * D, N, generator=g).half().cuda()
    nwg = B * H * ((N + 127) // 128)
    dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device="cuda")
    ptr = dbg.data_ptr()
    to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
    L = lib.lib
    lib.check(L.sdmi_debug_set(b"attn_occ", 18))
    for _ in range(2):
        ops.attention_vt(q, k, vt, H, N)
    lib.check(L.sdmi_debug_set(b"attn_dbg_lo", to_i32(ptr & 0xFFFFFFFF))); lib.check(L.sdmi_debug_set(b"attn_dbg_hi", to_i32(ptr >> 32)))
    ops.attention_vt(q, k, vt, H, N)
    torch.cuda.synchronize()
    lib.check(L.sdmi_debug_set(b"attn_dbg_lo", 0)); lib.check(L.sdmi_debug_set(b"attn_dbg_hi", 0))
    lib.check(L.sdmi_debug_set(b"attn_occ", 15))
    d = dbg.cpu().view(nwg * 4, 8).double()
    d = d[d[:, 5] > 0]
    per = d[:, :5] / d[:, 5:6]
    names = ["global loads + K frags + S MFMAs + V frags issued", "mask + max (waits for S)", "rescale + exp/pack + PV MFMAs issued",
             "vmcnt + ds_write", "barrier"]
    print(f"variant 18 sections, cycles per KV-tile iteration per wave (mean / p10 / p90 over {len(d)} waves):")
    for i, n in enumerate(names):
        c = per[:, i]
        print(f"   {n:52s} {c.mean():8.1f} {c.quantile(0.1):8.1f} {c.quantile(0.9):8.1f}")
    print(f"   {'total':52s} {per.sum(1).mean():8.1f}")


if __name__ == "__main__":
    if sys.argv[1:] == ["sections"]:
        sections()
        sys.exit(0)
    variants = [int(a) for a in sys.argv[1:]] or [15, 5, 0]
    bench(16, 8, 4096, 40, variants)
    bench(16, 8, 1024, 80, [15])
    bench(16, 8, 256, 160, [15])
