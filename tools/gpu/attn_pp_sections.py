#!/usr/bin/env python3
"""Section timers of the role-offset attention kernel (attn_occ 28 = two groups, 38 = three): cycles per KV-tile iteration per wave in
each section and in each barrier wait, by wave group.   python tools/gpu/attn_pp_sections.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
ops, lib = importlib.import_module(f"{PKG}.ops"), importlib.import_module(f"{PKG}._lib")


def run(var, ng, B=16, H=8, N=4096, D=40):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, N, H * D, generator=g).half().cuda()
    k = torch.randn(B, N, H * D, generator=g).half().cuda()
    vt = torch.randn(B, H * D, N, generator=g).half().cuda()
    nwg = B * H * ((N + ng * 128 - 1) // (ng * 128))
    dbg = torch.zeros(nwg * 4 * ng * 8, dtype=torch.int64, device="cuda")
    ptr = dbg.data_ptr()
    to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
    L = lib.lib
    lib.check(L.sdmi_debug_set(b"attn_occ", var))
    for _ in range(2):
        ops.attention_vt(q, k, vt, H, N)
    lib.check(L.sdmi_debug_set(b"attn_dbg_lo", to_i32(ptr & 0xFFFFFFFF))); lib.check(L.sdmi_debug_set(b"attn_dbg_hi", to_i32(ptr >> 32)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.attention_vt(q, k, vt, H, N)
    e1.record()
    torch.cuda.synchronize()
    lib.check(L.sdmi_debug_set(b"attn_dbg_lo", 0)); lib.check(L.sdmi_debug_set(b"attn_dbg_hi", 0))
    lib.check(L.sdmi_debug_set(b"attn_occ", 15))
    d = dbg.cpu().view(nwg, 4 * ng, 8).double()
    names = ["V1 work", "wait 1", "V2 work", "wait 2", "M work", "wait 3"]
    print(f"variant {var} ({ng} groups), instrumented launch {e0.elapsed_time(e1) * 1e3:.1f} us; cycles per KV-tile iteration per wave:")
    for grp in range(ng):
        w = d[:, grp * 4:(grp + 1) * 4].reshape(-1, 8)
        w = w[w[:, 6] > 0]
        per = w[:, :6] / w[:, 6:7]
        print(f"  group {grp}: " + "  ".join(f"{n} {per[:, i].mean():7.1f}" for i, n in enumerate(names)) + f"   total {per.sum(1).mean():7.1f}")


if __name__ == "__main__":
    run(38, 3)
    run(28, 2)
    run(38, 3, B=2, N=16384)
