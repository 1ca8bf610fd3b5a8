#!/usr/bin/env python3
"""Section timers of the ping-pong GEMM kernel (tuning aid): cycles per phase spent in L work / barrier a / M issue / barrier b."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = importlib.import_module("stable-diffusion-webui_amd._lib")
ops = importlib.import_module("stable-diffusion-webui_amd.ops")
L = lib.lib


def run(name, B, H, W, cin, cout, taps, cfg, bm=256):
    dev = torch.device("cuda")
    k = 3 if taps == 9 else 1
    x = torch.randn(B, H, W, cin, device=dev).half()
    w = (torch.randn(cout, cin, k, k, device=dev) * (cin * k * k) ** -0.5).half()
    wp = ops.pack_conv_weight(w)
    bn = {4: 256, 5: 320}[cfg]
    nblk = (B * H * W + bm - 1) // bm * (cout // bn)
    dbg = torch.zeros(nblk * 8 * 17, dtype=torch.int64, device=dev)
    lib.check(L.sdmi_debug_set(b"gemm_cfg", cfg)); lib.check(L.sdmi_debug_set(b"gemm_pipe", 3))
    for _ in range(3):
        ops.conv_gemm(x, wp, taps=taps)
    ptr = dbg.data_ptr()
    lo, hi = ptr & 0xFFFFFFFF, ptr >> 32
    to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
    lib.check(L.sdmi_debug_set(b"gemm_dbg_lo", to_i32(lo))); lib.check(L.sdmi_debug_set(b"gemm_dbg_hi", to_i32(hi)))
    ops.conv_gemm(x, wp, taps=taps)
    torch.cuda.synchronize()
    lib.check(L.sdmi_debug_set(b"gemm_dbg_lo", 0)); lib.check(L.sdmi_debug_set(b"gemm_dbg_hi", 0))
    lib.check(L.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(L.sdmi_debug_set(b"gemm_pipe", -1))
    d = dbg.cpu().view(nblk, 8, 17).double()
    print(f"{name}: {nblk} blocks; cycles per phase (mean over blocks and K tiles), MFMA floor per wave-phase = {2 * 2 * (bn // 64) * 16}")
    for g in (0, 1):
        v = d[:, 4 * g:4 * g + 4, :]
        nt = v[..., 16].sum()
        tot = 0.0
        for ph in range(4):
            m = v[..., ph * 4:ph * 4 + 4].sum(dim=(0, 1)) / nt
            tot += float(m.sum())
            print(f"  group {g} phase {ph}: L={m[0]:7.1f}  barrier_a={m[1]:7.1f}  M={m[2]:7.1f}  barrier_b={m[3]:7.1f}  sum={m.sum():7.1f}")
        print(f"  group {g}: K tile = {tot:7.1f} cycles (MFMA floor {4 * 2 * 2 * 2 * (bn // 64) * 16})")


if __name__ == "__main__":
    lib.require_device()
    run("unet conv3x3 320->320 @64^2 B16 (256x320)", 16, 64, 64, 320, 320, 9, 5)
    run("unet conv3x3 640->640 @32^2 B16 (256x320)", 16, 32, 32, 640, 640, 9, 5)
    run("unet conv3x3 1280->1280 @16^2 B16 (256x320)", 16, 16, 16, 1280, 1280, 9, 5)
    run("unet 1x1 320->320 @64^2 B16 (256x320, K = 320)", 16, 64, 64, 320, 320, 1, 5)
    run("vae conv3x3 512->512 @128^2 B2 (256x256)", 2, 128, 128, 512, 512, 9, 4)
