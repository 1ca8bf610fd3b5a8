#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the shapes of the C1 workload (SD1.5, 512^2, CFG batch 16) and the batch-8 VAE
decode.  HIP-event timed inside the library (sdmi_bench_conv_gemm) or with torch.cuda events around repeated launches.
Usage: python tools/bench_kernels.py [gemm|attn|norm|all]   (writes a table to stdout)"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
lib = importlib.import_module(f"{PKG}._lib")
ops = importlib.import_module(f"{PKG}.ops")
L = lib.lib


def bench_conv(B, H, W, cin, cout, taps=9, stride=1, up=False, c1=0, geglu=False, impl="mfma", iters=20):
    dev = torch.device("cuda")
    a0 = torch.randn(B, H, W, cin - c1, device=dev).half()
    a1 = torch.randn(B, H, W, c1, device=dev).half() if c1 else None
    wp = (torch.randn(cout, taps, cin, device=dev) * (cin * taps) ** -0.5).half().contiguous()
    if up:
        Ho, Wo = 2 * H, 2 * W
    elif taps == 9:
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    else:
        Ho, Wo = H, W
    n_out = cout // 2 if geglu else cout
    out = torch.empty(B, Ho, Wo, n_out, device=dev, dtype=torch.float16)
    bias = torch.zeros(cout, device=dev)
    d = lib.ConvDesc()
    d.a0, d.a1, d.w, d.bias, d.out = a0.data_ptr(), (a1.data_ptr() if c1 else None), wp.data_ptr(), bias.data_ptr(), out.data_ptr()
    d.c0, d.c1, d.lda0, d.lda1 = cin - c1, c1, cin - c1, c1
    d.B, d.Hi, d.Wi, d.Ho, d.Wo = B, H, W, Ho, Wo
    d.taps, d.stride, d.pad, d.up = taps, stride, 1 if taps == 9 else 0, 1 if up else 0
    d.N, d.ldo, d.flags, d.alpha, d.batch = cout, n_out, (lib.EP_GEGLU if geglu else 0), 1.0, 1
    d.force_generic = {"mfma": 0, "generic": 1, "mfma_reg": 2}[impl]
    wsb = L.sdmi_conv_splitk_workspace_bytes(B * Ho * Wo, cout, taps * cin, 1)
    if wsb and not geglu:
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        d.splitk_workspace, d.splitk_workspace_bytes = ws.data_ptr(), wsb
    ms = C.c_float(0)
    lib.check(L.sdmi_bench_conv_gemm(C.byref(d), iters, C.byref(ms), lib.stream_ptr()), "bench_conv")
    flops = 2.0 * B * Ho * Wo * cout * taps * cin
    return ms.value, flops / (ms.value * 1e-3) / 1e12


def bench_attn(B, H, N, M, D, iters=10):
    dev = torch.device("cuda")
    q = torch.randn(B, N, H * D, device=dev).half()
    k = torch.randn(B, M, H * D, device=dev).half()
    mpad = (M + 63) // 64 * 64
    vt = torch.randn(B, H * D, mpad, device=dev).half()
    for _ in range(2):
        ops.attention_vt(q, k, vt, H, M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attention_vt(q, k, vt, H, M)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 4.0 * B * H * N * M * D / (ms * 1e-3) / 1e12


def bench_gn(B, HW, C, iters=20):
    dev = torch.device("cuda")
    side = int(HW ** 0.5)
    x = torch.randn(B, side, side, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(2):
        ops.groupnorm(x, g, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.groupnorm(x, g, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 3.0 * x.numel() * 2 / (ms * 1e-3) / 1e12       # TB/s (read, read, write)


def bench_ln(rows, C, iters=20):
    dev = torch.device("cuda")
    x = torch.randn(rows, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(2):
        ops.layernorm(x, g, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.layernorm(x, g, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * x.numel() * 2 / (ms * 1e-3) / 1e12       # TB/s (read, write)


CFG_NAMES = {-1: "auto", 0: "128x128", 1: "256x64", 2: "64x64", 3: "128x128k32", 4: "256x256", 5: "256x320", 6: "256x128", 7: "128x64", 8: "128x320"}


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    lib.require_device()
    if what in ("gemm", "all"):
        shapes = [
            ("conv3x3 320->320 @64^2 B16", dict(B=16, H=64, W=64, cin=320, cout=320)),
            ("conv3x3 640->640 @32^2 B16", dict(B=16, H=32, W=32, cin=640, cout=640)),
            ("conv3x3 1280->1280 @16^2 B16", dict(B=16, H=16, W=16, cin=1280, cout=1280)),
            ("conv3x3 1280->1280 @8^2 B16", dict(B=16, H=8, W=8, cin=1280, cout=1280)),
            ("conv3x3 (1280+1280)->1280 @16^2", dict(B=16, H=16, W=16, cin=2560, cout=1280, c1=1280)),
            ("conv3x3 (640+320)->320 @64^2", dict(B=16, H=64, W=64, cin=960, cout=320, c1=320)),
            ("conv3x3 640->640 up @32^2", dict(B=16, H=32, W=32, cin=640, cout=640, up=True)),
            ("linear qk 320->640 tok65536", dict(B=16, H=64, W=64, cin=320, cout=640, taps=1)),
            ("linear o 320->320 tok65536", dict(B=16, H=64, W=64, cin=320, cout=320, taps=1)),
            ("linear ff1 geglu 320->2560", dict(B=16, H=64, W=64, cin=320, cout=2560, taps=1, geglu=True)),
            ("linear ff2 1280->320", dict(B=16, H=64, W=64, cin=1280, cout=320, taps=1)),
            ("linear ff1 geglu 640->5120 @32^2", dict(B=16, H=32, W=32, cin=640, cout=5120, taps=1, geglu=True)),
            ("linear ff1 geglu 1280->10240 @16^2", dict(B=16, H=16, W=16, cin=1280, cout=10240, taps=1, geglu=True)),
            ("vae conv3x3 128->128 @512^2 B2", dict(B=2, H=512, W=512, cin=128, cout=128)),
            ("vae conv3x3 256->256 @256^2 B2", dict(B=2, H=256, W=256, cin=256, cout=256)),
            ("vae conv3x3 512->512 @128^2 B2", dict(B=2, H=128, W=128, cin=512, cout=512)),
            ("gemm 8192x8192x8192 (1x1)", dict(B=1, H=8192, W=1, cin=8192, cout=8192, taps=1)),
        ]
        cfgs = [-1, 0, 3, 4, 5, 8, 7, 2]
        print("== implicit GEMM: TFLOP/s per tile config (glds); '-' = config does not fit the shape ==")
        print(f"{'shape':40s} | " + " | ".join(f"{CFG_NAMES[c]:>10s}" for c in cfgs) + " |   reg(auto)")
        bn = {0: 128, 1: 64, 2: 64, 3: 128, 4: 256, 5: 320, 6: 128, 7: 64, 8: 320}
        for name, kw in shapes:
            cells = []
            for c in cfgs:
                if c >= 0 and (kw["cout"] % bn[c] or (c in (5, 8) and kw.get("geglu"))):
                    cells.append(f"{'-':>10s}")
                    continue
                lib.check(L.sdmi_debug_set(b"gemm_cfg", c))
                try:
                    ms, tf = bench_conv(impl="mfma", iters=10, **kw)
                    cells.append(f"{tf:10.1f}")
                except Exception as e:                      # noqa: BLE001
                    cells.append(f"{'ERR':>10s}")
                    print("   error:", e)
            lib.check(L.sdmi_debug_set(b"gemm_cfg", -1))
            try:
                ms, tf = bench_conv(impl="mfma_reg", iters=10, **kw)
                cells.append(f"{tf:10.1f}")
            except Exception as e:                          # noqa: BLE001
                cells.append("ERR")
            print(f"{name:40s} | " + " | ".join(cells), flush=True)
    if what in ("pipe", "phase", "all"):
        shapes_p = [
            ("conv3x3 320->320 @64^2 B16", dict(B=16, H=64, W=64, cin=320, cout=320)),
            ("conv3x3 640->640 @32^2 B16", dict(B=16, H=32, W=32, cin=640, cout=640)),
            ("conv3x3 1280->1280 @16^2 B16", dict(B=16, H=16, W=16, cin=1280, cout=1280)),
            ("conv3x3 1280->1280 @8^2 B16", dict(B=16, H=8, W=8, cin=1280, cout=1280)),
            ("conv3x3 (640+320)->320 @64^2", dict(B=16, H=64, W=64, cin=960, cout=320, c1=320)),
            ("linear 320->320 tok65536", dict(B=16, H=64, W=64, cin=320, cout=320, taps=1)),
            ("linear qk 320->640 tok65536", dict(B=16, H=64, W=64, cin=320, cout=640, taps=1)),
            ("linear 640->640 tok16384", dict(B=16, H=32, W=32, cin=640, cout=640, taps=1)),
            ("linear 1280->1280 tok4096", dict(B=16, H=16, W=16, cin=1280, cout=1280, taps=1)),
            ("linear ff1 geglu 320->2560", dict(B=16, H=64, W=64, cin=320, cout=2560, taps=1, geglu=True)),
            ("linear ff1 geglu 640->5120 @32^2", dict(B=16, H=32, W=32, cin=640, cout=5120, taps=1, geglu=True)),
            ("linear ff1 geglu 1280->10240 @16^2", dict(B=16, H=16, W=16, cin=1280, cout=10240, taps=1, geglu=True)),
            ("linear ff2 1280->320", dict(B=16, H=64, W=64, cin=1280, cout=320, taps=1)),
            ("vae conv3x3 128->128 @512^2 B2", dict(B=2, H=512, W=512, cin=128, cout=128)),
            ("vae conv3x3 256->256 @256^2 B2", dict(B=2, H=256, W=256, cin=256, cout=256)),
            ("vae conv3x3 512->512 @128^2 B2", dict(B=2, H=128, W=128, cin=512, cout=512)),
            ("gemm 8192x8192x8192 (1x1)", dict(B=1, H=8192, W=1, cin=8192, cout=8192, taps=1)),
        ]
    if what in ("pipe", "all"):
        print("== two-stage (pipe=0) vs deep-pipelined (pipe=1; 2 = also 128x128) kernels, auto tile choice: TFLOP/s ==")
        for name, kw in shapes_p:
            cells = []
            for pv in (0, 1, 2):
                lib.check(L.sdmi_debug_set(b"gemm_pipe", pv))
                try:
                    ms, tf = bench_conv(impl="mfma", iters=10, **kw)
                    cells.append(f"{tf:8.1f}")
                except Exception as e:                      # noqa: BLE001
                    cells.append("ERR")
                    print("   error:", e)
            lib.check(L.sdmi_debug_set(b"gemm_pipe", 0))
            print(f"{name:40s} | " + " | ".join(cells), flush=True)
    if what in ("phase",):
        print("== two-stage (a) vs phase-split (b) kernels, TFLOP/s; auto tile choice, then forced 256x320 / 256x256 / 128x320 ==")
        bn = {5: 320, 4: 256, 8: 320}
        for name, kw in shapes_p:
            cells = []
            for cfg in (-1, 5, 4, 8):
                if cfg >= 0 and (kw["cout"] % bn[cfg] or (kw.get("geglu") and cfg != 4)):
                    cells.append("       -        ")
                    continue
                lib.check(L.sdmi_debug_set(b"gemm_cfg", cfg))
                pair = []
                for pv in (0, 3):
                    lib.check(L.sdmi_debug_set(b"gemm_pipe", pv))
                    try:
                        ms, tf = bench_conv(impl="mfma", iters=10, **kw)
                        pair.append(f"{tf:7.1f}")
                    except Exception as e:                  # noqa: BLE001
                        pair.append("ERR")
                        print("   error:", e)
                cells.append("/".join(pair))
            lib.check(L.sdmi_debug_set(b"gemm_pipe", -1)); lib.check(L.sdmi_debug_set(b"gemm_cfg", -1))
            print(f"{name:40s} | " + " | ".join(cells), flush=True)
    if what in ("split", "all"):
        print("== split-K on the deep levels: TFLOP/s for (cfg, slices); slices=1 is the plain kernel ==")
        deep = [("conv3x3 1280->1280 @16^2 B16", dict(B=16, H=16, W=16, cin=1280, cout=1280)),
                ("conv3x3 1280->1280 @8^2 B16", dict(B=16, H=8, W=8, cin=1280, cout=1280)),
                ("conv3x3 (1280+1280)->1280 @8^2", dict(B=16, H=8, W=8, cin=2560, cout=1280, c1=1280)),
                ("conv3x3 (1280+640)->1280 @16^2", dict(B=16, H=16, W=16, cin=1920, cout=1280, c1=640))]
        combos = [(c, sp) for c in (5, 8, 0, 3, 4) for sp in (1, 2, 3, 4, 6, 8)]
        for name, kw in deep:
            best = (0, None)
            line = []
            for c, sp in combos:
                lib.check(L.sdmi_debug_set(b"gemm_cfg", c)); lib.check(L.sdmi_debug_set(b"gemm_split", sp))
                try:
                    ms, tf = bench_conv(impl="mfma", iters=10, **kw)
                except Exception as e:                      # noqa: BLE001
                    tf = 0.0
                line.append(f"{CFG_NAMES[c]}/s{sp}:{tf:.0f}")
                if tf > best[0]:
                    best = (tf, f"{CFG_NAMES[c]}/s{sp}")
            lib.check(L.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(L.sdmi_debug_set(b"gemm_split", 0))
            ms, tf = bench_conv(impl="mfma", iters=10, **kw)
            print(f"{name:36s} auto={tf:.0f}  best={best[1]}:{best[0]:.0f}\n      " + " ".join(line), flush=True)
    if what in ("attn", "all"):
        print("== flash attention (ms | TFLOP/s): KV tile heuristic vs forced 64 ==")
        for name, kw in [("self d40 N4096 B16 H8", dict(B=16, H=8, N=4096, M=4096, D=40)),
                         ("self d80 N1024 B16 H8", dict(B=16, H=8, N=1024, M=1024, D=80)),
                         ("self d160 N256 B16 H8", dict(B=16, H=8, N=256, M=256, D=160)),
                         ("self d160 N64 B16 H8", dict(B=16, H=8, N=64, M=64, D=160)),
                         ("cross d40 N4096 M77", dict(B=16, H=8, N=4096, M=77, D=40)),
                         ("cross d80 N1024 M77", dict(B=16, H=8, N=1024, M=77, D=80)),
                         ("self d64 N4096 B8 H10 (sdxl)", dict(B=8, H=10, N=4096, M=4096, D=64)),
                         ("self d128 N4096 B4 H8", dict(B=4, H=8, N=4096, M=4096, D=128))]:
            r = []
            for kvt in (0, 64):
                lib.check(L.sdmi_debug_set(b"attn_kvt", kvt))
                try:
                    ms, tf = bench_attn(**kw)
                    r.append(f"{ms:8.3f} ms {tf:7.1f} TF")
                except Exception as e:                      # noqa: BLE001
                    r.append(f"ERR {e}")
            lib.check(L.sdmi_debug_set(b"attn_kvt", 0))
            print(f"{name:40s} | " + " | ".join(r), flush=True)
    if what in ("norm", "all"):
        print("== GroupNorm+SiLU (ms | TB/s algorithmic: 3 x tensor bytes) ==")
        for name, kw in [("B16 64^2 C320", dict(B=16, HW=4096, C=320)), ("B16 32^2 C640", dict(B=16, HW=1024, C=640)),
                         ("B16 16^2 C1280", dict(B=16, HW=256, C=1280)), ("B16 8^2 C1280", dict(B=16, HW=64, C=1280)),
                         ("B16 64^2 C960", dict(B=16, HW=4096, C=960)), ("B16 32^2 C1920", dict(B=16, HW=1024, C=1920)),
                         ("B8 512^2 C128 (vae)", dict(B=8, HW=262144, C=128)),
                         ("B8 256^2 C256 (vae)", dict(B=8, HW=65536, C=256))]:
            try:
                ms, tb = bench_gn(**kw)
                print(f"{name:40s} | {ms:8.3f} ms {tb:7.3f} TB/s", flush=True)
            except Exception as e:                          # noqa: BLE001
                print(f"{name:40s} | ERR {e}")
        print("== LayerNorm (ms | TB/s algorithmic: 2 x tensor bytes) ==")
        for rows, C in [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280), (8192, 2048)]:
            ms, tb = bench_ln(rows, C)
            print(f"rows {rows:6d} C {C:5d}                       | {ms:8.3f} ms {tb:7.3f} TB/s", flush=True)


if __name__ == "__main__":
    main()
