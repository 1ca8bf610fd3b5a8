#!/usr/bin/env python3
"""What a streaming kernel reaches on this box, next to the engine's norm kernels at the C1 shapes: device-to-device copy (read + write of
the same bytes as one LayerNorm), torch's own fp16 layer_norm / group_norm + silu (rocm kernels, yardstick only), and the engine's
layernorm / groupnorm.  Prints one JSON object.  (Does not import oracle/.)"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PKG = "stable-diffusion-webui_amd"


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    importlib.import_module(f"{PKG}._lib").require_device()
    out = {}
    lib = importlib.import_module(f"{PKG}._lib")
    for name, rows, c in (("level 0 (84 MB moved, MALL-sized)", 65536, 320), ("8 x level 0 (671 MB moved, HBM)", 524288, 320),
                          ("level 1", 16384, 640), ("8 x level 1", 131072, 640)):
        x = torch.randn(rows, c, device="cuda").half()
        y = torch.empty_like(x)
        g, b = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
        gh, bh = g.half(), b.half()
        nbytes = 2.0 * x.numel() * 2
        r = {"bytes_read_plus_written": nbytes}
        t = timed(lambda: y.copy_(x)); r["copy_us"] = round(t, 2); r["copy_TBps"] = round(nbytes / t / 1e6, 2)
        t = timed(lambda: F.layer_norm(x, (c,), gh, bh)); r["torch_layer_norm_us"] = round(t, 2); r["torch_layer_norm_TBps"] = round(nbytes / t / 1e6, 2)
        call = lambda: lib.lib.sdmi_layernorm(lib.ptr(x), lib.ptr(g), lib.ptr(b), lib.ptr(y), rows, c, 1e-5, lib.stream_ptr())
        t = timed(call); r["engine_layernorm_us"] = round(t, 2); r["engine_layernorm_TBps"] = round(nbytes / t / 1e6, 2)
        out[name] = r
        print(name, r, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stream_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
