#!/usr/bin/env python3
"""CPU half of a cheap forward-parity check: compare the UNet outputs tools/gpu/fwd_ab.py dumped on the GPU box with the fp32 CPU oracle.

    (GPU box, no torch import)   python tools/gpu/fwd_ab.py base <knob>=1 --rows 2 --fwd 2 --reps 1 --dump gpurun_out/fwd_dump.npz
    (here)                       python tools/cpu/fwd_parity.py gpurun_out/fwd_dump.npz

The dump holds the inputs and every setting's output; the weights are not shipped — both sides build them from the same seeded numpy
pool (fwd_ab.synthetic_weight), the oracle in fp32 from the fp16-rounded values the engine packed.  Prints rel-L2 of every setting
against the oracle (the engine's stated tolerance at the C1 shape is 2e-3, tests/test_gpu_c1_parity.py) and against the first setting.
This is test tooling: it imports oracle/ and is not imported by the product.
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "gpu"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fwd_ab  # noqa: E402
from oracle import unet as ounet  # noqa: E402


def main():
    z = np.load(sys.argv[1])
    model = str(z["model"])
    schema = importlib.import_module("stable-diffusion-webui_amd.schema")
    what = str(z["what"]) if "what" in z.files else "unet"
    rng = np.random.default_rng(0x5D15)                       # fwd_ab.main's pool: same generator, same first draw
    pool = rng.standard_normal(1 << 22, dtype=np.float32)
    t0 = time.time()
    x = torch.from_numpy(z["x"])
    if what == "vae":
        from oracle import vae as ovae
        cfg = {"sd15": schema.sd15_vae, "sdxl": schema.sdxl_vae, "tiny": schema.tiny_vae}[model]()
        ocfg = {"sd15": ovae.sd15_vae_config, "sdxl": ovae.sd15_vae_config, "tiny": ovae.tiny_vae_config}[model]()     # (SDXL's VAE: the SD1.5 geometry, its own scale factor)
        if model == "sdxl":
            ocfg.scale_factor = cfg.scale_factor
        sd = {schema.VAE_PREFIX + key: torch.from_numpy(fwd_ab.synthetic_weight(pool, key, tuple(shape), kind).astype(np.float32))
              for key, shape, kind in schema.vae_schema(cfg)}
        net = ovae.build_vae(ocfg, sd)
        with torch.no_grad():
            want = torch.cat([net.decode_first_stage(x[i:i + 1]) for i in range(x.shape[0])]).numpy().astype(np.float64)
    else:
        cfg = {"sd15": schema.sd15_unet, "sdxl": schema.sdxl_unet, "tiny": schema.tiny_unet}[model]()
        ocfg = {"sd15": ounet.sd15_config, "sdxl": ounet.sdxl_base_config, "tiny": ounet.tiny_config}[model]()
        sd = {schema.UNET_PREFIX + key: torch.from_numpy(fwd_ab.synthetic_weight(pool, key, tuple(shape), kind).astype(np.float32))
              for key, shape, kind in schema.unet_schema(cfg)}
        net = ounet.build_unet(ocfg, sd).float().eval()
        t, ctx = torch.from_numpy(z["t"]), torch.from_numpy(z["ctx"])
        y = torch.from_numpy(z["y"]) if z["y"].size else None
        with torch.no_grad():
            want = net(x, t, ctx, y).numpy().astype(np.float64)
    print(f"oracle forward: {x.shape[0]} rows in {time.time() - t0:.1f} s")
    base = None
    for i, s in enumerate(z["settings"]):
        got = z[f"out_{i}"].astype(np.float64)
        base = got if base is None else base
        rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        rel0 = float(np.linalg.norm(got - base) / np.linalg.norm(base))
        print(f"{str(s):48s} rel-L2 vs fp32 oracle {rel:.3e}   vs first setting {rel0:.3e}")


if __name__ == "__main__":
    main()
