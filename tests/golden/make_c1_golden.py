#!/usr/bin/env python3
"""Golden final latents of the C1 job at batch 1 (SD1.5 512x512, 20-step Euler a, cfg 7, Philox seed 1000, synthetic checkpoint
seed 0x5D15, synthetic prompt 50000) computed by the fp32 CPU ORACLE (oracle/pipeline.py) — about 3-5 minutes of CPU — plus the
same run under the reference's fp16-autocast rounding pattern (tests/fp16_emu.py).  tests/test_gpu_c1_parity.py compares the HIP
engine with these instead of re-running 40 full-size CPU UNet evaluations inside every GPU test session; with
SDMI_PARITY_FULL=1 it re-runs the oracle live and also checks this fixture against it.

    python tests/golden/make_c1_golden.py        # writes tests/golden/c1_euler_a_b1.npz
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import kdiffusion as kd, pipeline as opipe, unet as ou
    from fp16_emu import fp16_storage
    schema = importlib.import_module("stable-diffusion-webui_amd.schema")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd = schema.synthetic_state_dict(schema.sd15_unet(), None, dtype=torch.float16)
    om = opipe.OracleModel(sd, ou.sd15_config(), None)
    g = torch.Generator().manual_seed(50_000)
    cond, uncond = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    t0 = time.time()
    ref = opipe.sample(om, cond, uncond, [1000], 20, "euler_a", 7.0, (64, 64))
    t1 = time.time()
    with fp16_storage(om.unet):
        emu = opipe.sample(om, cond, uncond, [1000], 20, "euler_a", 7.0, (64, 64))
    t2 = time.time()
    rel = float((emu - ref).norm() / ref.norm())
    print(f"fp32 oracle {t1 - t0:.0f}s, fp16 emulation {t2 - t1:.0f}s, emulation vs fp32 rel-L2 {rel:.3e}")
    np.savez_compressed(os.path.join(HERE, "c1_euler_a_b1.npz"), final_latent_fp32_oracle=ref.numpy(),
                        final_latent_ref_fp16_emulation=emu.numpy(), emulation_vs_fp32=np.array(rel))


if __name__ == "__main__":
    main()
