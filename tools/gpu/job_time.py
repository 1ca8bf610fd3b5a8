#!/usr/bin/env python3
"""Time the C1 job (SD1.5 512x512, 20-step Euler a, batch 8, decode included) with whatever library SDMI_LIB points at — no knobs, no
profiling: the two-builds A/B of whole rounds (tools/gpu/r03_ab_rounds.sh).  Prints one JSON line.  (Does not import oracle/.)"""
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


def main():
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    sub("_lib").require_device()
    schema, sd_models, processing = sub("schema"), sub("sd_models"), sub("processing")
    ucfg, vcfg = schema.sd15_unet(), schema.sd15_vae()
    model = sd_models.SdModel(schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16), ucfg, vcfg, device=0, vae_decoder_only=True)
    g = torch.Generator().manual_seed(50_000)
    c, uc = torch.randn(8, 77, 768, generator=g).cuda(), torch.randn(8, 77, 768, generator=g).cuda()

    def job():
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=model, c=c, uc=uc, seed=1000, batch_size=8, n_iter=1, steps=20, cfg_scale=7.0,
                                                        width=512, height=512, sampler_name="Euler a", keep_latents=False)
        return processing.process_images(p)
    job(); job()
    ts = []
    for _ in range(jobs):
        torch.cuda.synchronize(); t0 = time.time(); job(); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    print(json.dumps({"lib": os.environ.get("SDMI_LIB", "default"), "ms_min": round(min(ts), 2), "ms_median": round(statistics.median(ts), 2),
                      "all": [round(t, 2) for t in ts]}), flush=True)


if __name__ == "__main__":
    main()
