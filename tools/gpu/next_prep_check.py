#!/usr/bin/env python3
"""One-off: exercise the next-round-prep build (SDMI_LIB=.../libsdmi_next.so) — LayerNorm fold on / off on the tiny UNet."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import rel_l2, seeded
PKG = "stable-diffusion-webui_amd"
schema = importlib.import_module(f"{PKG}.schema")
sd_models = importlib.import_module(f"{PKG}.sd_models")
ucfg, vcfg = schema.tiny_unet(), schema.tiny_vae()
sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16)
model = sd_models.SdModel(sd, ucfg, vcfg, device=0)
eng = model.engine
g = torch.Generator().manual_seed(11)
cond = torch.randn(4, 77, 64, generator=g)
x, t, ctx = seeded((2, 4, 16, 16), 2).cuda(), torch.tensor([700.0, 20.0]).cuda(), cond[:2].cuda()
a = eng.unet_forward(x, t, ctx)
eng.set_option("ln_fold", 1)
b = eng.unet_forward(x, t, ctx)
b2 = eng.unet_forward(x, t, ctx)
eng.set_option("ln_fold", 0)
c = eng.unet_forward(x, t, ctx)
torch.cuda.synchronize()
print(f"[ln_fold] folded vs separate LayerNorm rel-L2 {rel_l2(b.cpu(), a.cpu()):.3e}; deterministic {torch.equal(b, b2)}; off again equals first {torch.equal(a, c)}; finite {bool(torch.isfinite(b).all())}")
