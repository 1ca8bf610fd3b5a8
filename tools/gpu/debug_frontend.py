#!/usr/bin/env python3
"""One-off: localise the abort of the PIL front-end test (run with AMD_SERIALIZE_KERNEL=3)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from PIL import Image
PKG = "stable-diffusion-webui_amd"
schema = importlib.import_module(f"{PKG}.schema"); sd_models = importlib.import_module(f"{PKG}.sd_models"); processing = importlib.import_module(f"{PKG}.processing")
ucfg, vcfg = schema.tiny_unet(), schema.tiny_vae(ch_mult=(1, 1, 2, 2))
sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16)
model = sd_models.SdModel(sd, ucfg, vcfg, device=0)
g = torch.Generator().manual_seed(11)
cond, uncond = torch.randn(2, 77, 64, generator=g), torch.randn(2, 77, 64, generator=g)
rs = np.random.RandomState(12)
W = H = 128
base = Image.fromarray(rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8))
m = np.zeros((H, W), np.uint8); m[40:90, 30:100] = 255
rgba = np.zeros((H, W, 4), np.uint8); rgba[..., 3] = m
mask = Image.fromarray(rgba, "RGBA")
def job(**kw):
    return processing.StableDiffusionProcessingImg2Img(sd_model=model, c=cond, uc=uncond, seed=77, batch_size=2, steps=4, cfg_scale=4.0, width=W, height=H,
                                                       sampler_name="Euler a", denoising_strength=0.6, **kw)
def step(tag):
    torch.cuda.synchronize(); print("OK", tag, flush=True)
p = job(init_images=[base], mask_image=mask, inpainting_fill=0, inpaint_full_res=False, mask_blur=0)
res = processing.process_images(p); step("p")
print("p tensors", p.init_images.shape, p.init_images.dtype, p.latent_mask.shape, p.latent_mask.dtype, p.latent_mask.device, p.image_mask.shape, flush=True)
p2 = job(init_images=p.init_images.clone(), latent_mask=p.latent_mask.clone(), image_mask=p.image_mask.clone(), inpainting_fill=1)
p2.all_seeds = [77, 78]; p2.all_subseeds = [0, 0]
p2.init(None, p2.all_seeds, None); step("p2.init")
res2 = processing.process_images(p2); step("p2")
print("equal", torch.equal(res.latents, res2.latents))
for kw in (dict(inpainting_fill=1, inpaint_full_res=False, mask_blur=0), dict(inpainting_fill=1, inpaint_full_res=False, mask_blur=4)):
    r = processing.process_images(job(init_images=[base], mask_image=mask, **kw)); step(str(kw))
