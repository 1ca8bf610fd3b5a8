#!/usr/bin/env python3
"""fp32 CPU-ORACLE outputs at the full-size shapes `bench.py --config c2 | c3 | c4a | c4b` times (BASELINE.json configs[2..4]), which no
GPU test session can afford to recompute: each leg is minutes of host CPU, so the oracle is run ONCE here, in the authoring container,
and tests/test_gpu_fullsize_parity.py compares the HIP engine with the committed outputs (inputs are re-derived from the seeds).

    python tests/golden/make_fullsize_golden.py [leg ...]        # legs: c4_unet128 c3_sdxl128 vae1024 vae1024_xl enc512 c2_dpmpp2m
                                                                 #       c4a_hires c4b_img2img c3_sdxl_e2e   (round 4: compositions)

  c4_unet128   SD1.5 UNet, one CFG pair (2 rows), 128x128 latent (the hires pass of c4a: modules/processing.py:1364-1464;
               level-0 self-attention N = M = 16384, d = 40)
  c3_sdxl128   SDXL-base UNet (configs/sd_xl_inpaint.yaml:19-37 with 4 input channels), 1 row, 128x128 latent (1024x1024 images:
               d = 64, 10 / 20 heads, N = 4096 / 1024)
  vae1024      SD1.5 first stage: 128x128 latent -> 1024x1024 image (mid-block attention d = 512, N = 16384:
               modules/sd_hijack_optimizations.py:554-610 computes the same product chunked)
  vae1024_xl   the same decoder weights under the SDXL VAE configuration (scale_factor 0.13025) on a latent whose fp16 activations
               overflow without the range-extended decode (modules/processing.py:636-665: the reference re-runs such a VAE in fp32)
  enc512       SD1.5 first stage ENCODE of a 512x512 image -> posterior moments [1, 8, 64, 64] (modules/sd_samplers_common.py:87-112)
  c2_dpmpp2m   the c2 job at batch 1: 50-step DPM++ 2M on the Karras schedule, cfg 7, 512x512, Philox seed 2000 — final latent
               (modules/sd_samplers_kdiffusion.py:12,18,116-127)

  c4a_hires    the c4a job composed end to end at batch 2: txt2img 512x512, 2 Euler-a steps -> bilinear latent upscale to 128x128
               (modules/processing.py:1364-1398) -> second pass (sample_img2img with steps given, 2 evaluations, :1429-1454) ->
               first-stage decode at 1024x1024 of image 0 (sub-sampled)
  c4b_img2img  the c4b job at batch 2: first-stage encode of two 512x512 images (modules/processing.py:1745-1760) -> Euler-a img2img,
               steps 4, denoising 0.5 (t_enc = 2: 3 evaluations, modules/sd_samplers_kdiffusion.py:134-178) -> decode at 512x512
  c3_sdxl_e2e  SDXL-base, batch 1, 3 Euler-a steps with CFG 5 at a 128x128 latent (vector conditioning y / uy through the CFG
               denoiser): final latent

  round 6 — the BASELINE.json jobs at their full step counts (one image each; tests/test_gpu_fullsize_parity.py runs the engine at the
  BENCHED batch with this image as image 0 / 7, so the driver's default `pytest -m gpu` holds every config end to end):
  c1_b8_img7     image 7 of the C1 batch (Philox seed 1007, prompt generator 50007): 20-step Euler a, cfg 7 — image 0 is
                 c1_euler_a_b1.npz (make_c1_golden.py)
  c3_sdxl_e2e30  SDXL-base, 30 Euler-a steps, cfg 5, 128x128 latent, seed 4300: final latent
  c4a_hires20    txt2img 512x512 20 Euler-a steps -> bilinear latent upscale -> second pass with steps given (20, denoise 0.75) ->
                 decode at 1024x1024 (sub-sampled), seed 4400

Attention products above 4096 query rows are evaluated in row blocks (oracle.unet.QUERY_CHUNK: softmax rows are independent, the
result is the unchunked product's) — the memory bound modules/sub_quadratic_attention.py puts on the reference's own product.
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

# input definitions shared with tests/test_gpu_fullsize_parity.py (which imports this module for them)
SPEC = {
    "c4_unet128": dict(x=((2, 4, 128, 128), 401), t=[801.0, 201.0], ctx=((2, 77, 768), 402)),
    "c3_sdxl128": dict(x=((1, 4, 128, 128), 411), t=[601.0], ctx=((1, 77, 2048), 412), y=((1, 2816), 413)),
    "vae1024": dict(z=((1, 4, 128, 128), 421), z_scale=0.18215 * 4.5),
    "vae1024_xl": dict(z=((1, 4, 128, 128), 423), z_scale=0.13025 * 4.5, weight_gain=2.0e5),
    "enc512": dict(x=((1, 3, 512, 512), 431)),
    "c2_dpmpp2m": dict(prompt_seed=50_002, seed=2000, steps=50, cfg=7.0),
    "c4a_hires": dict(prompt_seed=50_004, seeds=[4000, 4001], steps=2, cfg=7.0, denoising_strength=0.75),
    "c4b_img2img": dict(prompt_seed=50_005, seeds=[4100, 4101], steps=4, cfg=7.0, denoising_strength=0.5, image_seed=441),
    "c3_sdxl_e2e": dict(prompt_seed=50_003, seeds=[4200], steps=3, cfg=5.0),
    # round 6: the BASELINE.json jobs at their FULL step counts, one image each; the engine runs them at the benched batch with this
    # image as image 0 (prompt pair of image i: generator seed prompt_seed + i, Philox seed seeds[0] + i)
    "c1_b8_img7": dict(prompt_seed=50_007, seeds=[1007], steps=20, cfg=7.0),
    "c3_sdxl_e2e30": dict(prompt_seed=50_013, seeds=[4300], steps=30, cfg=5.0),
    "c4a_hires20": dict(prompt_seed=50_014, seeds=[4400], steps=20, cfg=7.0, denoising_strength=0.75),
}


def seeded(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def xl_decoder_state_dict(schema, gain):
    """Synthetic first stage in the SDXL VAE configuration whose decoder residual stream passes fp16's 65504 after the first mid block
    (what the real SDXL VAE does on some images: the reference re-runs it in fp32, modules/processing.py:636-665) — the plain decode
    overflows, the fp32 oracle and the range-extended decode stay finite (same construction as
    tests/test_gpu_models.py::test_nan_check_and_vae_range_extended_retry on the tiny decoder)."""
    sd = schema.synthetic_state_dict(None, schema.sdxl_vae(), dtype=torch.float16)
    k = schema.VAE_PREFIX + "decoder.mid.block_1.conv2.weight"
    sd[k] = (sd[k].float() * gain).half()
    return sd


def main(legs):
    from oracle import kdiffusion as kd, pipeline as opipe, unet as ou, vae as ov
    schema = importlib.import_module("stable-diffusion-webui_amd.schema")
    torch.set_num_threads(int(os.environ.get("SDMI_GOLDEN_THREADS", min(32, os.cpu_count() or 1))))
    ou.QUERY_CHUNK = ov.QUERY_CHUNK = 2048
    done = {}

    def save(name, **arrs):
        np.savez_compressed(os.path.join(HERE, f"fullsize_{name}.npz"), **arrs)
        print(f"[{name}] saved ({', '.join(f'{k}{tuple(v.shape)}' for k, v in arrs.items())})", flush=True)

    if "c4_unet128" in legs:
        s = SPEC["c4_unet128"]
        sd = schema.synthetic_state_dict(schema.sd15_unet(), None, dtype=torch.float16)
        net = ou.build_unet(ou.sd15_config(), sd)
        t0 = time.time()
        with torch.no_grad():
            out = net(seeded(*s["x"]), torch.tensor(s["t"]), seeded(*s["ctx"]).half().float())
        print(f"[c4_unet128] {time.time() - t0:.0f}s", flush=True)
        save("c4_unet128", out=out.numpy())
        del net, sd
    if "c3_sdxl128" in legs:
        s = SPEC["c3_sdxl128"]
        sd = schema.synthetic_state_dict(schema.sdxl_unet(), None, dtype=torch.float16)
        net = ou.build_unet(ou.sdxl_base_config(), sd)
        del sd
        t0 = time.time()
        with torch.no_grad():
            out = net(seeded(*s["x"]), torch.tensor(s["t"]), seeded(*s["ctx"]).half().float(), seeded(*s["y"]).half().float())
        print(f"[c3_sdxl128] {time.time() - t0:.0f}s", flush=True)
        save("c3_sdxl128", out=out.numpy())
        del net
    if "vae1024" in legs or "enc512" in legs:
        sd = schema.synthetic_state_dict(None, schema.sd15_vae(), dtype=torch.float16)
        vae = ov.build_vae(ov.sd15_vae_config(), sd)
        if "vae1024" in legs:
            s = SPEC["vae1024"]
            t0 = time.time()
            with torch.no_grad():
                img = vae.decode_first_stage(seeded(*s["z"]) * s["z_scale"])
            print(f"[vae1024] {time.time() - t0:.0f}s", flush=True)
            # every 4th pixel of the whole image + one dense 128x128 window (so an error pattern with period 4 cannot hide)
            save("vae1024", out_sub4=img[:, :, ::4, ::4].numpy(), out_window=img[:, :, 448:576, 448:576].numpy(),
                 norm=np.array(float(img.norm())), mean=np.array(float(img.mean())))
        if "enc512" in legs:
            s = SPEC["enc512"]
            t0 = time.time()
            with torch.no_grad():
                mom = vae.encode_moments(seeded(*s["x"]).clamp(-1, 1))
            print(f"[enc512] {time.time() - t0:.0f}s", flush=True)
            save("enc512", moments=mom.numpy())
        del vae, sd
    if "vae1024_xl" in legs:
        s = SPEC["vae1024_xl"]
        sd = xl_decoder_state_dict(schema, s["weight_gain"])
        cfg = ov.sd15_vae_config()
        cfg.scale_factor = 0.13025
        vae = ov.build_vae(cfg, sd)
        t0 = time.time()
        with torch.no_grad():
            img = vae.decode_first_stage(seeded(*s["z"]) * s["z_scale"])
        print(f"[vae1024_xl] {time.time() - t0:.0f}s", flush=True)
        save("vae1024_xl", out_sub4=img[:, :, ::4, ::4].numpy(), out_window=img[:, :, 448:576, 448:576].numpy(),
             norm=np.array(float(img.norm())), mean=np.array(float(img.mean())))
        del vae, sd
    if "c2_dpmpp2m" in legs:
        s = SPEC["c2_dpmpp2m"]
        sd = schema.synthetic_state_dict(schema.sd15_unet(), None, dtype=torch.float16)
        om = opipe.OracleModel(sd, ou.sd15_config(), None)
        g = torch.Generator().manual_seed(s["prompt_seed"])
        cond, uncond = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
        t0 = time.time()
        ref = opipe.sample(om, cond, uncond, [s["seed"]], s["steps"], "dpmpp_2m", s["cfg"], (64, 64), scheduler="karras")
        print(f"[c2_dpmpp2m] {time.time() - t0:.0f}s", flush=True)
        save("c2_dpmpp2m", final_latent=ref.numpy())
    if "c4a_hires" in legs or "c4b_img2img" in legs:
        sd = schema.synthetic_state_dict(schema.sd15_unet(), schema.sd15_vae(), dtype=torch.float16)
        om = opipe.OracleModel(sd, ou.sd15_config(), ov.sd15_vae_config())
        del sd
        if "c4a_hires" in legs:
            s = SPEC["c4a_hires"]
            g = torch.Generator().manual_seed(s["prompt_seed"])
            cond, uncond = torch.randn(2, 77, 768, generator=g), torch.randn(2, 77, 768, generator=g)
            t0 = time.time()
            lat = opipe.txt2img_hires(om, cond.half().float(), uncond.half().float(), s["seeds"], s["steps"], "euler_a", s["cfg"], (64, 64),
                                      hr_scale=2.0, denoising_strength=s["denoising_strength"])
            print(f"[c4a_hires] sampling {time.time() - t0:.0f}s", flush=True)
            with torch.no_grad():
                img = om.vae.decode_first_stage(lat[:1])
            print(f"[c4a_hires] {time.time() - t0:.0f}s", flush=True)
            save("c4a_hires", final_latent=lat.numpy(), image0_sub4=img[:, :, ::4, ::4].numpy(), image0_window=img[:, :, 448:576, 448:576].numpy())
        if "c4b_img2img" in legs:
            s = SPEC["c4b_img2img"]
            g = torch.Generator().manual_seed(s["prompt_seed"])
            cond, uncond = torch.randn(2, 77, 768, generator=g), torch.randn(2, 77, 768, generator=g)
            image = torch.rand((2, 3, 512, 512), generator=torch.Generator().manual_seed(s["image_seed"]))
            t0 = time.time()
            with torch.no_grad():
                init = om.vae.encode_first_stage_mean(image.half().float() * 2 - 1)
            lat = opipe.sample(om, cond.half().float(), uncond.half().float(), s["seeds"], s["steps"], "euler_a", s["cfg"], (64, 64),
                               init_latent=init, denoising_strength=s["denoising_strength"], img2img_steps_given=False)
            with torch.no_grad():
                img = opipe.decode(om, lat)
            print(f"[c4b_img2img] {time.time() - t0:.0f}s", flush=True)
            save("c4b_img2img", init_latent=init.numpy(), final_latent=lat.numpy(), images_sub2=img[:, :, ::2, ::2].numpy())
        del om
    if "c3_sdxl_e2e" in legs:
        s = SPEC["c3_sdxl_e2e"]
        sd = schema.synthetic_state_dict(schema.sdxl_unet(), None, dtype=torch.float16)
        om = opipe.OracleModel(sd, ou.sdxl_base_config(), None)
        del sd
        g = torch.Generator().manual_seed(s["prompt_seed"])
        cond, uncond = torch.randn(1, 77, 2048, generator=g), torch.randn(1, 77, 2048, generator=g)
        y, uy = torch.randn(1, 2816, generator=g), torch.randn(1, 2816, generator=g)
        t0 = time.time()
        lat = opipe.sample(om, cond.half().float(), uncond.half().float(), s["seeds"], s["steps"], "euler_a", s["cfg"], (128, 128),
                           y=y.half().float(), uy=uy.half().float())
        print(f"[c3_sdxl_e2e] {time.time() - t0:.0f}s", flush=True)
        save("c3_sdxl_e2e", final_latent=lat.numpy())
        del om
    if "c1_b8_img7" in legs:
        s = SPEC["c1_b8_img7"]
        sd = schema.synthetic_state_dict(schema.sd15_unet(), None, dtype=torch.float16)
        om = opipe.OracleModel(sd, ou.sd15_config(), None)
        del sd
        g = torch.Generator().manual_seed(s["prompt_seed"])
        cond, uncond = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
        t0 = time.time()
        lat = opipe.sample(om, cond, uncond, s["seeds"], s["steps"], "euler_a", s["cfg"], (64, 64))
        print(f"[c1_b8_img7] {time.time() - t0:.0f}s", flush=True)
        save("c1_b8_img7", final_latent=lat.numpy())
        del om
    if "c4a_hires20" in legs:
        s = SPEC["c4a_hires20"]
        sd = schema.synthetic_state_dict(schema.sd15_unet(), schema.sd15_vae(), dtype=torch.float16)
        om = opipe.OracleModel(sd, ou.sd15_config(), ov.sd15_vae_config())
        del sd
        g = torch.Generator().manual_seed(s["prompt_seed"])
        cond, uncond = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
        t0 = time.time()
        lat = opipe.txt2img_hires(om, cond.half().float(), uncond.half().float(), s["seeds"], s["steps"], "euler_a", s["cfg"], (64, 64),
                                  hr_scale=2.0, denoising_strength=s["denoising_strength"])
        print(f"[c4a_hires20] sampling {time.time() - t0:.0f}s", flush=True)
        with torch.no_grad():
            img = om.vae.decode_first_stage(lat[:1])
        print(f"[c4a_hires20] {time.time() - t0:.0f}s", flush=True)
        save("c4a_hires20", final_latent=lat.numpy(), image0_sub4=img[:, :, ::4, ::4].numpy(), image0_window=img[:, :, 448:576, 448:576].numpy())
        del om
    if "c3_sdxl_e2e30" in legs:
        s = SPEC["c3_sdxl_e2e30"]
        sd = schema.synthetic_state_dict(schema.sdxl_unet(), None, dtype=torch.float16)
        om = opipe.OracleModel(sd, ou.sdxl_base_config(), None)
        del sd
        g = torch.Generator().manual_seed(s["prompt_seed"])
        cond, uncond = torch.randn(1, 77, 2048, generator=g), torch.randn(1, 77, 2048, generator=g)
        y, uy = torch.randn(1, 2816, generator=g), torch.randn(1, 2816, generator=g)
        t0 = time.time()
        lat = opipe.sample(om, cond.half().float(), uncond.half().float(), s["seeds"], s["steps"], "euler_a", s["cfg"], (128, 128),
                           y=y.half().float(), uy=uy.half().float())
        print(f"[c3_sdxl_e2e30] {time.time() - t0:.0f}s", flush=True)
        save("c3_sdxl_e2e30", final_latent=lat.numpy())
        del om
    return done


if __name__ == "__main__":
    main(sys.argv[1:] or list(SPEC))
