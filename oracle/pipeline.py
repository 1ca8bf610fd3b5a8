"""End-to-end CPU restatement of the txt2img / img2img hot path (oracle; tests + bench cpu_baseline only).

Follows the call stack in SURVEY.md section 3.1: ``StableDiffusionProcessingTxt2Img.sample``
(modules/processing.py:1307-1346) -> ``KDiffusionSampler.sample`` (modules/sd_samplers_kdiffusion.py:190-234)
-> k-diffusion sampler loop -> ``CFGDenoiser.forward`` (modules/sd_samplers_cfg_denoiser.py:156-311) ->
``CompVisDenoiser`` -> UNet; then ``decode_latent_batch`` (modules/processing.py:625-672) and the uint8
conversion (:1004-1005, :1034-1035).  Everything runs in fp32 on CPU = the reference's CI configuration
``--use-cpu all --no-half --disable-opt-split-attention`` (.github/workflows/run_tests.yaml:44-56).
"""
from __future__ import annotations

import torch

from . import kdiffusion as kd
from .rng import ImageRNG
from .unet import UNetConfig, build_unet
from .vae import VAEConfig, build_vae, to_uint8_hwc


class OracleModel:
    def __init__(self, state_dict, unet_cfg: UNetConfig, vae_cfg: VAEConfig | None = None):
        self.unet = build_unet(unet_cfg, state_dict)
        self.vae = build_vae(vae_cfg, state_dict) if vae_cfg is not None else None
        ac = state_dict.get("alphas_cumprod")
        self.alphas_cumprod = ac.float() if ac is not None else kd.make_alphas_cumprod()

    def apply_model(self, x, t, cond, y=None):
        return self.unet(x, t, cond, y)


# sampler -> (default scheduler, discard_next_to_last_sigma): the options column of modules/sd_samplers_kdiffusion.py:11-27
SAMPLER_OPTIONS = {
    "euler_a": (None, False), "euler": (None, False), "lms": (None, False), "heun": (None, False),
    "dpmpp_2m": ("karras", False), "dpmpp_2s_a": ("karras", False), "dpm_2": ("karras", True), "dpm_2_a": ("karras", True),
    "restart": ("karras", False), "lcm": (None, False), "dpm_fast": (None, False), "dpm_adaptive": (None, False),
    "dpmpp_sde": ("karras", False), "dpmpp_2m_sde": ("exponential", False), "dpmpp_2m_sde_heun": ("exponential", False),
    "dpmpp_3m_sde": ("exponential", True),
}


def get_sigmas(model_wrap: kd.CompVisDenoiser, sampler: str, steps: int, scheduler: str = "automatic"):
    """modules/sd_samplers_kdiffusion.py:79-132 with default options (sigma_min/max/rho overrides off)."""
    from . import schedulers as osch
    default, discard = SAMPLER_OPTIONS[sampler]
    steps += 1 if discard else 0
    if scheduler == "automatic":
        scheduler = default
    if scheduler is None:
        sigmas = model_wrap.get_sigmas(steps)
    else:
        fn, need_inner = osch.SCHEDULERS[scheduler]
        smin, smax = model_wrap.sigmas[0].item(), model_wrap.sigmas[-1].item()
        sigmas = fn(steps, smin, smax, model_wrap) if need_inner else fn(steps, smin, smax)
    if discard:
        sigmas = torch.cat([sigmas[:-2], sigmas[-1:]])
    return sigmas


@torch.no_grad()
def sample(model: OracleModel, cond, uncond, seeds, steps, sampler="euler_a", cfg_scale=7.0,
           latent_hw=(64, 64), eta=None, s_noise=1.0, init_latent=None, denoising_strength=0.75,
           y=None, uy=None, record=None, img2img_steps_given=True, scheduler="automatic", mask=None, parameterization="eps",
           s_min_uncond=0.0, image_cond=None, image_cfg_scale=None, noise_multiplier=1.0, refiner=None, s_churn=0.0, s_tmin=0.0,
           s_tmax=float("inf"), unipc_options=None):
    """Returns final latents (B,4,h,w) fp32.  ``init_latent`` switches to the img2img arithmetic
    (modules/sd_samplers_kdiffusion.py:134-143); ``mask`` (1 = keep the original latent) adds the inpainting blends of
    modules/sd_samplers_cfg_denoiser.py:186-187 / 292-293 and the final blend of modules/processing.py:1776-1784."""
    b = len(seeds)
    rng = ImageRNG((4, latent_hw[0], latent_hw[1]), seeds)
    x = rng.next()
    if init_latent is not None and noise_multiplier != 1.0:      # modules/processing.py:1762-1764 (img2img only)
        x = x * noise_multiplier
    nmask = None if mask is None else 1.0 - mask

    def apply_model(xi, t, c, ic=None):
        if ic is not None:                       # inpainting / edit checkpoints: the UNet input is cat([x, c_concat], dim=1)
            xi = torch.cat([xi, ic], dim=1)
        if y is not None:
            return model.apply_model(xi, t, c, y if xi.shape[0] == y.shape[0] else torch.cat([y, uy]))   # cond rows only: uncond skipped
        return model.apply_model(xi, t, c)

    def _refiner(cfg, extra, wrap_cls=None):
        """p.refiner_checkpoint / p.refiner_switch_at (modules/processing.py:177-178, 882-885): the refiner's UNet behind the same
        kind of wrapper, with the conds computed for it (p.setup_conds after the reload, sd_samplers_common.py:198)."""
        if refiner is None:
            return
        rm = refiner["model"]
        am = lambda xi, t, c, ic=None: rm.apply_model(xi if ic is None else torch.cat([xi, ic], dim=1), t, c)
        inner = wrap_cls(am, rm.alphas_cumprod) if wrap_cls is not None else am
        cfg.refiner = dict(inner_model=inner, cond=refiner["cond"], uncond=refiner["uncond"], switch_at=refiner.get("switch_at"),
                           by_sample_steps=refiner.get("by_sample_steps", False), extra=extra)
        cfg.total_steps = refiner.get("total_steps", steps)

    def _edit(cfg):                              # InstructPix2Pix: cond_stage_key "edit" + p.image_cfg_scale (cfg_denoiser.py:166)
        if image_cfg_scale is not None:
            cfg.is_edit_cond_stage, cfg.image_cfg_scale = True, image_cfg_scale
            cfg.init_latent = init_latent if init_latent is not None else torch.zeros_like(x)

    def finish(samples):
        if mask is not None:
            samples = samples * nmask + init_latent * mask
        return samples

    if sampler in ("ddim", "plms", "ddim_cfgpp", "unipc"):
        # CFGDenoiserTimesteps: inner model is apply_model on integer timesteps, CFG combines eps.
        if parameterization == "v":
            inner = lambda xi, ti, ci, ic=None: kd.timesteps_v_to_eps(model.alphas_cumprod, xi, ti, apply_model(xi, ti, ci, ic))
        else:
            inner = lambda xi, ti, ci, ic=None: apply_model(xi, ti, ci, ic)
        cfg = kd.CFGDenoiser(inner, mask, nmask, init_latent)
        cfg.mask_before_denoising = True
        _edit(cfg)
        ts = kd.ddim_timesteps(steps)
        extra = dict(uncond=uncond, cond=cond, cond_scale=cfg_scale, s_min_uncond=s_min_uncond, image_cond=image_cond)
        _refiner(cfg, extra)
        if init_latent is not None:
            total, t_enc = kd.setup_img2img_steps(steps, denoising_strength, img2img_steps_given)
            ts = kd.ddim_timesteps(total)
            ac = model.alphas_cumprod
            x = init_latent * torch.sqrt(ac[ts[t_enc]]) + x * torch.sqrt(1 - ac[ts[t_enc]])       # sd_samplers_timesteps.py:103-107
            ts = ts[:t_enc]
        if sampler == "unipc":                   # default opts.uni_pc_* (modules/shared_options.py:402-405)
            from . import unipc
            return finish(unipc.sample_unipc(cfg, x, ts, model.alphas_cumprod, extra, callback=record, is_img2img=init_latent is not None,
                                             **(unipc_options or {})))      # opts.uni_pc_variant / _skip_type / _order / _lower_order_final
        if sampler == "plms":
            return finish(kd.sample_plms(cfg, x, ts, model.alphas_cumprod, extra, callback=record))
        if sampler == "ddim_cfgpp":
            return finish(kd.sample_ddim_cfgpp(cfg, x, ts, model.alphas_cumprod, extra, rng.next,
                                               eta=0.0 if eta is None else eta, callback=record))
        return finish(kd.sample_ddim(cfg, x, ts, model.alphas_cumprod, extra, rng.next,
                                     eta=0.0 if eta is None else eta, callback=record))

    if sampler == "lcm":                         # CFGDenoiserLCM: always the LCM eps wrapper (modules/sd_samplers_lcm.py:83-90)
        wrap = kd.LCMCompVisDenoiser(apply_model, model.alphas_cumprod)
    else:
        wrap = (kd.CompVisVDenoiser if parameterization == "v" else kd.CompVisDenoiser)(apply_model, model.alphas_cumprod)
    cfg = kd.CFGDenoiser(wrap, mask, nmask, init_latent)
    _edit(cfg)
    extra = dict(uncond=uncond, cond=cond, cond_scale=cfg_scale, s_min_uncond=s_min_uncond, image_cond=image_cond)
    _refiner(cfg, extra, type(wrap))
    if init_latent is None:
        sigmas = get_sigmas(wrap, sampler, steps, scheduler)
        x = x * sigmas[0]
    else:
        total, t_enc = kd.setup_img2img_steps(steps, denoising_strength, img2img_steps_given)
        sigmas = get_sigmas(wrap, sampler, total, scheduler)[total - t_enc - 1:]
        x = init_latent + x * sigmas[0]
    anc = dict(eta=1.0 if eta is None else eta, s_noise=s_noise, callback=record)
    if sampler == "euler_a":
        return finish(kd.sample_euler_ancestral(cfg, x, sigmas, extra, rng.next, **anc))
    if sampler == "dpm_2_a":
        return finish(kd.sample_dpm_2_ancestral(cfg, x, sigmas, extra, rng.next, **anc))
    if sampler == "dpmpp_2s_a":
        return finish(kd.sample_dpmpp_2s_ancestral(cfg, x, sigmas, extra, rng.next, **anc))
    if sampler == "dpm_fast":        # sigma range / budget as modules/sd_samplers_kdiffusion.py:155-161 (img2img) and :203-208 (txt2img)
        if init_latent is None:
            smin, smax, n = wrap.sigmas[0].item(), wrap.sigmas[-1].item(), steps
        else:
            smin, smax, n = sigmas[-2], sigmas[0], len(sigmas) - 1
        return finish(kd.sample_dpm_fast(cfg, x, smin, smax, n, extra, rng.next, eta=1.0 if eta is None else eta, s_noise=s_noise,
                                         callback=record))
    if sampler == "dpm_adaptive":    # default eta = opts.eta_ancestral = 1: noise from the job's ImageRNG (k-diffusion default_noise_sampler)
        smin, smax = (wrap.sigmas[0].item(), wrap.sigmas[-1].item()) if init_latent is None else (sigmas[-2], sigmas[0])
        return finish(kd.sample_dpm_adaptive(cfg, x, smin, smax, extra, lambda *a: rng.next(), eta=1.0 if eta is None else eta,
                                             s_noise=s_noise, callback=record))
    if sampler in ("dpmpp_sde", "dpmpp_2m_sde", "dpmpp_2m_sde_heun", "dpmpp_3m_sde"):
        # modules/sd_samplers_common.py:334-342: Brownian tree per image seed over the positive range of the FULL schedule
        from .brownian import BrownianTreeNoiseSampler
        full = sigmas if init_latent is None else get_sigmas(wrap, sampler, total, scheduler)
        ns = BrownianTreeNoiseSampler(x, full[full > 0].min(), full.max(), seed=list(seeds))
        if sampler == "dpmpp_sde":
            return finish(kd.sample_dpmpp_sde(cfg, x, sigmas, extra, ns, **anc))
        if sampler == "dpmpp_3m_sde":
            return finish(kd.sample_dpmpp_3m_sde(cfg, x, sigmas, extra, ns, **anc))
        return finish(kd.sample_dpmpp_2m_sde(cfg, x, sigmas, extra, ns, solver_type="heun" if sampler.endswith("heun") else "midpoint", **anc))
    if sampler == "lcm":
        return finish(kd.sample_lcm(cfg, x, sigmas, extra, rng.next, callback=record))
    if sampler == "restart":
        return finish(kd.restart_sampler(cfg, x, sigmas, extra, rng.next, callback=record, s_noise=s_noise))
    fn = {"euler": kd.sample_euler, "dpmpp_2m": kd.sample_dpmpp_2m, "heun": kd.sample_heun, "dpm_2": kd.sample_dpm_2,
          "lms": kd.sample_lms}.get(sampler)
    if fn is None:
        raise ValueError(sampler)
    if sampler in ("euler", "heun", "dpm_2") and s_churn:     # opts.s_churn / s_tmin / s_tmax / s_noise (sd_samplers_kdiffusion.py:164-183)
        return finish(fn(cfg, x, sigmas, extra, noise_fn=rng.next, callback=record, s_churn=s_churn, s_tmin=s_tmin, s_tmax=s_tmax, s_noise=s_noise))
    return finish(fn(cfg, x, sigmas, extra, callback=record))


@torch.no_grad()
def decode(model: OracleModel, latents):
    """decode_latent_batch: one image at a time (modules/processing.py:631-632)."""
    return torch.stack([model.vae.decode_first_stage(latents[i:i + 1])[0] for i in range(latents.shape[0])])


@torch.no_grad()
def txt2img(model: OracleModel, cond, uncond, seeds, steps, sampler="euler_a", cfg_scale=7.0,
            latent_hw=(64, 64), **kw):
    lat = sample(model, cond, uncond, seeds, steps, sampler, cfg_scale, latent_hw, **kw)
    img = decode(model, lat)
    return lat, img, to_uint8_hwc(img)


@torch.no_grad()
def txt2img_hires(model: OracleModel, cond, uncond, seeds, steps, sampler="euler_a", cfg_scale=7.0, latent_hw=(64, 64),
                  hr_scale=2.0, denoising_strength=0.75, mode="bilinear", antialias=False):
    """txt2img + latent hires fix (modules/processing.py:1349-1464): first pass, F.interpolate of the latent (not decoded; mode and
    antialias flag of the "Latent*" upscaler, modules/shared.py:55-62), fresh ImageRNG noise with the same seeds (:1429), second pass =
    sample_img2img with steps given (:1454)."""
    first = sample(model, cond, uncond, seeds, steps, sampler, cfg_scale, latent_hw)
    th, tw = int(latent_hw[0] * hr_scale), int(latent_hw[1] * hr_scale)
    up = torch.nn.functional.interpolate(first, size=(th, tw), mode=mode, antialias=antialias)
    return sample(model, cond, uncond, seeds, steps, sampler, cfg_scale, (th, tw), init_latent=up,
                  denoising_strength=denoising_strength, img2img_steps_given=True)


# ---- image conditioning of inpainting / edit checkpoints (c_concat) -------------------------------------------------------
@torch.no_grad()
def txt2img_image_conditioning(model: OracleModel, batch, height, width):
    """modules/processing.py:100-111 for conditioning_key hybrid / concat: the "masked image" is all 0.5 (everything masked),
    encoded (images_tensor_to_samples, modules/sd_samplers_common.py:96-113: image * 2 - 1 -> first stage; posterior mean
    here), with a mask channel of ones in front."""
    image = torch.ones(batch, 3, height, width) * 0.5
    latent = model.vae.encode_first_stage_mean(image * 2 - 1)
    return torch.nn.functional.pad(latent, (0, 0, 0, 0, 1, 0), value=1.0)


@torch.no_grad()
def inpainting_image_conditioning(model: OracleModel, source_image, latent_hw, image_mask=None, mask_weight=1.0, round_image_mask=True):
    """modules/processing.py:332-374.  ``source_image`` [B,3,H,W] in [-1,1]; ``image_mask`` [1,1,H,W] in [0,1] (1 = repaint)."""
    if image_mask is not None:
        conditioning_mask = torch.round(image_mask) if round_image_mask else image_mask
    else:
        conditioning_mask = source_image.new_ones(1, 1, *source_image.shape[-2:])
    conditioning_image = torch.lerp(source_image, source_image * (1.0 - conditioning_mask), mask_weight)
    conditioning_image = model.vae.encode_first_stage_mean(conditioning_image)
    conditioning_mask = torch.nn.functional.interpolate(conditioning_mask, size=latent_hw)
    conditioning_mask = conditioning_mask.expand(conditioning_image.shape[0], -1, -1, -1)
    return torch.cat([conditioning_mask, conditioning_image], dim=1)


@torch.no_grad()
def edit_image_conditioning(model: OracleModel, source_image):
    """modules/processing.py:321-324: the UNSCALED posterior mode of the source image."""
    mean, _ = torch.chunk(model.vae.encode_moments(source_image), 2, dim=1)
    return mean


# ---- hires fix through an image-space upscaler ----------------------------------------------------------------------------
def hires_target_resolution(width, height, hr_scale=2.0, hr_resize_x=0, hr_resize_y=0, opt_f=8):
    """modules/processing.py:1221-1250 -> (upscale_to_x, upscale_to_y, truncate_x, truncate_y)."""
    tx = ty = 0
    if hr_resize_x == 0 and hr_resize_y == 0:
        ux, uy = int(width * hr_scale), int(height * hr_scale)
    elif hr_resize_y == 0:
        ux, uy = hr_resize_x, hr_resize_x * height // width
    elif hr_resize_x == 0:
        ux, uy = hr_resize_y * width // height, hr_resize_y
    else:
        if width / height < hr_resize_x / hr_resize_y:
            ux, uy = hr_resize_x, hr_resize_x * height // width
        else:
            ux, uy = hr_resize_y * width // height, hr_resize_y
        tx, ty = (ux - hr_resize_x) // opt_f, (uy - hr_resize_y) // opt_f
    return ux, uy, tx, ty


def resize_image_mode0(im, width, height, upscaler_name=None):
    """images.resize_image(0, ...) (modules/images.py:252-291) over the built-in PIL scalers (modules/upscaler.py:54-76, 107-154);
    pinned by tests/golden/resize_image.npz."""
    from PIL import Image
    lanczos, nearest = Image.Resampling.LANCZOS, Image.Resampling.NEAREST
    if upscaler_name is None or upscaler_name == "None" or im.mode == 'L':
        return im.resize((width, height), resample=lanczos)
    scale = max(width / im.width, height / im.height)
    if scale > 1.0:
        kind = {"Lanczos": lanczos, "Nearest": nearest}[upscaler_name]
        dest_w, dest_h = int((im.width * scale) // 8 * 8), int((im.height * scale) // 8 * 8)
        for i in range(3):
            if im.width >= dest_w and im.height >= dest_h and (i > 0 or scale != 1):
                break
            shape = (im.width, im.height)
            im = im.resize((int(im.width * scale), int(im.height * scale)), resample=kind)
            if shape == (im.width, im.height):
                break
        if im.width != dest_w or im.height != dest_h:
            im = im.resize((dest_w, dest_h), resample=lanczos)
    if im.width != width or im.height != height:
        im = im.resize((width, height), resample=lanczos)
    return im


@torch.no_grad()
def txt2img_hires_image(model: OracleModel, cond, uncond, seeds, steps, sampler="euler_a", cfg_scale=7.0, width=64, height=64,
                        hr_scale=2.0, hr_resize_x=0, hr_resize_y=0, denoising_strength=0.75, upscaler="Lanczos", opt_f=8):
    """txt2img + hires fix with an image-space upscaler (modules/processing.py:1353-1354, 1400-1427): the first pass is decoded,
    converted to uint8 PIL images, resized, re-encoded (posterior mean) and truncated to the requested size."""
    import numpy as np
    from PIL import Image
    first = sample(model, cond, uncond, seeds, steps, sampler, cfg_scale, (height // opt_f, width // opt_f))
    ux, uy, tx, ty = hires_target_resolution(width, height, hr_scale, hr_resize_x, hr_resize_y, opt_f)
    u8 = to_uint8_hwc(decode(model, first))
    batch = [np.moveaxis(np.array(resize_image_mode0(Image.fromarray(im), ux, uy, upscaler)).astype(np.float32) / 255.0, 2, 0) for im in u8]
    decoded = torch.from_numpy(np.array(batch))
    up = model.vae.encode_first_stage_mean(decoded * 2 - 1)
    up = up[:, :, ty // 2:up.shape[2] - (ty + 1) // 2, tx // 2:up.shape[3] - (tx + 1) // 2]
    return sample(model, cond, uncond, seeds, steps, sampler, cfg_scale, tuple(up.shape[-2:]), init_latent=up,
                  denoising_strength=denoising_strength, img2img_steps_given=True)
