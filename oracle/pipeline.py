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
    "restart": ("karras", False), "lcm": (None, False),
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
           s_min_uncond=0.0):
    """Returns final latents (B,4,h,w) fp32.  ``init_latent`` switches to the img2img arithmetic
    (modules/sd_samplers_kdiffusion.py:134-143); ``mask`` (1 = keep the original latent) adds the inpainting blends of
    modules/sd_samplers_cfg_denoiser.py:186-187 / 292-293 and the final blend of modules/processing.py:1776-1784."""
    b = len(seeds)
    rng = ImageRNG((4, latent_hw[0], latent_hw[1]), seeds)
    x = rng.next()
    nmask = None if mask is None else 1.0 - mask

    def apply_model(xi, t, c):
        if y is not None:
            return model.apply_model(xi, t, c, y if xi.shape[0] == y.shape[0] else torch.cat([y, uy]))   # cond rows only: uncond skipped
        return model.apply_model(xi, t, c)

    def finish(samples):
        if mask is not None:
            samples = samples * nmask + init_latent * mask
        return samples

    if sampler in ("ddim", "plms", "ddim_cfgpp", "unipc"):
        # CFGDenoiserTimesteps: inner model is apply_model on integer timesteps, CFG combines eps.
        if parameterization == "v":
            inner = lambda xi, ti, ci: kd.timesteps_v_to_eps(model.alphas_cumprod, xi, ti, apply_model(xi, ti, ci))
        else:
            inner = lambda xi, ti, ci: apply_model(xi, ti, ci)
        cfg = kd.CFGDenoiser(inner, mask, nmask, init_latent)
        cfg.mask_before_denoising = True
        ts = kd.ddim_timesteps(steps)
        extra = dict(uncond=uncond, cond=cond, cond_scale=cfg_scale, s_min_uncond=s_min_uncond)
        if init_latent is not None:
            total, t_enc = kd.setup_img2img_steps(steps, denoising_strength, img2img_steps_given)
            ts = kd.ddim_timesteps(total)
            ac = model.alphas_cumprod
            x = init_latent * torch.sqrt(ac[ts[t_enc]]) + x * torch.sqrt(1 - ac[ts[t_enc]])       # sd_samplers_timesteps.py:103-107
            ts = ts[:t_enc]
        if sampler == "unipc":                   # default opts.uni_pc_* (modules/shared_options.py:402-405)
            from . import unipc
            return finish(unipc.sample_unipc(cfg, x, ts, model.alphas_cumprod, extra, callback=record, is_img2img=init_latent is not None))
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
    extra = dict(uncond=uncond, cond=cond, cond_scale=cfg_scale, s_min_uncond=s_min_uncond)
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
    if sampler == "lcm":
        return finish(kd.sample_lcm(cfg, x, sigmas, extra, rng.next, callback=record))
    if sampler == "restart":
        return finish(kd.restart_sampler(cfg, x, sigmas, extra, rng.next, callback=record, s_noise=s_noise))
    fn = {"euler": kd.sample_euler, "dpmpp_2m": kd.sample_dpmpp_2m, "heun": kd.sample_heun, "dpm_2": kd.sample_dpm_2,
          "lms": kd.sample_lms}.get(sampler)
    if fn is None:
        raise ValueError(sampler)
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
                  hr_scale=2.0, denoising_strength=0.75, mode="bilinear"):
    """txt2img + latent hires fix (modules/processing.py:1349-1464): first pass, F.interpolate of the latent (not decoded),
    fresh ImageRNG noise with the same seeds (:1429), second pass = sample_img2img with steps given (:1454)."""
    first = sample(model, cond, uncond, seeds, steps, sampler, cfg_scale, latent_hw)
    th, tw = int(latent_hw[0] * hr_scale), int(latent_hw[1] * hr_scale)
    up = torch.nn.functional.interpolate(first, size=(th, tw), mode=mode, antialias=False)
    return sample(model, cond, uncond, seeds, steps, sampler, cfg_scale, (th, tw), init_latent=up,
                  denoising_strength=denoising_strength, img2img_steps_given=True)
