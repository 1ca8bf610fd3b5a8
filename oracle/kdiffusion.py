"""fp32 CPU restatement of the sampler layer (oracle; tests only).

Third-party arithmetic restated from its published algorithm, anchored on the reference's call sites:
  crowsonkb/k-diffusion @ ab527a9a6d347f364e3d185ba6d714e22d80cb3c (pinned modules/launch_utils.py:349-357)
    external.DiscreteSchedule / DiscreteEpsDDPMDenoiser / CompVisDenoiser
        -> constructed at modules/sd_samplers_kdiffusion.py:53-64
    sampling.get_sigmas_karras, get_ancestral_step, sample_euler, sample_euler_ancestral,
    sampling.sample_dpmpp_2m  -> selected by the table at modules/sd_samplers_kdiffusion.py:11-27
    sampling.to_d             -> overridden in-tree at modules/sd_schedulers.py:10-15 (followed)
  ldm "linear" beta schedule  -> configs/v1-inference.yaml:5-9, restated in-tree at
                                 modules/models/diffusion/ddpm_edit.py:133-154
  DDIM                        -> modules/sd_samplers_timesteps_impl.py:12-40 (in-repo; pinned by fixture)
  DDIM timesteps              -> modules/sd_samplers_timesteps.py:86-96
  img2img step arithmetic     -> modules/sd_samplers_common.py:22-31, sd_samplers_kdiffusion.py:134-143
Known answers checked in tests: sigma_min 0.0291672, sigma_max 14.614641, get_sigmas(20) table,
Karras rho=7 n=50 endpoints (hints at modules/shared_options.py:396-397).
"""
from __future__ import annotations

import math

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------
# DDPM schedule
# ---------------------------------------------------------------------------------------------
def make_alphas_cumprod(linear_start=0.00085, linear_end=0.0120, n=1000) -> torch.Tensor:
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2
    alphas = 1.0 - betas.numpy()
    return torch.tensor(np.cumprod(alphas, axis=0), dtype=torch.float32)


def rescale_zero_terminal_snr_abar(alphas_cumprod):
    """modules/sd_models.py:628-644 (pinned by tests/golden/zsnr.npz): shift sqrt(alpha_bar) to end at zero, rescale to keep its
    first value, square, and set the last entry to the reference's constant."""
    s = alphas_cumprod.sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = s - sT
    s = s * (s0 / (s0 - sT))
    out = s ** 2
    out[-1] = 4.8973451890853435e-08
    return out


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def append_dims(x, n):
    return x[(...,) + (None,) * (n - x.ndim)]


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return append_zero(sigmas)


class DiscreteSchedule:
    def __init__(self, sigmas: torch.Tensor, quantize: bool = False):
        self.sigmas = sigmas
        self.log_sigmas = sigmas.log()
        self.quantize = quantize

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def get_sigmas(self, n=None):
        if n is None:
            return append_zero(self.sigmas.flip(0))
        t_max = len(self.sigmas) - 1
        t = torch.linspace(t_max, 0, n)
        return append_zero(self.t_to_sigma(t))

    def sigma_to_t(self, sigma, quantize=None):
        quantize = self.quantize if quantize is None else quantize
        log_sigma = sigma.log()
        dists = log_sigma - self.log_sigmas[:, None]
        if quantize:
            return dists.abs().argmin(dim=0).view(sigma.shape)
        low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=self.log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = self.log_sigmas[low_idx], self.log_sigmas[high_idx]
        w = (low - log_sigma) / (low - high)
        w = w.clamp(0, 1)
        t = (1 - w) * low_idx + w * high_idx
        return t.view(sigma.shape)

    def t_to_sigma(self, t):
        t = t.float()
        low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
        log_sigma = (1 - w) * self.log_sigmas[low_idx] + w * self.log_sigmas[high_idx]
        return log_sigma.exp()


class CompVisDenoiser(DiscreteSchedule):
    """eps-prediction wrapper: denoised = x + eps(x*c_in, t(sigma)) * c_out, c_out=-sigma, c_in=1/sqrt(sigma^2+1)."""

    def __init__(self, apply_model, alphas_cumprod, quantize=False):
        super().__init__(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5, quantize)
        self.apply_model = apply_model     # callable(x, t, cond) -> eps
        self.sigma_data = 1.0

    def get_scalings(self, sigma):
        c_out = -sigma
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_out, c_in

    def _eps(self, x_scaled, t, cond, image_cond):
        """apply_model; ``image_cond`` is the c_concat the model concatenates to its (already scaled) input along the
        channels (ldm DiffusionWrapper, conditioning_key "hybrid" / "concat"; fed from sd_samplers_cfg_denoiser.py:193-209)."""
        if image_cond is None:
            return self.apply_model(x_scaled, t, cond)
        return self.apply_model(x_scaled, t, cond, image_cond)

    def __call__(self, input, sigma, cond, image_cond=None):
        c_out, c_in = [append_dims(x, input.ndim) for x in self.get_scalings(sigma)]
        eps = self._eps(input * c_in, self.sigma_to_t(sigma), cond, image_cond)
        return input + eps * c_out


class CompVisVDenoiser(CompVisDenoiser):
    """v-prediction wrapper (k-diffusion external.DiscreteVDDPMDenoiser / CompVisVDenoiser, [3P]; constructed at
    modules/sd_samplers_kdiffusion.py:60-62 for parameterization == "v"): denoised = v(x*c_in, t) * c_out + x * c_skip."""

    def get_scalings(self, sigma):
        c_skip = self.sigma_data ** 2 / (sigma ** 2 + self.sigma_data ** 2)
        c_out = -sigma * self.sigma_data / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out, c_in

    def __call__(self, input, sigma, cond, image_cond=None):
        c_skip, c_out, c_in = [append_dims(x, input.ndim) for x in self.get_scalings(sigma)]
        return self._eps(input * c_in, self.sigma_to_t(sigma), cond, image_cond) * c_out + input * c_skip


class LCMCompVisDenoiser(CompVisDenoiser):
    """modules/sd_samplers_lcm.py:10-63 (pinned by tests/golden/lcm.npz): the eps wrapper over the 50 "original" LCM timesteps
    (every 20th alpha, counted back from t = 999) and the consistency-model boundary scaling of its output."""

    def __init__(self, apply_model, alphas_cumprod):
        timesteps, original = 1000, 50
        self.skip_steps = timesteps // original
        valid = torch.zeros((original,), dtype=torch.float32)
        for k in range(original):
            valid[original - 1 - k] = alphas_cumprod[timesteps - 1 - k * self.skip_steps]
        super().__init__(apply_model, valid, quantize=None)

    def get_sigmas(self, n=None):
        if n is None:
            return append_zero(self.sigmas.flip(0))
        start, end = self.sigma_to_t(self.sigma_max), self.sigma_to_t(self.sigma_min)
        return append_zero(self.t_to_sigma(torch.linspace(start, end, n)))

    def sigma_to_t(self, sigma, quantize=None):
        dists = sigma.log() - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape) * self.skip_steps + (self.skip_steps - 1)

    def t_to_sigma(self, timestep):
        t = torch.clamp(((timestep - (self.skip_steps - 1)) / self.skip_steps).float(), min=0, max=(len(self.sigmas) - 1))
        return super().t_to_sigma(t)

    def get_scaled_out(self, sigma, output, input):
        sigma_data = 0.5
        scaled_timestep = append_dims(self.sigma_to_t(sigma), output.ndim) * 10.0
        c_skip = sigma_data ** 2 / (scaled_timestep ** 2 + sigma_data ** 2)
        c_out = scaled_timestep / (scaled_timestep ** 2 + sigma_data ** 2) ** 0.5
        return c_out * output + c_skip * input

    def __call__(self, input, sigma, cond, image_cond=None):
        c_out, c_in = [append_dims(x, input.ndim) for x in self.get_scalings(sigma)]
        eps = self._eps(input * c_in, self.sigma_to_t(sigma), cond, image_cond)
        return self.get_scaled_out(sigma, input + eps * c_out, input)


def timesteps_v_to_eps(alphas_cumprod, x_t, t, v):
    """CompVisTimestepsVDenoiser.predict_eps_from_z_and_v (modules/sd_samplers_timesteps.py:38-39)."""
    return torch.sqrt(alphas_cumprod)[t.to(torch.int), None, None, None] * v + \
        torch.sqrt(1 - alphas_cumprod)[t.to(torch.int), None, None, None] * x_t


# ---------------------------------------------------------------------------------------------
# CFG  (modules/sd_samplers_cfg_denoiser.py:74-82, 156-311 for the plain txt2img case:
#       one cond per image with weight 1.0, equal token counts, batch_cond_uncond on, no mask)
# ---------------------------------------------------------------------------------------------
def refiner_due(step, total_steps, sigma, sigmas_table, switch_at, has_refiner, on_refiner, by_sample_steps=False, enable_hr=False,
                is_hr_pass=False, hires_fix_refiner_pass="second pass"):
    """The decision part of apply_refiner (modules/sd_samplers_common.py:158-190; pinned by tests/golden/refiner.npz): progress
    is measured in model timesteps — the table entry nearest to the current sigma, or the timestep itself for the DDIM family
    (``sigmas_table`` None) — unless opts.refiner_switch_by_sample_steps; then the switch point, the checkpoint identity and the
    hires-pass option are checked."""
    if by_sample_steps or sigma is None:
        completed_ratio = step / total_steps
    else:
        if sigmas_table is not None:
            timestep = torch.argmin(torch.abs(sigmas_table - torch.max(sigma)))
        else:
            timestep = torch.max(sigma).to(dtype=int)
        completed_ratio = (999 - timestep) / 1000
    if switch_at is not None and completed_ratio < switch_at:
        return False
    if not has_refiner or on_refiner:
        return False
    if enable_hr:
        if hires_fix_refiner_pass == "first pass" and is_hr_pass:
            return False
        if hires_fix_refiner_pass == "second pass" and not is_hr_pass:
            return False
    return True


class CFGDenoiser:
    """modules/sd_samplers_cfg_denoiser.py:35-311, pinned by tests/golden/cfg_denoiser.npz (the reference class executed over
    twenty scenarios).  ``cond`` is a tensor (one prompt of weight 1 per image) or the (conds_list, tensor) pair that
    prompt_parser.reconstruct_multicond_batch returns (AND composition, :169); ``inner_model(x_in, sigma_in, cond_in)`` — with a
    fourth argument, the per-row image conditioning, when ``image_cond`` is given.  The reference's options are attributes.
    Not restated: script callbacks, refiner switch, live previews, opts.batch_cond_uncond = False (same rows, smaller calls)."""

    def __init__(self, inner_model, mask=None, nmask=None, init_latent=None):
        self.inner_model = inner_model
        self.step = 0
        self.total_steps = None
        self.mask, self.nmask, self.init_latent = mask, nmask, init_latent
        self.mask_before_denoising = False
        self.cond_scale_miltiplier = 1.0                     # :61-64
        self.need_last_noise_uncond = False
        self.last_noise_uncond = None
        self.image_cfg_scale = None                          # with is_edit_model: InstructPix2Pix (:166)
        self.is_edit_cond_stage = False                      # shared.sd_model.cond_stage_key == "edit"
        self.skip_early_cond = 0.0                           # opts.skip_early_cond
        self.s_min_uncond_all = False                        # opts.s_min_uncond_all
        self.pad_cond_uncond = False                         # opts.pad_cond_uncond (needs empty_prompt)
        self.pad_cond_uncond_v0 = False                      # opts.pad_cond_uncond_v0
        self.empty_prompt = None                             # shared.sd_model.cond_stage_model_empty_prompt
        self.padded_cond_uncond = self.padded_cond_uncond_v0 = False
        self.skipped_uncond = False
        self.adm = False                                     # shared.sd_model.model.conditioning_key == "crossattn-adm" (unCLIP, :192-194)
        self.refiner = None                                  # dict(inner_model=, cond=, uncond=, switch_at=[, extra=the sampler's dict])
        self.on_refiner = False

    def apply_refiner(self, sigma):
        """modules/sd_samplers_common.py:158-202 + CFGDenoiser.update_inner_model (cfg_denoiser.py:93-98): from the switch point on
        the refiner checkpoint's wrapped model and its own conds replace the base model's; the sampler loop's extra_args dict is
        updated in place, so later steps pass the new conds by themselves."""
        r = self.refiner
        table = getattr(self.inner_model, "sigmas", None)
        if r is None or not refiner_due(self.step, self.total_steps, sigma, table, r.get("switch_at"), True, self.on_refiner,
                                        r.get("by_sample_steps", False)):
            return False
        self.inner_model, self.on_refiner = r["inner_model"], True
        if r.get("extra") is not None:
            r["extra"]["cond"], r["extra"]["uncond"] = r["cond"], r["uncond"]
        return True

    @staticmethod
    def combine_denoised(x_out, conds_list, uncond_n, cond_scale):                       # :73-82
        denoised_uncond = x_out[-uncond_n:]
        denoised = torch.clone(denoised_uncond)
        for i, conds in enumerate(conds_list):
            for cond_index, weight in conds:
                denoised[i] += (x_out[cond_index] - denoised_uncond[i]) * (weight * cond_scale)
        return denoised

    def combine_denoised_for_edit_model(self, x_out, cond_scale):                         # :84-88
        out_cond, out_img_cond, out_uncond = x_out.chunk(3)
        return out_uncond + cond_scale * (out_cond - out_img_cond) + self.image_cfg_scale * (out_img_cond - out_uncond)

    def _pad(self, tensor, uncond):                                                       # :100-155
        if self.pad_cond_uncond_v0:
            if uncond.shape[1] < tensor.shape[1]:
                uncond = torch.hstack([uncond, uncond[:, -1:].repeat([1, tensor.shape[1] - uncond.shape[1], 1])])
            else:
                uncond = uncond[:, :tensor.shape[1]]
            self.padded_cond_uncond_v0 = True
        elif self.pad_cond_uncond:
            empty = self.empty_prompt
            n = (tensor.shape[1] - uncond.shape[1]) // empty.shape[1]
            if n < 0:
                tensor = torch.cat([tensor, empty.repeat((tensor.shape[0], -n, 1))], axis=1)
            elif n > 0:
                uncond = torch.cat([uncond, empty.repeat((uncond.shape[0], n, 1))], axis=1)
            self.padded_cond_uncond = n != 0
        return tensor, uncond

    def __call__(self, x, sigma, uncond, cond, cond_scale, s_min_uncond=0.0, image_cond=None):
        b = x.shape[0]
        if self.apply_refiner(sigma):                                                     # :160-162
            cond, uncond = self.refiner["cond"], self.refiner["uncond"]
        conds_list, tensor = cond if isinstance(cond, tuple) else ([[(i, 1.0)] for i in range(b)], cond)
        is_edit = self.is_edit_cond_stage and self.image_cfg_scale is not None and self.image_cfg_scale != 1.0
        assert not is_edit or all(len(c) == 1 for c in conds_list)
        if self.mask_before_denoising and self.mask is not None:                          # :186-187
            x = x * self.nmask + self.init_latent * self.mask
        repeats = [len(c) for c in conds_list]
        rep = lambda t: torch.cat([torch.stack([t[i] for _ in range(n)]) for i, n in enumerate(repeats)])
        tail = [x, x] if is_edit else [x]
        x_in = torch.cat([rep(x)] + tail)                                                 # :203-209
        sigma_in = torch.cat([rep(sigma)] + [sigma] * len(tail))
        image_cond_in = None
        if image_cond is not None:
            image_uncond = torch.zeros_like(image_cond) if self.adm else image_cond      # unCLIP: c_adm of the uncond rows is zero
            image_cond_in = torch.cat([rep(image_cond), image_uncond] + ([torch.zeros_like(self.init_latent)] if is_edit else []))
        skip_uncond = False                                                               # :218-230
        if self.skip_early_cond != 0. and self.step / self.total_steps <= self.skip_early_cond:
            skip_uncond = True
        elif (self.step % 2 or self.s_min_uncond_all) and s_min_uncond > 0 and sigma[0] < s_min_uncond and not is_edit:
            skip_uncond = True
        self.skipped_uncond = skip_uncond
        if skip_uncond:
            x_in, sigma_in = x_in[:-b], sigma_in[:-b]
        self.padded_cond_uncond = self.padded_cond_uncond_v0 = False
        if tensor.shape[1] != uncond.shape[1] and (self.pad_cond_uncond_v0 or self.pad_cond_uncond):
            tensor, uncond = self._pad(tensor, uncond)

        def run(a, e, c):
            if image_cond_in is None:
                return self.inner_model(x_in[a:e], sigma_in[a:e], c)
            return self.inner_model(x_in[a:e], sigma_in[a:e], c, image_cond_in[a:e])

        if tensor.shape[1] == uncond.shape[1] or skip_uncond:                              # :240-246
            cond_in = torch.cat([tensor, uncond, uncond]) if is_edit else tensor if skip_uncond else torch.cat([tensor, uncond])
            x_out = run(0, x_in.shape[0], cond_in)
        else:                                                                             # :253-268: token counts differ
            x_out = torch.zeros_like(x_in)
            n = tensor.shape[0]
            x_out[:n] = run(0, n, tensor)
            x_out[-b:] = run(x_in.shape[0] - b, x_in.shape[0], uncond)
        image_index = [c[0][0] for c in conds_list]
        if skip_uncond:                                                                   # :270-273
            x_out = torch.cat([x_out, torch.cat([x_out[i:i + 1] for i in image_index])])
        if self.need_last_noise_uncond:
            self.last_noise_uncond = torch.clone(x_out[-b:])                              # :281-282
        if is_edit:
            denoised = self.combine_denoised_for_edit_model(x_out, cond_scale * self.cond_scale_miltiplier)
        elif skip_uncond:
            denoised = self.combine_denoised(x_out, conds_list, b, 1.0)
        else:
            denoised = self.combine_denoised(x_out, conds_list, b, cond_scale * self.cond_scale_miltiplier)
        if not self.mask_before_denoising and self.mask is not None:                      # :292-293
            denoised = denoised * self.nmask + self.init_latent * self.mask
        self.last_latent = torch.cat([x_out[i:i + 1] for i in image_index])               # base get_pred_x0 = x_out (:90-91)
        self.step += 1
        return denoised


# ---------------------------------------------------------------------------------------------
# samplers (noise_fn() returns the next per-image noise tensor = TorchHijack.randn_like,
#           modules/sd_samplers_common.py:205-226)
# ---------------------------------------------------------------------------------------------
def to_d(x, sigma, denoised):
    return (x - denoised) / sigma            # modules/sd_schedulers.py:10-15


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def sample_euler_ancestral(model, x, sigmas, extra_args, noise_fn, eta=1.0, s_noise=1.0, callback=None):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta=eta)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        d = to_d(x, sigmas[i], denoised)
        dt = sigma_down - sigmas[i]
        x = x + d * dt
        if sigmas[i + 1] > 0:
            x = x + noise_fn() * s_noise * sigma_up
    return x


class DPMSolver:
    """k-diffusion sampling.DPMSolver (third-party, crowsonkb/k-diffusion @ ab527a9 — restated from the published code, not
    pinned; call site modules/sd_samplers_kdiffusion.py:24): DPM-Solver-1/2/3 steps in t = -log(sigma) on the eps prediction
    eps = (x - denoised) / sigma, with a per-step cache so the first evaluation is shared by the orders."""

    def __init__(self, model, extra_args=None, info_callback=None):
        self.model, self.extra_args, self.info_callback = model, ({} if extra_args is None else extra_args), info_callback

    @staticmethod
    def t(sigma):
        return -sigma.log()

    @staticmethod
    def sigma(t):
        return t.neg().exp()

    def eps(self, eps_cache, key, x, t):
        if key in eps_cache:
            return eps_cache[key], eps_cache
        sigma = self.sigma(t) * x.new_ones([x.shape[0]])
        eps = (x - self.model(x, sigma, **self.extra_args)) / self.sigma(t)
        return eps, {key: eps, **eps_cache}

    def dpm_solver_1_step(self, x, t, t_next, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, 'eps', x, t)
        return x - self.sigma(t_next) * h.expm1() * eps, eps_cache

    def dpm_solver_2_step(self, x, t, t_next, r1=1 / 2, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, 'eps', x, t)
        s1 = t + r1 * h
        u1 = x - self.sigma(s1) * (r1 * h).expm1() * eps
        eps_r1, eps_cache = self.eps(eps_cache, 'eps_r1', u1, s1)
        x_2 = x - self.sigma(t_next) * h.expm1() * eps - self.sigma(t_next) / (2 * r1) * h.expm1() * (eps_r1 - eps)
        return x_2, eps_cache

    def dpm_solver_3_step(self, x, t, t_next, r1=1 / 3, r2=2 / 3, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, 'eps', x, t)
        s1, s2 = t + r1 * h, t + r2 * h
        u1 = x - self.sigma(s1) * (r1 * h).expm1() * eps
        eps_r1, eps_cache = self.eps(eps_cache, 'eps_r1', u1, s1)
        u2 = x - self.sigma(s2) * (r2 * h).expm1() * eps - self.sigma(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1) * (eps_r1 - eps)
        eps_r2, eps_cache = self.eps(eps_cache, 'eps_r2', u2, s2)
        x_3 = x - self.sigma(t_next) * h.expm1() * eps - self.sigma(t_next) / r2 * (h.expm1() / h - 1) * (eps_r2 - eps)
        return x_3, eps_cache

    def dpm_solver_fast(self, x, t_start, t_end, nfe, noise_fn, eta=0., s_noise=1.):
        m = math.floor(nfe / 3) + 1
        ts = torch.linspace(t_start, t_end, m + 1)
        orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
        for i in range(len(orders)):
            eps_cache = {}
            t, t_next = ts[i], ts[i + 1]
            if eta:
                sd, su = get_ancestral_step(self.sigma(t), self.sigma(t_next), eta)
                t_next_ = torch.minimum(t_end, self.t(sd))
                su = (self.sigma(t_next) ** 2 - self.sigma(t_next_) ** 2) ** 0.5
            else:
                t_next_, su = t_next, 0.
            eps, eps_cache = self.eps(eps_cache, 'eps', x, t)
            denoised = x - self.sigma(t) * eps
            if self.info_callback is not None:
                self.info_callback({'x': x, 'i': i, 't': ts[i], 't_up': t, 'denoised': denoised})
            if orders[i] == 1:
                x, eps_cache = self.dpm_solver_1_step(x, t, t_next_, eps_cache=eps_cache)
            elif orders[i] == 2:
                x, eps_cache = self.dpm_solver_2_step(x, t, t_next_, eps_cache=eps_cache)
            else:
                x, eps_cache = self.dpm_solver_3_step(x, t, t_next_, eps_cache=eps_cache)
            x = x + su * s_noise * noise_fn()
        return x


def sample_dpm_fast(model, x, sigma_min, sigma_max, n, extra_args, noise_fn, eta=0., s_noise=1., callback=None):
    """k-diffusion sample_dpm_fast: DPM-Solver-Fast with a fixed budget of n model evaluations between sigma_max and sigma_min
    (modules/sd_samplers_kdiffusion.py:203-210 passes the wrapped model's sigma range and n = steps)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError('sigma_min and sigma_max must not be 0')
    solver = DPMSolver(model, extra_args)
    if callback is not None:
        solver.info_callback = lambda info: callback({'sigma': solver.sigma(info['t']), 'sigma_hat': solver.sigma(info['t_up']), **info})
    return solver.dpm_solver_fast(x, solver.t(torch.tensor(sigma_max)), solver.t(torch.tensor(sigma_min)), n, noise_fn, eta, s_noise)


class PIDStepSizeController:
    """k-diffusion sampling.PIDStepSizeController (restated; unpinned): PID control of the log step size on the inverse error."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        self.h = h
        self.b1 = (pcoeff + icoeff + dcoeff) / order
        self.b2 = -(pcoeff + 2 * dcoeff) / order
        self.b3 = dcoeff / order
        self.accept_safety = accept_safety
        self.eps = eps
        self.errs = []

    @staticmethod
    def limiter(x):
        return 1 + math.atan(x - 1)

    def propose_step(self, error):
        inv_error = 1 / (float(error) + self.eps)
        if not self.errs:
            self.errs = [inv_error, inv_error, inv_error]
        self.errs[0] = inv_error
        factor = self.errs[0] ** self.b1 * self.errs[1] ** self.b2 * self.errs[2] ** self.b3
        factor = self.limiter(factor)
        accept = factor >= self.accept_safety
        if accept:
            self.errs[2] = self.errs[1]
            self.errs[1] = self.errs[0]
        self.h *= factor
        return accept


def dpm_solver_adaptive(solver: DPMSolver, x, t_start, t_end, noise_sampler, order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0.,
                        icoeff=1., dcoeff=0., accept_safety=0.81, eta=0., s_noise=1.):
    """k-diffusion DPMSolver.dpm_solver_adaptive (restated; unpinned): embedded pairs DPM-Solver-1/2 (order 2) or 2/3 (order 3),
    mixed absolute / relative error norm, PID step-size controller; with eta the accepted step lands on the ancestral sigma_down
    and fresh noise over [sigma(s), sigma(t)] is added."""
    if order not in {2, 3}:
        raise ValueError('order should be 2 or 3')
    forward = t_end > t_start
    if not forward and eta:
        raise ValueError('eta must be 0 for reverse sampling')
    h_init = abs(h_init) * (1 if forward else -1)
    atol, rtol = torch.tensor(atol), torch.tensor(rtol)
    s = t_start
    x_prev = x
    pid = PIDStepSizeController(h_init, pcoeff, icoeff, dcoeff, 1.5 if eta else order, accept_safety)
    info = {'steps': 0, 'nfe': 0, 'n_accept': 0, 'n_reject': 0}
    while (s < t_end - 1e-5) if forward else (s > t_end + 1e-5):
        eps_cache = {}
        t = torch.minimum(t_end, s + pid.h) if forward else torch.maximum(t_end, s + pid.h)
        if eta:
            sd, su = get_ancestral_step(solver.sigma(s), solver.sigma(t), eta)
            t_ = torch.minimum(t_end, solver.t(sd))
            su = (solver.sigma(t) ** 2 - solver.sigma(t_) ** 2) ** 0.5
        else:
            t_, su = t, 0.
        eps, eps_cache = solver.eps(eps_cache, 'eps', x, s)
        denoised = x - solver.sigma(s) * eps
        if order == 2:
            x_low, eps_cache = solver.dpm_solver_1_step(x, s, t_, eps_cache=eps_cache)
            x_high, eps_cache = solver.dpm_solver_2_step(x, s, t_, eps_cache=eps_cache)
        else:
            x_low, eps_cache = solver.dpm_solver_2_step(x, s, t_, r1=1 / 3, eps_cache=eps_cache)
            x_high, eps_cache = solver.dpm_solver_3_step(x, s, t_, eps_cache=eps_cache)
        delta = torch.maximum(atol, rtol * torch.maximum(x_low.abs(), x_prev.abs()))
        error = torch.linalg.norm((x_low - x_high) / delta) / x.numel() ** 0.5
        accept = pid.propose_step(error)
        if accept:
            x_prev = x_low
            x = x_high + su * s_noise * noise_sampler(solver.sigma(s), solver.sigma(t))
            s = t
            info['n_accept'] += 1
        else:
            info['n_reject'] += 1
        info['nfe'] += order
        info['steps'] += 1
        if solver.info_callback is not None:
            solver.info_callback({'x': x, 'i': info['steps'] - 1, 't': s, 't_up': s, 'denoised': denoised, 'error': error, 'h': pid.h, **info})
    return x, info


def sample_dpm_adaptive(model, x, sigma_min, sigma_max, extra_args, noise_sampler, eta=0., s_noise=1., callback=None, order=3, rtol=0.05,
                        atol=0.0078, h_init=0.05, return_info=False):
    """k-diffusion sample_dpm_adaptive; modules/sd_samplers_kdiffusion.py:205-208 passes the wrapped model's sigma range."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError('sigma_min and sigma_max must not be 0')
    solver = DPMSolver(model, extra_args)
    if callback is not None:
        solver.info_callback = lambda info: callback({'sigma': solver.sigma(info['t']), 'sigma_hat': solver.sigma(info['t_up']), **info})
    x, info = dpm_solver_adaptive(solver, x, solver.t(torch.tensor(sigma_max)), solver.t(torch.tensor(sigma_min)), noise_sampler, order, rtol,
                                  atol, h_init, 0., 1., 0., 0.81, eta, s_noise)
    return (x, info) if return_info else x


def sample_dpmpp_sde(model, x, sigmas, extra_args, noise_sampler, eta=1., s_noise=1., callback=None, r=1 / 2):
    """k-diffusion sample_dpmpp_sde (DPM-Solver++ (stochastic), restated; unpinned): two model evaluations per step, both sub-steps
    ancestral in the (sigma_down, sigma_up) split, noise from the Brownian tree over [sigma(t), sigma(s)] and [sigma(t), sigma(t_next)]."""
    s_in = x.new_ones([x.shape[0]])
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        if sigmas[i + 1] == 0:
            d = to_d(x, sigmas[i], denoised)
            dt = sigmas[i + 1] - sigmas[i]
            x = x + d * dt
        else:
            t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
            h = t_next - t
            s = t + h * r
            fac = 1 / (2 * r)
            sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(s), eta)
            s_ = t_fn(sd)
            x_2 = (sigma_fn(s_) / sigma_fn(t)) * x - (t - s_).expm1() * denoised
            x_2 = x_2 + noise_sampler(sigma_fn(t), sigma_fn(s)) * s_noise * su
            denoised_2 = model(x_2, sigma_fn(s) * s_in, **extra_args)
            sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(t_next), eta)
            t_next_ = t_fn(sd)
            denoised_d = (1 - fac) * denoised + fac * denoised_2
            x = (sigma_fn(t_next_) / sigma_fn(t)) * x - (t - t_next_).expm1() * denoised_d
            x = x + noise_sampler(sigma_fn(t), sigma_fn(t_next)) * s_noise * su
    return x


def sample_dpmpp_2m_sde(model, x, sigmas, extra_args, noise_sampler, eta=1., s_noise=1., callback=None, solver_type='midpoint'):
    """k-diffusion sample_dpmpp_2m_sde (restated; unpinned): DPM-Solver++(2M) SDE, 'midpoint' or 'heun' second-order correction."""
    if solver_type not in {'heun', 'midpoint'}:
        raise ValueError('solver_type must be \'heun\' or \'midpoint\'')
    s_in = x.new_ones([x.shape[0]])
    old_denoised, h_last = None, None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        if sigmas[i + 1] == 0:
            x = denoised
        else:
            t, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - t
            eta_h = eta * h
            x = sigmas[i + 1] / sigmas[i] * (-eta_h).exp() * x + (-h - eta_h).expm1().neg() * denoised
            if old_denoised is not None:
                r = h_last / h
                if solver_type == 'heun':
                    x = x + ((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / r) * (denoised - old_denoised)
                else:
                    x = x + 0.5 * (-h - eta_h).expm1().neg() * (1 / r) * (denoised - old_denoised)
            if eta:
                x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * s_noise
        old_denoised, h_last = denoised, (-sigmas[i + 1].log() + sigmas[i].log()) if sigmas[i + 1] > 0 else None
    return x


def sample_dpmpp_3m_sde(model, x, sigmas, extra_args, noise_sampler, eta=1., s_noise=1., callback=None):
    """k-diffusion sample_dpmpp_3m_sde (restated; unpinned): third-order multistep DPM-Solver++ SDE."""
    s_in = x.new_ones([x.shape[0]])
    denoised_1, denoised_2 = None, None
    h_1, h_2 = None, None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        if sigmas[i + 1] == 0:
            x = denoised
        else:
            t, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - t
            h_eta = h * (eta + 1)
            x = torch.exp(-h_eta) * x + (-h_eta).expm1().neg() * denoised
            if h_2 is not None:
                r0, r1 = h_1 / h, h_2 / h
                d1_0 = (denoised - denoised_1) / r0
                d1_1 = (denoised_1 - denoised_2) / r1
                d1 = d1_0 + (d1_0 - d1_1) * r0 / (r0 + r1)
                d2 = (d1_0 - d1_1) / (r0 + r1)
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                x = x + phi_2 * d1 - phi_3 * d2
            elif h_1 is not None:
                r = h_1 / h
                d = (denoised - denoised_1) / r
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                x = x + phi_2 * d
            if eta:
                x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * h * eta).expm1().neg().sqrt() * s_noise
            h_1, h_2 = h, h_1
        denoised_1, denoised_2 = denoised, denoised_1
    return x


def sample_lcm(model, x, sigmas, extra_args, noise_fn, callback=None):
    """modules/sd_samplers_lcm.py:66-80: x <- denoised (+ sigma_next * noise while sigma_next > 0)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        x = denoised
        if sigmas[i + 1] > 0:
            x = x + sigmas[i + 1] * noise_fn()
    return x


def sample_euler(model, x, sigmas, extra_args, noise_fn=None, callback=None, s_churn=0.0, s_tmin=0.0, s_tmax=float('inf'), s_noise=1.0):
    """k-diffusion sample_euler (Algorithm 2 of Karras et al.).  s_churn = 0 is the reference default
    (modules/sd_samplers_common.py:242): sigma_hat == sigma and the loop is the one pinned bit-exactly to
    modules/models/sd3/sd3_impls.py:145-163 (tests/golden/euler_twin.npz); with s_churn > 0 the noise level is first raised to
    sigma_hat = sigma * (1 + gamma) with fresh noise (drawn only for the churned steps, as k-diffusion @ ab527a9a does)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        sigma_hat, gamma = _churn(sigmas, i, s_churn, s_tmin, s_tmax)
        if gamma > 0:
            x = x + noise_fn() * s_noise * (sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in, **extra_args)
        d = to_d(x, sigma_hat, denoised)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        dt = sigmas[i + 1] - sigma_hat
        x = x + d * dt
    return x


def sample_dpmpp_2m(model, x, sigmas, extra_args, noise_fn=None, callback=None):
    s_in = x.new_ones([x.shape[0]])
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    old_denoised = None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        if old_denoised is None or sigmas[i + 1] == 0:
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised
        else:
            h_last = t - t_fn(sigmas[i - 1])
            r = h_last / h
            denoised_d = (1 + 1 / (2 * r)) * denoised - (1 / (2 * r)) * old_denoised
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised_d
        old_denoised = denoised
    return x


# --- further k-diffusion samplers of the table at modules/sd_samplers_kdiffusion.py:11-27 (restated from
# crowsonkb/k-diffusion @ ab527a9 sampling.py; [3P], not on disk: anchored on the reference's table and option lists
# :36-46; parity of these restatements is UNPINNED, see the package docstring) -----------------------------------
def _churn(sigmas, i, s_churn, s_tmin, s_tmax):
    gamma = min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0
    return sigmas[i] * (gamma + 1), gamma


def sample_heun(model, x, sigmas, extra_args, noise_fn=None, callback=None, s_churn=0.0, s_tmin=0.0, s_tmax=float('inf'), s_noise=1.0):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        sigma_hat, gamma = _churn(sigmas, i, s_churn, s_tmin, s_tmax)
        if gamma > 0:
            x = x + noise_fn() * s_noise * (sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in, **extra_args)
        d = to_d(x, sigma_hat, denoised)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        dt = sigmas[i + 1] - sigma_hat
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            denoised_2 = model(x_2, sigmas[i + 1] * s_in, **extra_args)
            d_2 = to_d(x_2, sigmas[i + 1], denoised_2)
            d_prime = (d + d_2) / 2
            x = x + d_prime * dt
    return x


def sample_dpm_2(model, x, sigmas, extra_args, noise_fn=None, callback=None, s_churn=0.0, s_tmin=0.0, s_tmax=float('inf'), s_noise=1.0):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        sigma_hat, gamma = _churn(sigmas, i, s_churn, s_tmin, s_tmax)
        if gamma > 0:
            x = x + noise_fn() * s_noise * (sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in, **extra_args)
        d = to_d(x, sigma_hat, denoised)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        if sigmas[i + 1] == 0:
            dt = sigmas[i + 1] - sigma_hat
            x = x + d * dt
        else:
            sigma_mid = sigma_hat.log().lerp(sigmas[i + 1].log(), 0.5).exp()
            dt_1 = sigma_mid - sigma_hat
            dt_2 = sigmas[i + 1] - sigma_hat
            x_2 = x + d * dt_1
            denoised_2 = model(x_2, sigma_mid * s_in, **extra_args)
            d_2 = to_d(x_2, sigma_mid, denoised_2)
            x = x + d_2 * dt_2
    return x


def sample_dpm_2_ancestral(model, x, sigmas, extra_args, noise_fn, eta=1.0, s_noise=1.0, callback=None):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta=eta)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        d = to_d(x, sigmas[i], denoised)
        if sigma_down == 0:
            dt = sigma_down - sigmas[i]
            x = x + d * dt
        else:
            sigma_mid = sigmas[i].log().lerp(sigma_down.log(), 0.5).exp()
            dt_1 = sigma_mid - sigmas[i]
            dt_2 = sigma_down - sigmas[i]
            x_2 = x + d * dt_1
            denoised_2 = model(x_2, sigma_mid * s_in, **extra_args)
            d_2 = to_d(x_2, sigma_mid, denoised_2)
            x = x + d_2 * dt_2
            x = x + noise_fn() * s_noise * sigma_up
    return x


def linear_multistep_coeff(order, t, i, j):
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f'Order {order} too high for step {i}')

    def fn(tau):
        prod = 1.
        for k in range(order):
            if j == k:
                continue
            prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod
    return integrate.quad(fn, t[i], t[i + 1], epsrel=1e-4)[0]


def sample_lms(model, x, sigmas, extra_args, noise_fn=None, callback=None, order=4):
    s_in = x.new_ones([x.shape[0]])
    sigmas_cpu = sigmas.detach().cpu().numpy()
    ds = []
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        d = to_d(x, sigmas[i], denoised)
        ds.append(d)
        if len(ds) > order:
            ds.pop(0)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        cur_order = min(i + 1, order)
        coeffs = [linear_multistep_coeff(cur_order, sigmas_cpu, i, j) for j in range(cur_order)]
        x = x + sum(coeff * d for coeff, d in zip(coeffs, reversed(ds)))
    return x


def sample_dpmpp_2s_ancestral(model, x, sigmas, extra_args, noise_fn, eta=1.0, s_noise=1.0, callback=None):
    s_in = x.new_ones([x.shape[0]])
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta=eta)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        if sigma_down == 0:
            d = to_d(x, sigmas[i], denoised)
            dt = sigma_down - sigmas[i]
            x = x + d * dt
        else:
            t, t_next = t_fn(sigmas[i]), t_fn(sigma_down)
            r = 1 / 2
            h = t_next - t
            s = t + r * h
            x_2 = (sigma_fn(s) / sigma_fn(t)) * x - (-h * r).expm1() * denoised
            denoised_2 = model(x_2, sigma_fn(s) * s_in, **extra_args)
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised_2
        if sigmas[i + 1] > 0:
            x = x + noise_fn() * s_noise * sigma_up
    return x


def sample_plms(model, x, timesteps, alphas_cumprod, extra_args, callback=None):
    """modules/sd_samplers_timesteps_impl.py:85-137 (in-repo; pinned by tests/golden/plms.npz)."""
    alphas = alphas_cumprod[timesteps]
    alphas_prev = alphas_cumprod[torch.nn.functional.pad(timesteps[:-1], pad=(1, 0))].to(torch.float64)
    sqrt_one_minus_alphas = torch.sqrt(1 - alphas)
    s_in = x.new_ones([x.shape[0]])
    s_x = x.new_ones((x.shape[0], 1, 1, 1))
    old_eps = []

    def get_x_prev_and_pred_x0(e_t, index):
        a_t = alphas[index].item() * s_x
        a_prev = alphas_prev[index].item() * s_x
        sqrt_one_minus_at = sqrt_one_minus_alphas[index].item() * s_x
        pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
        dir_xt = (1. - a_prev).sqrt() * e_t
        x_prev = a_prev.sqrt() * pred_x0 + dir_xt
        return x_prev, pred_x0

    for i in range(len(timesteps) - 1):
        index = len(timesteps) - 1 - i
        ts = timesteps[index].item() * s_in
        t_next = timesteps[max(index - 1, 0)].item() * s_in
        e_t = model(x, ts, **extra_args)
        if len(old_eps) == 0:
            x_prev, pred_x0 = get_x_prev_and_pred_x0(e_t, index)
            e_t_next = model(x_prev, t_next, **extra_args)
            e_t_prime = (e_t + e_t_next) / 2
        elif len(old_eps) == 1:
            e_t_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_t_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_t_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        x_prev, pred_x0 = get_x_prev_and_pred_x0(e_t_prime, index)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        x = x_prev
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': 0, 'sigma_hat': 0, 'denoised': pred_x0})
    return x


def restart_sampler(model, x, sigmas, extra_args, noise_fn, callback=None, s_noise=1.0, restart_list=None):
    """modules/sd_samplers_extra.py:6-74 (in-repo; pinned by tests/golden/restart.npz): Heun steps over a Karras schedule with
    "restart" segments that re-noise from sigma ~0.1 back up to sigma ~2."""
    s_in = x.new_ones([x.shape[0]])
    step_id = 0

    def heun_step(x, old_sigma, new_sigma, second_order=True):
        nonlocal step_id
        denoised = model(x, old_sigma * s_in, **extra_args)
        d = to_d(x, old_sigma, denoised)
        if callback is not None:
            callback({'x': x, 'i': step_id, 'sigma': new_sigma, 'sigma_hat': old_sigma, 'denoised': denoised})
        dt = new_sigma - old_sigma
        if new_sigma == 0 or not second_order:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            denoised_2 = model(x_2, new_sigma * s_in, **extra_args)
            d_2 = to_d(x_2, new_sigma, denoised_2)
            d_prime = (d + d_2) / 2
            x = x + d_prime * dt
        step_id += 1
        return x

    step_list = restart_step_list(sigmas, restart_list)
    last_sigma = None
    for old_sigma, new_sigma in step_list:
        if last_sigma is None:
            last_sigma = old_sigma
        elif last_sigma < old_sigma:
            x = x + noise_fn() * s_noise * (old_sigma ** 2 - last_sigma ** 2) ** 0.5
        x = heun_step(x, old_sigma, new_sigma)
        last_sigma = new_sigma
    return x


def restart_step_list(sigmas, restart_list=None):
    """The (old_sigma, new_sigma) sequence of modules/sd_samplers_extra.py:38-62."""
    steps = sigmas.shape[0] - 1
    if restart_list is None:
        if steps >= 20:
            restart_steps = 9
            restart_times = 1
            if steps >= 36:
                restart_steps = steps // 4
                restart_times = 2
            sigmas = get_sigmas_karras(steps - restart_steps * restart_times, sigmas[-2].item(), sigmas[0].item())
            restart_list = {0.1: [restart_steps + 1, restart_times, 2]}
        else:
            restart_list = {}
    restart_list = {int(torch.argmin(abs(sigmas - key), dim=0)): value for key, value in restart_list.items()}
    step_list = []
    for i in range(len(sigmas) - 1):
        step_list.append((sigmas[i], sigmas[i + 1]))
        if i + 1 in restart_list:
            restart_steps, restart_times, restart_max = restart_list[i + 1]
            min_idx = i + 1
            max_idx = int(torch.argmin(abs(sigmas - restart_max), dim=0))
            if max_idx < min_idx:
                sigma_restart = get_sigmas_karras(restart_steps, sigmas[min_idx].item(), sigmas[max_idx].item())[:-1]
                while restart_times > 0:
                    restart_times -= 1
                    step_list.extend(zip(sigma_restart[:-1], sigma_restart[1:]))
    return step_list


def ddim_timesteps(steps: int) -> torch.Tensor:
    """modules/sd_samplers_timesteps.py:94"""
    return torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)


def sample_ddim(model, x, timesteps, alphas_cumprod, extra_args, noise_fn, eta=0.0, callback=None):
    """modules/sd_samplers_timesteps_impl.py:12-40.  ``model(x, t*s_in, **extra_args)`` returns eps (the CFG
    combine of eps predictions, CFGDenoiserTimesteps); alphas_prev is float64 as in the reference (:15)."""
    alphas = alphas_cumprod[timesteps]
    alphas_prev = alphas_cumprod[torch.nn.functional.pad(timesteps[:-1], pad=(1, 0))].to(torch.float64)
    sqrt_one_minus_alphas = torch.sqrt(1 - alphas)
    sigmas = eta * np.sqrt((1 - alphas_prev.cpu().numpy()) / (1 - alphas.cpu()) * (1 - alphas.cpu() / alphas_prev.cpu().numpy()))
    s_in = x.new_ones((x.shape[0]))
    s_x = x.new_ones((x.shape[0], 1, 1, 1))
    for i in range(len(timesteps) - 1):
        index = len(timesteps) - 1 - i
        e_t = model(x, timesteps[index].item() * s_in, **extra_args)
        a_t = alphas[index].item() * s_x
        a_prev = alphas_prev[index].item() * s_x
        sigma_t = sigmas[index].item() * s_x
        sqrt_one_minus_at = sqrt_one_minus_alphas[index].item() * s_x
        pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
        dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
        noise = sigma_t * noise_fn()
        x = a_prev.sqrt() * pred_x0 + dir_xt + noise
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': 0, 'sigma_hat': 0, 'denoised': pred_x0})
    return x


def sample_ddim_cfgpp(model, x, timesteps, alphas_cumprod, extra_args, noise_fn, eta=0.0, callback=None):
    """modules/sd_samplers_timesteps_impl.py:43-82 (pinned by tests/golden/ddim.npz, keys cfgpp_*).  ``model`` is the CFG
    denoiser: the function sets cond_scale_miltiplier = 1/12.5 and reads model.last_noise_uncond after every call."""
    alphas = alphas_cumprod[timesteps]
    alphas_prev = alphas_cumprod[torch.nn.functional.pad(timesteps[:-1], pad=(1, 0))].to(torch.float64)
    sqrt_one_minus_alphas = torch.sqrt(1 - alphas)
    sigmas = eta * np.sqrt((1 - alphas_prev.cpu().numpy()) / (1 - alphas.cpu()) * (1 - alphas.cpu() / alphas_prev.cpu().numpy()))
    model.cond_scale_miltiplier = 1 / 12.5
    model.need_last_noise_uncond = True
    s_in = x.new_ones((x.shape[0]))
    s_x = x.new_ones((x.shape[0], 1, 1, 1))
    for i in range(len(timesteps) - 1):
        index = len(timesteps) - 1 - i
        e_t = model(x, timesteps[index].item() * s_in, **extra_args)
        last_noise_uncond = model.last_noise_uncond
        a_t = alphas[index].item() * s_x
        a_prev = alphas_prev[index].item() * s_x
        sigma_t = sigmas[index].item() * s_x
        sqrt_one_minus_at = sqrt_one_minus_alphas[index].item() * s_x
        pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
        dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * last_noise_uncond
        noise = sigma_t * noise_fn()
        x = a_prev.sqrt() * pred_x0 + dir_xt + noise
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': 0, 'sigma_hat': 0, 'denoised': pred_x0})
    return x


def setup_img2img_steps(requested_steps: int, denoising_strength: float, steps_given: bool = True):
    """modules/sd_samplers_common.py:22-31.  ``steps_given`` = the caller passed ``steps`` (hires second pass,
    processing.py:1454) -> first branch; plain img2img (processing.py:1774 passes no steps, img2img_fix_steps off) ->
    second branch."""
    if steps_given:
        steps = int(requested_steps / min(denoising_strength, 0.999)) if denoising_strength > 0 else 0
        t_enc = requested_steps - 1
    else:
        steps = requested_steps
        t_enc = int(min(denoising_strength, 0.999) * steps)
    return steps, t_enc
