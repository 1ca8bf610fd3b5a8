"""The handful of reference settings the hot path reads (mirror of the relevant ``shared.opts`` keys).

Standalone (bench, tests) ``opts`` / ``state`` / ``cmd_opts`` are the objects below: the same names with the reference's defaults
(modules/shared_options.py, lines cited per key).  Inside the webui the extension script calls ``bind_webui(modules.shared, ...)``
(webui_bridge.bind_shared) and the three names become views of the webui's OWN objects — every ``shared.opts.x`` /
``shared.state.x`` in this package is an attribute lookup at the time of use, so the user's sampler settings, Interrupt / Skip,
the progress counters and the live preview reach the engine samplers exactly as they reach the stock ones
(modules/sd_samplers_kdiffusion.py:8-9, modules/sd_samplers_common.py:256-263, modules/sd_samplers_cfg_denoiser.py:157-158, 304).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Options:
    randn_source: str = "NV"                       # :183  (the engine implements the Philox "NV" source natively)
    eta_noise_seed_delta: int = 0                  # :399
    eta_ancestral: float = 1.0                     # :390
    eta_ddim: float = 0.0                          # :389
    uni_pc_variant: str = "bh1"                    # :402  (bh1 | bh2 | vary_coeff)
    uni_pc_skip_type: str = "time_uniform"         # :403
    uni_pc_order: int = 3                          # :404
    uni_pc_lower_order_final: bool = True          # :405
    s_churn: float = 0.0                           # :392
    s_tmin: float = 0.0                            # :393
    s_tmax: float = 0.0                            # :394  (0 = inf)
    s_noise: float = 1.0                           # :395
    sigma_min: float = 0.0                         # :396  (0 = model default)
    sigma_max: float = 0.0                         # :397
    rho: float = 0.0                               # :398
    always_discard_next_to_last_sigma: bool = False  # :400
    sgm_noise_multiplier: bool = False             # :401
    use_old_karras_scheduler_sigmas: bool = False
    batch_cond_uncond: bool = True                 # :242
    s_min_uncond: float = 0.0                      # :234  (NGMS: skip the negative prompt on alternate steps below this sigma)
    s_min_uncond_all: bool = False                 # :235
    pad_cond_uncond: bool = False                  # :239
    pad_cond_uncond_v0: bool = False               # :240
    skip_early_cond: float = 0.0                   # :407
    img2img_extra_noise: float = 0.0
    inpainting_mask_weight: float = 1.0            # :216
    overlay_inpaint: bool = True                   # :229 (inpainting: composite the unmasked original over the result)
    upscaler_for_img2img: str = None               # :106
    hires_fix_refiner_pass: str = "second pass"    # :185
    refiner_switch_by_sample_steps: bool = False   # :256
    initial_noise_multiplier: float = 1.0          # :217
    img2img_fix_steps: bool = False
    enable_quantization: bool = False              # :176
    use_downcasted_alpha_bar: bool = False         # :255
    sd_noise_schedule: str = "Default"             # :406 ("Default" | "Zero Terminal SNR")
    live_previews_enable: bool = False             # :374 (the webui's default is True; standalone there is nobody to show it to)
    show_progress_every_n_steps: int = 10          # :377
    live_preview_content: str = "Prompt"           # :380 ("Combined" | "Prompt" | "Negative prompt")
    token_merging_ratio: float = 0.0               # :229 (ToMe patches torch modules the engine UNet never calls: refused, sd_unet.py)
    token_merging_ratio_hr: float = 0.0            # :231
    hypertile_enable_unet: bool = False            # extensions-builtin/hypertile/scripts/hypertile_script.py:83
    hypertile_enable_unet_secondpass: bool = False  # :84
    CLIP_stop_at_last_layers: int = 1              # :170 ("Clip skip")
    sdxl_clip_l_skip: bool = False                 # :222
    beta_dist_alpha: float = 0.6                   # :408
    beta_dist_beta: float = 0.6                    # :409
    no_dpmpp_sde_batch_determinism: bool = False   # :252 (compatibility: DPM++ SDE noise from the batch generator instead of per-seed trees)
    tiling: bool = False                           # :228
    auto_vae_precision_bfloat16: bool = False      # :181  ("Automatically convert VAE to bfloat16")
    auto_vae_precision: bool = True                # :182  ("Automatically revert VAE to 32-bit floats")
    disable_mmap_load_safetensors: bool = False    # :285
    mi355x_devices: str = ""                       # (engine option) "0,1,2,3": devices of THIS process a job's images are spread over (parallel.DevicePool); empty = one
    mi355x_devices_serial: bool = False            # run the pool's workers one after the other (the host-emulated CPU tier)
    sdmi_accuracy_mode: bool = False               # (engine option, not a webui setting) carry the UNet's residual stream with ~22 bits: sd_models.set_accuracy_mode
    mi355x_auto_cfg_pairs: bool = True             # (engine option; default ON since round 6: 19.56 -> 20.05 images/s on the drop-in path, bench.py dropin legs) Mi355xUnet.forward behind the webui's stock CFG denoiser: the engine finds the [x | x] batch itself (sd_unet.py)


opts = Options()


@dataclass
class CmdOpts:
    """modules/cmd_args.py flags the path reads."""
    disable_nan_check: bool = False                # cmd_args.py:23
    no_half: bool = False
    no_half_vae: bool = False


cmd_opts = CmdOpts()
weight_load_location = None                        # modules/shared.py:20 (None = cpu for .ckpt, the device name for safetensors)


class State:
    """modules/shared_state.py: only the fields the sampler loop touches."""
    def __init__(self):
        self.interrupted = False
        self.skipped = False
        self.sampling_step = 0
        self.sampling_steps = 0
        self.current_latent = None


class TotalTqdm:
    """modules/shared_total_tqdm.py: the console progress bar Sampler.callback_state advances (sd_samplers_common.py:263)."""
    def update(self):
        pass


_own_opts, _own_cmd_opts, _own_state, _own_total_tqdm = opts, cmd_opts, State(), TotalTqdm()
state = _own_state
total_tqdm = _own_total_tqdm


def store_latent(decoded):
    """modules/sd_samplers_common.py:115-120 without the preview decode (standalone there is no viewer): the latest x0 prediction
    stays readable as ``state.current_latent``.  ``bind_webui`` replaces this with the webui's own function, which also renders the
    preview every ``show_progress_every_n_steps`` steps."""
    state.current_latent = decoded


class _WebuiView:
    """``modules.shared.<name>`` first, this package's default object for the names the webui's does not carry (an older webui without
    an option, ``cmd_opts`` flags of a fork).  The webui object is fetched from its module at every access: ``modules.shared.opts`` /
    ``state`` are module attributes the webui itself rebinds in places (shared_init, tests)."""

    def __init__(self, module, name, fallback):
        object.__setattr__(self, "_module", module)
        object.__setattr__(self, "_name", name)
        object.__setattr__(self, "_fallback", fallback)

    def __getattr__(self, item):
        target = getattr(object.__getattribute__(self, "_module"), object.__getattribute__(self, "_name"), None)
        if target is not None:
            try:
                return getattr(target, item)
            except AttributeError:
                pass
        return getattr(object.__getattribute__(self, "_fallback"), item)

    def __setattr__(self, item, value):
        target = getattr(object.__getattribute__(self, "_module"), object.__getattribute__(self, "_name"), None)
        setattr(object.__getattribute__(self, "_fallback") if target is None else target, item, value)


def bind_webui(webui_shared, webui_store_latent=None, webui_mask_blend_args=None):
    """Make ``opts`` / ``state`` / ``cmd_opts`` / ``total_tqdm`` views of ``modules.shared``'s and ``store_latent`` the webui's own
    (modules/sd_samplers_common.py:115).  Idempotent; ``unbind_webui`` restores the standalone objects."""
    global opts, state, cmd_opts, total_tqdm, store_latent, webui, MaskBlendArgs
    opts = _WebuiView(webui_shared, "opts", _own_opts)
    state = _WebuiView(webui_shared, "state", _own_state)
    cmd_opts = _WebuiView(webui_shared, "cmd_opts", _own_cmd_opts)
    total_tqdm = _WebuiView(webui_shared, "total_tqdm", _own_total_tqdm)
    if webui_store_latent is not None:
        store_latent = webui_store_latent
    if webui_mask_blend_args is not None:
        MaskBlendArgs = webui_mask_blend_args
    webui = webui_shared


def unbind_webui():
    global opts, state, cmd_opts, total_tqdm, store_latent, webui, MaskBlendArgs
    opts, state, cmd_opts, total_tqdm, webui = _own_opts, _own_state, _own_cmd_opts, _own_total_tqdm, None
    store_latent, MaskBlendArgs = _standalone_store_latent, _standalone_mask_blend_args


class MaskBlendArgs:
    """modules/scripts.py:16-26 (what Script.on_mask_blend receives); ``bind_webui`` swaps in the webui's own class."""
    def __init__(self, current_latent, nmask, init_latent, mask, blended_latent, denoiser=None, sigma=None):
        self.current_latent = current_latent
        self.nmask = nmask
        self.init_latent = init_latent
        self.mask = mask
        self.blended_latent = blended_latent
        self.denoiser = denoiser
        self.is_final_blend = denoiser is None
        self.sigma = sigma


_standalone_store_latent, _standalone_mask_blend_args = store_latent, MaskBlendArgs
webui = None                                       # modules.shared once bound
sd_upscalers = []                                  # UpscalerData entries (upscaler.py; modules/shared.py:64)
sd_model = None                                    # set by sd_models.SdModel (schedulers read is_sdxl, sd_schedulers.py:57)
