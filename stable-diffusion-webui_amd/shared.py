"""The handful of reference settings the hot path reads (mirror of the relevant ``shared.opts`` keys).

Inside the webui the real ``modules.shared.opts`` is used instead; standalone (bench, tests) this
object provides the same names with the reference's defaults (modules/shared_options.py, lines cited per key).
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
    upscaler_for_img2img: str = None               # :106
    hires_fix_refiner_pass: str = "second pass"    # :185
    refiner_switch_by_sample_steps: bool = False   # :256
    initial_noise_multiplier: float = 1.0          # :217
    img2img_fix_steps: bool = False
    enable_quantization: bool = False              # :176
    use_downcasted_alpha_bar: bool = False         # :255
    sd_noise_schedule: str = "Default"             # :406 ("Default" | "Zero Terminal SNR")
    live_previews_enable: bool = False             # :374 (fused path requires previews off)
    CLIP_stop_at_last_layers: int = 1              # :170 ("Clip skip")
    sdxl_clip_l_skip: bool = False                 # :222
    beta_dist_alpha: float = 0.6                   # :408
    beta_dist_beta: float = 0.6                    # :409
    no_dpmpp_sde_batch_determinism: bool = False   # :252 (compatibility: DPM++ SDE noise from the batch generator instead of per-seed trees)
    tiling: bool = False                           # :228
    auto_vae_precision_bfloat16: bool = False      # :181  ("Automatically convert VAE to bfloat16")
    auto_vae_precision: bool = True                # :182  ("Automatically revert VAE to 32-bit floats")
    disable_mmap_load_safetensors: bool = False    # :285


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


state = State()
sd_upscalers = []                                  # UpscalerData entries (upscaler.py; modules/shared.py:64)
sd_model = None                                    # set by sd_models.SdModel (schedulers read is_sdxl, sd_schedulers.py:57)
