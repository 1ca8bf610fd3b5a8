"""Checkpoint -> engine.  Keeps the reference's loader contract: a flat state dict keyed ``model.diffusion_model.*``,
``first_stage_model.*`` (modules/sd_models.py:312-329 read_state_dict, :262-281 key fix-ups, :392/:452-454 prefixes) is
the weight source; the engine consumes it tensor by tensor (sdmi_unet_load_tensor / sdmi_vae_load_tensor).

``SdModel`` is the slice of the reference's ``shared.sd_model`` object the hot path touches: ``alphas_cumprod``,
``parameterization``, ``is_sdxl``, ``decode_first_stage`` / ``encode_first_stage`` / ``get_first_stage_encoding``.
"""
from __future__ import annotations

import dataclasses
import os
import types
from typing import Optional

import torch

from . import schema, shared
from .engine import Engine


# modules/sd_models.py:243-251 — old SD1 checkpoints store the CLIP text model one level up; SD 2.1 Turbo ships in SGM layout
checkpoint_dict_replacements_sd1 = {
    'cond_stage_model.transformer.embeddings.': 'cond_stage_model.transformer.text_model.embeddings.',
    'cond_stage_model.transformer.encoder.': 'cond_stage_model.transformer.text_model.encoder.',
    'cond_stage_model.transformer.final_layer_norm.': 'cond_stage_model.transformer.text_model.final_layer_norm.',
}
checkpoint_dict_replacements_sd2_turbo = {
    'conditioner.embedders.0.': 'cond_stage_model.',
}


def transform_checkpoint_dict_key(k, replacements):
    """modules/sd_models.py:254-259"""
    for text, replacement in replacements.items():
        if k.startswith(text):
            k = replacement + k[len(text):]
    return k


def get_state_dict_from_checkpoint(pl_sd: dict) -> dict:
    """modules/sd_models.py:262-281: unwrap a pytorch-lightning "state_dict" and apply the key fix-ups, in place."""
    pl_sd = pl_sd.pop("state_dict", pl_sd)
    pl_sd.pop("state_dict", None)
    ln = pl_sd.get('conditioner.embedders.0.model.ln_final.weight')
    is_sd2_turbo = ln is not None and ln.size()[0] == 1024
    table = checkpoint_dict_replacements_sd2_turbo if is_sd2_turbo else checkpoint_dict_replacements_sd1
    sd = {}
    for k, v in pl_sd.items():
        new_key = transform_checkpoint_dict_key(k, table)
        if new_key is not None:
            sd[new_key] = v
    pl_sd.clear()
    pl_sd.update(sd)
    return pl_sd


def read_state_dict(checkpoint_file: str, print_global_state=False, map_location=None) -> dict:
    """modules/sd_models.py:312-329: .safetensors through safetensors (mmap load_file, or a whole-file read when
    opts.disable_mmap_load_safetensors), anything else through torch.load; then the key fix-ups of :262-281."""
    _, ext = os.path.splitext(checkpoint_file)
    device = map_location or getattr(shared, "weight_load_location", None) or "cpu"
    if ext.lower() == ".safetensors":
        import safetensors.torch
        if not getattr(shared.opts, "disable_mmap_load_safetensors", False):
            pl_sd = safetensors.torch.load_file(checkpoint_file, device=str(device))
        else:
            pl_sd = safetensors.torch.load(open(checkpoint_file, 'rb').read())
            pl_sd = {k: v.to(device) for k, v in pl_sd.items()}
    else:
        pl_sd = torch.load(checkpoint_file, map_location=device, weights_only=True)
    if print_global_state and "global_step" in pl_sd:
        print(f"Global Step: {pl_sd['global_step']}")
    return get_state_dict_from_checkpoint(pl_sd)


# modules/sd_vae.py:13 — keys of a standalone VAE file that are not weights
vae_ignore_keys = {"model_ema.decay", "model_ema.num_updates"}


def load_vae_dict(filename: str, map_location=None) -> dict:
    """modules/sd_vae.py:188-191"""
    vae_ckpt = read_state_dict(filename, map_location=map_location)
    return {k: v for k, v in vae_ckpt.items() if k[0:4] != "loss" and k not in vae_ignore_keys}


def guess_unet_config(sd: dict) -> schema.UNetConfig:
    """modules/sd_models_config.py:72-115 reduced to the two families on the path: SDXL is recognised by the
    conditioner / label_emb keys, everything else with a 768-wide attn2.to_k is SD1.x."""
    cfg = schema.sdxl_unet() if schema.UNET_PREFIX + "label_emb.0.0.weight" in sd else schema.sd15_unet()
    conv_in = sd.get(schema.UNET_PREFIX + "input_blocks.0.0.weight")
    if conv_in is not None and conv_in.shape[1] != cfg.in_channels:      # 9: inpainting, 8: InstructPix2Pix (sd_models_config.py:100-107)
        cfg = dataclasses.replace(cfg, in_channels=int(conv_in.shape[1]))
    return cfg


class SdModel:
    def __init__(self, state_dict: dict, unet_cfg: Optional[schema.UNetConfig] = None,
                 vae_cfg: Optional[schema.VAEConfig] = None, device: int = 0, load_vae: bool = True,
                 vae_decoder_only: bool = False, parameterization: str = "eps", cond_stage_key: Optional[str] = None,
                 conditioning_key: Optional[str] = None, embedder=None, noise_augmentor=None, depth_model=None):
        self.unet_cfg = unet_cfg or guess_unet_config(state_dict)
        # unCLIP checkpoints (SD 2.1-unclip: conditioning_key "crossattn-adm", configs/v2-1-stable-unclip-*.yaml) carry the same
        # label_emb vector input as SDXL but are not SDXL: the vector is the CLIP image embedding + its noise-level embedding
        self.is_sdxl = self.unet_cfg.adm_in_channels is not None and conditioning_key != "crossattn-adm"
        # what the reference reads from the checkpoint's yaml (modules/sd_models_config.py:72-115): 9 input channels = inpainting
        # ("hybrid": c_concat = mask + masked-image latent), 8 = InstructPix2Pix (also "hybrid", cond_stage_key "edit")
        cin = self.unet_cfg.in_channels
        self.cond_stage_key = cond_stage_key or ("edit" if cin == 8 else "txt")
        self.is_sdxl_inpaint = self.is_sdxl and cin == 9
        # 5 input channels = depth2img (LatentDepth2ImageDiffusion: "hybrid", c_concat = the MiDaS depth map at latent size)
        self.is_depth2img = cin == 5 and not self.is_sdxl
        self.model = types.SimpleNamespace(conditioning_key=conditioning_key or ("hybrid" if cin in (5, 8, 9) and not self.is_sdxl else "crossattn"))
        # torch modules of the host application that build the image conditioning of those two families (modules/processing.py:304-340);
        # they run once per job, outside the sampling loop, exactly as in the reference
        self.embedder, self.noise_augmentor, self.depth_model = embedder, noise_augmentor, depth_model
        shared.sd_model = self                            # the reference's global (modules/shared.py); schedulers read is_sdxl
        self.vae_cfg = vae_cfg or (schema.sdxl_vae() if self.is_sdxl else schema.sd15_vae())
        assert parameterization in ("eps", "v")
        self.parameterization = parameterization          # "v": SD 2.x 768-v (modules/sd_models_config.py:86-94)
        self.device = torch.device("cuda", device)
        ac = state_dict.get("alphas_cumprod")
        self.alphas_cumprod = (ac.float().cpu() if ac is not None else schema.make_alphas_cumprod())
        self.alphas_cumprod_original = self.alphas_cumprod.clone()       # modules/sd_models.py:441 (what the schedule overrides start from)
        self.engine = Engine(device)
        self.engine.load_unet(self.unet_cfg, state_dict)
        self._checkpoint = state_dict                     # kept by reference: the "weights backup" LoRA rewrites start from
        self.has_vae = False
        self.vae_range_extended = False
        self.loaded_vae_file = None
        self._vae_decoder_only = vae_decoder_only
        if load_vae and any(k.startswith(schema.VAE_PREFIX) for k in state_dict):
            self.engine.load_vae(self.vae_cfg, state_dict, decoder_only=vae_decoder_only)
            self.has_vae = True
        self.scale_factor = self.vae_cfg.scale_factor

    # --- external VAE (modules/sd_vae.py:194-280): "SD VAE" setting / per-checkpoint .vae.safetensors ----------------------------
    def load_vae(self, vae_file: Optional[str] = None, vae_source: str = "from unknown source", vae_dict: Optional[dict] = None):
        """modules/sd_vae.py:194-235 load_vae: replace the first stage's weights by a standalone VAE file (keys WITHOUT the
        ``first_stage_model.`` prefix, "loss*" and EMA bookkeeping keys dropped); ``vae_file=None`` restores the checkpoint's own
        VAE (restore_base_vae, :246-252).  The engine re-packs the decoder (and encoder) in place."""
        if vae_file is None and vae_dict is None:
            if self.loaded_vae_file is not None:
                self.engine.load_vae(self.vae_cfg, self._checkpoint, decoder_only=self._vae_decoder_only)
            self.loaded_vae_file = None
            return
        if vae_dict is None:
            assert os.path.isfile(vae_file), f"VAE {vae_source} doesn't exist: {vae_file}"
            vae_dict = load_vae_dict(vae_file)
        self.engine.load_vae(self.vae_cfg, vae_dict, prefix="", decoder_only=self._vae_decoder_only)
        self.has_vae = True
        self.loaded_vae_file = vae_file or "<dict>"

    def set_vae_range_extended(self, on: bool):
        """The engine's stand-in for the reference's fp16 -> fp32 / bf16 VAE fallback (modules/processing.py:636-665)."""
        self.engine.set_option("vae_range_extend", 1 if on else 0)
        self.vae_range_extended = bool(on)

    def set_accuracy_mode(self, on: bool):
        """The engine's counterpart of the reference's --no-half / upcast switches (modules/devices.py:284-295 keeps GroupNorm in fp32
        even on the GPU path; --no-half runs the whole model in fp32): every UNet tensor that is not a matrix-core operand — the carried
        residual stream, the skip_connection and first-conv outputs, the latent on its way into conv_in — as (hi, lo) fp16 pairs
        (~22 bits).  One CFG forward at the C1 shape: 1.5e-3 -> below 1e-3 from the fp32 oracle (tests/test_gpu_c1_parity.py::
        test_c1_unet_forward_accuracy_mode_vs_oracle; measured values and cost in profiles/r06_parity.json).
        Engines that cannot take the pairs (force_generic / use_glds = 0 cross-check settings) run the plain fp16 stream."""
        on = bool(on)
        if on != getattr(self, "accuracy_mode", False):
            self.engine.set_option("residual_fp32", 1 if on else 0)
            self.accuracy_mode = on

    def unet_checkpoint_tensor(self, engine_key: str) -> torch.Tensor:
        """The unmodified checkpoint weight of a UNet layer (extensions-builtin/Lora/networks.py:423-432 keeps the same thing
        as ``network_weights_backup``)."""
        return self._checkpoint[schema.UNET_PREFIX + engine_key]

    # --- the methods the reference calls on shared.sd_model around the sampler -----------------------------------
    def decode_first_stage(self, z: torch.Tensor) -> torch.Tensor:
        """LatentDiffusion.decode_first_stage: image in [-1, 1], fp32 NCHW (whole batch in one pass)."""
        return self.engine.vae_decode(z)

    def encode_first_stage(self, x: torch.Tensor) -> torch.Tensor:
        """Returns the posterior moments (mean | logvar), as AutoencoderKL.encode's DiagonalGaussianDistribution holds."""
        return self.engine.vae_encode_moments(x)

    def get_first_stage_encoding(self, moments: torch.Tensor, sample: bool = False, generator=None) -> torch.Tensor:
        """scale_factor * (mean [+ std * eps]); the reference samples (sd3_impls.py:369-374 twin), the measurement plan
        uses the mean for determinism."""
        mean, logvar = torch.chunk(moments, 2, dim=1)
        if not sample:
            from . import ops
            mean = mean.contiguous()
            return ops.lincomb(torch.empty_like(mean), [mean], [float(self.scale_factor)])
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        eps = torch.randn(mean.shape, generator=generator, device=mean.device, dtype=mean.dtype)
        return ((mean + std * eps) * self.scale_factor).contiguous()


def rescale_zero_terminal_snr_abar(alphas_cumprod: torch.Tensor) -> torch.Tensor:
    """modules/sd_models.py:628-644 ("Zero Terminal SNR", arXiv 2305.08891): sqrt(alpha_bar) is shifted so that the last timestep
    carries no signal and rescaled so that the first keeps its value; the last entry is then pinned to 4.8973451890853435e-08
    (not exactly zero: sigma stays finite)."""
    root = alphas_cumprod.sqrt()
    first, last = root[0].clone(), root[-1].clone()
    root = (root - last) * (first / (first - last))
    out = root ** 2
    out[-1] = 4.8973451890853435e-08
    return out


def apply_alpha_schedule_override(sd_model, p=None):
    """modules/sd_models.py:647-668, called once per job (modules/processing.py:930): start from the checkpoint's schedule,
    optionally round it through fp16 (opts.use_downcasted_alpha_bar), optionally rescale it to zero terminal SNR
    (opts.sd_noise_schedule).  The samplers read sd_model.alphas_cumprod when their denoiser wrapper is built."""
    if not hasattr(sd_model, 'alphas_cumprod') or not hasattr(sd_model, 'alphas_cumprod_original'):
        return
    opts = shared.opts
    sd_model.alphas_cumprod = sd_model.alphas_cumprod_original.clone()
    if opts.use_downcasted_alpha_bar:
        if p is not None:
            p.extra_generation_params['Downcast alphas_cumprod'] = opts.use_downcasted_alpha_bar
        sd_model.alphas_cumprod = sd_model.alphas_cumprod.half()
    if opts.sd_noise_schedule == "Zero Terminal SNR":
        if p is not None:
            p.extra_generation_params['Noise Schedule'] = opts.sd_noise_schedule
        sd_model.alphas_cumprod = rescale_zero_terminal_snr_abar(sd_model.alphas_cumprod)


def load_model(checkpoint_file: Optional[str] = None, state_dict: Optional[dict] = None, device: int = 0, **kw) -> SdModel:
    """modules/sd_models.py:786 load_model, reduced to what feeds the engine."""
    if state_dict is None:
        state_dict = read_state_dict(checkpoint_file)
    return SdModel(state_dict, device=device, **kw)
