"""Entry point of the path: ``StableDiffusionProcessingTxt2Img`` / ``...Img2Img`` + ``process_images`` with the reference's
field names and control flow (modules/processing.py:136-227, 819-1150, 1166-1464, 1557-1789), driving the engine.

Inside the webui the reference's own ``modules/processing.py`` stays in charge and reaches the engine through the
plugin boundaries.  This mirror exists so the same flow runs standalone (bench.py, tests, multi-GPU
runner) where the webui's Python dependencies are absent: prompts are replaced by ready conditioning tensors
(``p.c`` / ``p.uc``; the CLIP text encoder is row N2 of SURVEY.md section 8f), images come back as uint8 HWC arrays.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np
import torch

from . import ops, prompt_parser, sd_models, sd_samplers, shared
from .rng import ImageRNG

opt_C = 4
opt_f = 8


@dataclass
class StableDiffusionProcessing:
    sd_model: Any = None
    prompt: Any = ""
    negative_prompt: Any = ""
    c: Optional[torch.Tensor] = None                 # conditioning  [N, 77k, ctx]  (N = batch_size * n_iter)
    uc: Optional[torch.Tensor] = None                # unconditional conditioning
    y: Optional[torch.Tensor] = None                 # SDXL vector conditioning [N, 2816] (cond) / uy (uncond)
    uy: Optional[torch.Tensor] = None
    seed: int = -1
    subseed: int = -1
    subseed_strength: float = 0
    seed_resize_from_h: int = -1
    seed_resize_from_w: int = -1
    sampler_name: str = None
    scheduler: str = None
    batch_size: int = 1
    n_iter: int = 1
    steps: int = 50
    cfg_scale: float = 7.0
    width: int = 512
    height: int = 512
    denoising_strength: float = None
    eta: float = None
    s_min_uncond: float = None
    s_churn: float = None
    s_tmax: float = None
    s_tmin: float = None
    s_noise: float = None
    sampler_noise_scheduler_override: Any = None
    is_hr_pass: bool = False
    tiling: bool = None                               # :160 — None = opts.tiling; Conv2d padding_mode 'circular' for seamless textures
    inpainting_mask_weight: float = None              # opts.inpainting_mask_weight ("Conditional mask weight")
    refiner_sd_model: Any = None                      # p.refiner_checkpoint (:177, 882-885) as a second, resident SdModel
    refiner_switch_at: float = None                   # :178
    refiner_c: Optional[torch.Tensor] = None          # conds encoded by the refiner's own text encoder (p.setup_conds after the switch)
    refiner_uc: Optional[torch.Tensor] = None
    refiner_y: Optional[torch.Tensor] = None
    refiner_uy: Optional[torch.Tensor] = None
    is_using_inpainting_conditioning: bool = False
    # runtime
    sampler: Any = None
    rng: Any = None
    seeds: List[int] = None
    all_seeds: List[int] = None
    subseeds: List[int] = None
    all_subseeds: List[int] = None
    iteration: int = 0
    extra_generation_params: dict = field(default_factory=dict)
    override_settings: dict = field(default_factory=dict)        # :156 — options for THIS job (API / pasted infotext), put back afterwards
    override_settings_restore_afterwards: bool = True             # :157
    keep_latents: bool = True

    def __post_init__(self):
        """modules/processing.py:246-251: unset sampler parameters fall back to the options; s_tmax 0 means infinity."""
        opts = shared.opts
        self.s_min_uncond = self.s_min_uncond if self.s_min_uncond is not None else opts.s_min_uncond
        self.s_churn = self.s_churn if self.s_churn is not None else opts.s_churn
        self.s_tmin = self.s_tmin if self.s_tmin is not None else opts.s_tmin
        self.s_tmax = (self.s_tmax if self.s_tmax is not None else opts.s_tmax) or float('inf')
        self.s_noise = self.s_noise if self.s_noise is not None else opts.s_noise

    def init(self, all_prompts, all_seeds, all_subseeds):
        pass

    # ---- image conditioning (c_concat) of inpainting / edit checkpoints: modules/processing.py:100-133, 298-393 -------------
    def _conditioning_key(self):
        return getattr(getattr(self.sd_model, "model", None), "conditioning_key", "crossattn")

    def txt2img_image_conditioning(self, x, width=None, height=None):
        """:100-133, 298-301.  Inpainting checkpoints: everything is masked, so the "masked image" is flat 0.5, encoded, behind a
        mask channel of ones; ordinary checkpoints get the reference's dummy [B,5,1,1] zeros (never read)."""
        b = x.shape[0]
        if self._conditioning_key() in {'hybrid', 'concat'} or getattr(self.sd_model, "is_sdxl_inpaint", False):
            self.is_using_inpainting_conditioning = self._conditioning_key() in {'hybrid', 'concat'}
            # image size = latent size x the first stage's downscale (= width / height for the real 8x VAE; test VAEs are shallower)
            f = 2 ** (len(self.sd_model.vae_cfg.ch_mult) - 1)
            image = torch.zeros((b, 3, x.shape[2] * f, x.shape[3] * f), dtype=torch.float32, device=x.device)   # 0.5 * 2 - 1
            latent = self.sd_model.get_first_stage_encoding(self.sd_model.encode_first_stage(image))
            ones = torch.ones((b, 1, *latent.shape[2:]), dtype=torch.float32, device=x.device)
            return torch.cat([ones, latent], dim=1).contiguous()
        if self._conditioning_key() == "crossattn-adm":      # unCLIP (:113-115): no image to embed in txt2img -> a zero c_adm
            na = getattr(self.sd_model, "noise_augmentor", None)
            dim = 2 * na.time_embed.dim if na is not None else self.sd_model.unet_cfg.adm_in_channels
            return x.new_zeros(b, dim)
        return x.new_zeros(b, 5, 1, 1)

    def depth2img_image_conditioning(self, source_image, latent_hw):
        """:304-320.  The MiDaS depth model and its input transform (ldm.data.util.AddMiDaS) belong to the host application (they are
        part of the depth2img checkpoint / the ldm package, torch modules that run once per job); the depth map is resized to the latent
        grid (bicubic) and normalised to [-1, 1] over the whole batch, and travels to the UNet as the fifth input channel (c_concat).
        ``latent_hw``: the reference encodes the image a second time only to read this shape off the result."""
        depth_model = getattr(self.sd_model, "depth_model", None)
        if depth_model is None:
            raise NotImplementedError("depth2img checkpoints need sd_model.depth_model (the checkpoint's MiDaS module)")
        try:
            from ldm.data.util import AddMiDaS
        except ImportError as e:
            raise NotImplementedError("depth2img conditioning needs ldm.data.util.AddMiDaS (present in a webui install)") from e
        # the FIRST image's MiDaS input stands for the whole batch, as in the reference (:307-309)
        sample = AddMiDaS(model_type="dpt_hybrid")({"jpg": source_image[0].permute(1, 2, 0)})       # HWC in [-1, 1] -> the network's CHW input
        midas_in = torch.from_numpy(sample["midas_in"]).to(device=source_image.device)
        depth = depth_model(midas_in.unsqueeze(0).repeat(self.batch_size, 1, 1, 1))
        depth = torch.nn.functional.interpolate(depth, size=tuple(latent_hw), mode="bicubic", align_corners=False)
        lo, hi = torch.aminmax(depth)
        return (2. * (depth - lo) / (hi - lo) - 1.).float().contiguous()

    def unclip_image_conditioning(self, source_image):
        """:327-333: c_adm = CLIP image embedding of the source image (+ the noise-level embedding at level 0) — the host's
        embedder / noise augmentor modules; the CFG denoiser hands it to the UNet's vector input, zeros on the uncond rows."""
        embedder = getattr(self.sd_model, "embedder", None)
        if embedder is None:
            raise NotImplementedError("unCLIP checkpoints need sd_model.embedder (the checkpoint's CLIP image embedder)")
        c_adm = embedder(source_image)
        na = getattr(self.sd_model, "noise_augmentor", None)
        if na is not None:
            noise_level = torch.zeros((c_adm.shape[0],), dtype=torch.long, device=c_adm.device)
            c_adm, noise_level_emb = na(c_adm, noise_level=noise_level)
            c_adm = torch.cat((c_adm, noise_level_emb), 1)
        return c_adm.float().contiguous()

    def inpainting_image_conditioning(self, source_image, latent_image, image_mask=None, round_image_mask=True):
        """:332-374.  ``source_image`` [B,3,H,W] in [-1,1] on the device; ``image_mask`` a float tensor [1,1,H,W] in [0,1]
        (1 = repaint).  c_concat = cat([mask at latent size (nearest), encode(lerp(image, image * (1 - mask), weight))])."""
        self.is_using_inpainting_conditioning = True
        dev = source_image.device
        if image_mask is not None:
            conditioning_mask = image_mask.to(dev, torch.float32).reshape(-1, 1, *image_mask.shape[-2:])
            if round_image_mask:
                conditioning_mask = torch.round(conditioning_mask)
        else:
            conditioning_mask = source_image.new_ones(1, 1, *source_image.shape[-2:])
        weight = self.inpainting_mask_weight if self.inpainting_mask_weight is not None else shared.opts.inpainting_mask_weight
        # lerp(s, s * (1 - m), w) = s * (1 - w * m): one blend launch with keep = 1 - w * m
        keep = ops.lincomb(torch.empty_like(conditioning_mask), [torch.ones_like(conditioning_mask), conditioning_mask.contiguous()],
                           [1.0, -float(weight)])
        zeros = torch.zeros_like(source_image)
        conditioning_image = ops.mask_blend(source_image.clone().contiguous(), zeros, zeros, keep.expand_as(source_image))
        conditioning_image = self.sd_model.get_first_stage_encoding(self.sd_model.encode_first_stage(conditioning_image))
        conditioning_mask = ops.latent_resize(conditioning_mask.contiguous(), tuple(latent_image.shape[-2:]), "nearest")
        conditioning_mask = conditioning_mask.expand(conditioning_image.shape[0], -1, -1, -1)
        return torch.cat([conditioning_mask, conditioning_image], dim=1).contiguous()

    def edit_image_conditioning(self, source_image):
        """:321-324: InstructPix2Pix conditions on the UNSCALED posterior mode of the source image."""
        mean, _ = torch.chunk(self.sd_model.encode_first_stage(source_image), 2, dim=1)
        return mean.contiguous()

    def img2img_image_conditioning(self, source_image, latent_image, image_mask=None, round_image_mask=True):
        """:376-398, in the reference's order: depth2img, edit, inpainting, unCLIP, SDXL inpainting, dummy."""
        if getattr(self.sd_model, "is_depth2img", False):       # the reference asks isinstance(sd_model, LatentDepth2ImageDiffusion)
            return self.depth2img_image_conditioning(source_image, latent_image.shape[2:])
        if getattr(self.sd_model, "cond_stage_key", "txt") == "edit":
            return self.edit_image_conditioning(source_image)
        if self._conditioning_key() in {'hybrid', 'concat'}:
            return self.inpainting_image_conditioning(source_image, latent_image, image_mask=image_mask, round_image_mask=round_image_mask)
        if self._conditioning_key() == "crossattn-adm":
            return self.unclip_image_conditioning(source_image)
        if getattr(self.sd_model, "is_sdxl_inpaint", False):
            return self.inpainting_image_conditioning(source_image, latent_image, image_mask=image_mask)
        return latent_image.new_zeros(latent_image.shape[0], 5, 1, 1)

    def sample(self, conditioning, unconditional_conditioning, seeds, subseeds, subseed_strength, prompts):
        raise NotImplementedError()

    def close(self):
        self.sampler = None


@dataclass
class StableDiffusionProcessingTxt2Img(StableDiffusionProcessing):
    enable_hr: bool = False
    denoising_strength: float = 0.75
    hr_scale: float = 2.0
    hr_upscaler: str = "Latent"
    hr_second_pass_steps: int = 0
    hr_resize_x: int = 0
    hr_resize_y: int = 0
    hr_scheduler: str = None
    hr_sampler_name: str = None
    hr_c: Optional[torch.Tensor] = None              # hires prompt conditioning (calculate_hr_conds, :1468-1490); default: p.c / p.uc
    hr_uc: Optional[torch.Tensor] = None
    hr_sd_model: Any = None                          # hires checkpoint (:1253-1259, 1360-1361): another resident SdModel
    firstpass_image: Any = None                      # a PIL image to run the hires pass ON instead of generating the first pass (:182, 1310-1332)
    hr_upscale_to_x: int = 0
    hr_upscale_to_y: int = 0
    truncate_x: int = 0
    truncate_y: int = 0

    def calculate_target_resolution(self):
        """modules/processing.py:1213-1250 (without the pre-1.0 compatibility option)."""
        if self.hr_resize_x == 0 and self.hr_resize_y == 0:
            self.hr_upscale_to_x = int(self.width * self.hr_scale)
            self.hr_upscale_to_y = int(self.height * self.hr_scale)
        elif self.hr_resize_y == 0:
            self.hr_upscale_to_x = self.hr_resize_x
            self.hr_upscale_to_y = self.hr_resize_x * self.height // self.width
        elif self.hr_resize_x == 0:
            self.hr_upscale_to_x = self.hr_resize_y * self.width // self.height
            self.hr_upscale_to_y = self.hr_resize_y
        else:
            target_w, target_h = self.hr_resize_x, self.hr_resize_y
            if self.width / self.height < self.hr_resize_x / self.hr_resize_y:
                self.hr_upscale_to_x = self.hr_resize_x
                self.hr_upscale_to_y = self.hr_resize_x * self.height // self.width
            else:
                self.hr_upscale_to_x = self.hr_resize_y * self.width // self.height
                self.hr_upscale_to_y = self.hr_resize_y
            self.truncate_x = (self.hr_upscale_to_x - target_w) // opt_f
            self.truncate_y = (self.hr_upscale_to_y - target_h) // opt_f

    def init(self, all_prompts, all_seeds, all_subseeds):
        """:1252-1305: target size, latent vs image-space upscaler."""
        if self.enable_hr:
            self.latent_scale_mode = {"Latent": "bilinear", "Latent (nearest)": "nearest", "Latent (bicubic)": "bicubic",
                                      "Latent (nearest-exact)": "nearest-exact", "Latent (antialiased)": "bilinear",
                                      "Latent (bicubic antialiased)": "bicubic"}.get(self.hr_upscaler)       # shared.py:54-62
            self.latent_scale_antialias = self.hr_upscaler in ("Latent (antialiased)", "Latent (bicubic antialiased)")
            if self.latent_scale_mode is None:
                from . import upscaler
                if not shared.sd_upscalers:
                    shared.sd_upscalers = upscaler.builtin_upscalers()
                if not any(x.name == self.hr_upscaler for x in shared.sd_upscalers):
                    raise Exception(f"could not find upscaler named {self.hr_upscaler}")                    # :1285-1286
            self.calculate_target_resolution()
            # the infotext keys of :1223-1228, 1254, 1262-1265, 1278, 1301-1305 (the hires prompt entries belong to the text layer)
            info = self.extra_generation_params
            info["Denoising strength"] = self.denoising_strength                                            # :1254
            if self.hr_resize_x == 0 and self.hr_resize_y == 0:
                info["Hires upscale"] = self.hr_scale
            else:
                info["Hires resize"] = f"{self.hr_resize_x}x{self.hr_resize_y}"
            if self.hr_sd_model is not None:
                info["Hires checkpoint"] = getattr(self.hr_sd_model, "short_title", None) or getattr(self.hr_sd_model, "name", "hires checkpoint")
            if self.hr_sampler_name is not None and self.hr_sampler_name != self.sampler_name:
                info["Hires sampler"] = self.hr_sampler_name
            info.setdefault("Hires schedule type", None)       # set by KDiffusionSampler.get_sigmas during the second pass
            if self.hr_second_pass_steps:
                info["Hires steps"] = self.hr_second_pass_steps
            if self.hr_upscaler is not None:
                info["Hires upscaler"] = self.hr_upscaler

    def sample(self, conditioning, unconditional_conditioning, seeds, subseeds, subseed_strength, prompts):
        """modules/processing.py:1307-1362"""
        self.sampler = sd_samplers.create_sampler(self.sampler_name, self.sd_model)
        if self.firstpass_image is not None and self.enable_hr:
            # :1310-1332 — no first pass: the given picture is what the hires pass starts from.  Image-space upscalers take it as the
            # "decoded first pass" in [-1, 1]; latent upscalers get its VAE encoding (images_tensor_to_samples, the 'Full' method: the
            # engine has no approximate encoders)
            if getattr(shared.opts, "sd_vae_encode_method", "Full") != "Full":
                raise NotImplementedError(f"sd_vae_encode_method {shared.opts.sd_vae_encode_method!r}: the engine encodes with the full VAE only")
            image = np.moveaxis(np.array(self.firstpass_image).astype(np.float32) / 255.0, 2, 0)
            image = torch.from_numpy(np.expand_dims(image, 0)).to(self.sd_model.device, dtype=torch.float32)
            image = ops.lincomb(torch.empty_like(image), [image, torch.ones_like(image)], [2.0, -1.0])        # image * 2 - 1
            if self.latent_scale_mode is None:
                samples, decoded_samples = None, image
            else:
                samples, decoded_samples = self.sd_model.get_first_stage_encoding(self.sd_model.encode_first_stage(image)), None
            # the reference carries ONE row here and lets torch broadcast it against the batch's noise and conds in the second pass
            # (:1429-1454); the engine's kernels take whole batches, so the row is repeated for every image of the batch
            nb = len(seeds) if seeds is not None else self.batch_size
            if nb > 1:
                if samples is not None:
                    samples = samples.expand(nb, *samples.shape[1:]).contiguous()
                if decoded_samples is not None:
                    decoded_samples = decoded_samples.expand(nb, *decoded_samples.shape[1:]).contiguous()
        else:
            x = self.rng.next()
            samples = self.sampler.sample(self, x, conditioning, unconditional_conditioning,
                                          image_conditioning=self.txt2img_image_conditioning(x))
            del x
            if not self.enable_hr:
                return samples
            decoded_samples = None
            if self.latent_scale_mode is None:               # image-space upscaler: the first pass is decoded (:1353-1354)
                # decoded by the model that is loaded NOW (the refiner, if it switched in during the first pass): :1353-1354 use shared.sd_model
                decoded_samples = decode_latent_batch(self.sampler.sd_model if self.sampler is not None else self.sd_model, samples)
        first_model = self.sd_model
        if self.hr_sd_model is not None:                     # :1360-1361 reload_model_weights(hr_checkpoint_info): both stay resident
            self.sd_model = self.hr_sd_model
        try:
            return self.sample_hr_pass(samples, decoded_samples, seeds, subseeds, subseed_strength, prompts, conditioning,
                                       unconditional_conditioning)
        finally:
            self.sd_model = first_model

    def sample_hr_pass(self, samples, decoded_samples, seeds, subseeds, subseed_strength, prompts, conditioning, unconditional_conditioning):
        """modules/processing.py:1364-1464.  Latent upscalers (shared.py:54-62) resample the undecoded first-pass latents on the
        device; image-space upscalers get the decoded uint8 images (PIL, exactly as the reference hands them to
        images.resize_image) and the result is re-encoded.  Fresh ImageRNG noise; second pass = sample_img2img at hr size."""
        if shared.state.interrupted:                         # :1365-1366 — an interrupted first pass is returned as it is
            return samples
        self.is_hr_pass = True
        target_w, target_h = self.hr_upscale_to_x, self.hr_upscale_to_y
        name = self.hr_sampler_name or self.sampler_name
        self.sampler = sd_samplers.create_sampler(name, self.sd_model)
        if self.latent_scale_mode is not None:
            # K16: [B,4,h,w] resample = F.interpolate(..., mode, antialias) (modules/processing.py:1392)
            samples = ops.latent_resize(samples, (target_h // opt_f, target_w // opt_f), self.latent_scale_mode,
                                        antialias=getattr(self, "latent_scale_antialias", False))
            # :1395-1399 (at the default mask weight 1.0 the hires pass of an inpainting checkpoint is conditioned like txt2img)
            weight = self.inpainting_mask_weight if self.inpainting_mask_weight is not None else shared.opts.inpainting_mask_weight
            if weight < 1.0 and self._conditioning_key() in {'hybrid', 'concat'}:
                image_conditioning = self.img2img_image_conditioning(self.sd_model.decode_first_stage(samples), samples)
            else:
                image_conditioning = self.txt2img_image_conditioning(samples, target_w, target_h)
        else:
            from PIL import Image
            from . import upscaler
            u8 = ops.image_to_u8(decoded_samples).cpu().numpy()          # clamp((x + 1) / 2) * 255 -> uint8 HWC (:1401-1406)
            batch_images = []
            for x_sample in u8:
                image = upscaler.resize_image(0, Image.fromarray(x_sample), target_w, target_h, upscaler_name=self.hr_upscaler)
                batch_images.append(np.moveaxis(np.array(image).astype(np.float32) / 255.0, 2, 0))
            decoded = torch.from_numpy(np.array(batch_images)).to(decoded_samples.device, dtype=torch.float32)
            image = ops.lincomb(torch.empty_like(decoded), [decoded, torch.ones_like(decoded)], [2.0, -1.0])   # image * 2 - 1
            samples = self.sd_model.get_first_stage_encoding(self.sd_model.encode_first_stage(image))          # :1421
            image_conditioning = self.img2img_image_conditioning(decoded, samples)       # (sic) the [0,1] image, as :1423 passes it
        if hasattr(shared.state, "nextjob"):
            shared.state.nextjob()                            # :1425
        # :1427
        samples = samples[:, :, self.truncate_y // 2:samples.shape[2] - (self.truncate_y + 1) // 2,
                          self.truncate_x // 2:samples.shape[3] - (self.truncate_x + 1) // 2].contiguous()
        if image_conditioning.shape[-2:] != (1, 1) and image_conditioning.shape[-2:] != samples.shape[-2:]:
            image_conditioning = image_conditioning[:, :, self.truncate_y // 2:image_conditioning.shape[2] - (self.truncate_y + 1) // 2,
                                                    self.truncate_x // 2:image_conditioning.shape[3] - (self.truncate_x + 1) // 2].contiguous()
        self.rng = ImageRNG(samples.shape[1:], self.seeds, subseeds=getattr(self, "subseeds", None), subseed_strength=self.subseed_strength,
                            seed_resize_from_h=self.seed_resize_from_h, seed_resize_from_w=self.seed_resize_from_w,
                            eta_noise_seed_delta=shared.opts.eta_noise_seed_delta, device=samples.device)     # :1429
        noise = self.rng.next()
        lo = self.iteration * self.batch_size
        hr_c = conditioning if self.hr_c is None else prompt_parser.slice_conds(self.hr_c, lo, lo + self.batch_size, samples.device)
        hr_uc = unconditional_conditioning if self.hr_uc is None else prompt_parser.slice_conds(self.hr_uc, lo, lo + self.batch_size, samples.device)
        samples = self.sampler.sample_img2img(self, samples, noise, hr_c, hr_uc,
                                              steps=self.hr_second_pass_steps or self.steps, image_conditioning=image_conditioning)
        self.is_hr_pass = False
        return samples


@dataclass
class StableDiffusionProcessingImg2Img(StableDiffusionProcessing):
    init_images: Any = None                       # float tensor [N,3,H,W] in [0,1] (the reference builds it at :1700-1728)
    denoising_strength: float = 0.75
    latent_mask: Optional[torch.Tensor] = None    # [N,4,h,w] latent-space mask (1 = keep original), optional
    image_mask: Optional[torch.Tensor] = None     # [1,1,H,W] image-space mask in [0,1] (1 = repaint): conditions inpainting checkpoints
    mask_round: bool = True
    inpainting_fill: int = 1                      # 0 fill, 1 original, 2 latent noise, 3 latent nothing (0 needs the PIL inputs below)
    initial_noise_multiplier: float = None        # opts.initial_noise_multiplier
    # The reference's own PIL front-end (modules/processing.py:1602-1728), taken when init_images is a list of PIL images: `mask_image`
    # is the reference's `mask` argument (RGBA from the UI or an L / RGB image), `latent_mask_image` its PIL `latent_mask`
    mask_image: Any = None
    latent_mask_image: Any = None
    mask_blur_x: int = 4
    mask_blur_y: int = 4
    mask_blur: int = None                         # sets both (the reference's property, :1592-1600)
    inpainting_mask_invert: int = 0
    inpaint_full_res: bool = True
    inpaint_full_res_padding: int = 0
    resize_mode: int = 0
    overlay_images: Any = None
    paste_to: Any = None
    mask_for_overlay: Any = None
    image_cfg_scale: float = None                 # InstructPix2Pix
    init_latent: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None
    nmask: Optional[torch.Tensor] = None
    image_conditioning_all: Optional[torch.Tensor] = None

    def init(self, all_prompts, all_seeds, all_subseeds):
        """modules/processing.py:1602-1757 from the point where the image tensor exists: image*2-1 -> VAE encode -> latent
        (posterior mean, deterministic), the "masked content" fills 2 / 3 over the latent mask (:1747-1753) and the image
        conditioning of inpainting / edit checkpoints (:1755)."""
        self.extra_generation_params["Denoising strength"] = self.denoising_strength       # :1603
        if getattr(self.sd_model, "cond_stage_key", "txt") != "edit":
            self.image_cfg_scale = None                                                    # :1605
        if self.initial_noise_multiplier is None:
            self.initial_noise_multiplier = shared.opts.initial_noise_multiplier
        if isinstance(self.init_images, (list, tuple)) and len(self.init_images) and not torch.is_tensor(self.init_images[0]):
            self._init_from_pil()
        image = (self.init_images.to(self.sd_model.device, dtype=torch.float32) * 2.0 - 1.0).contiguous()
        moments = self.sd_model.encode_first_stage(image)
        self.init_latent_all = self.sd_model.get_first_stage_encoding(moments)
        if self.resize_mode == 3:                                                          # :1733-1734 "Just resize (latent upscale)"
            self.init_latent_all = ops.latent_resize(self.init_latent_all.contiguous(), (self.height // opt_f, self.width // opt_f), "bilinear")
        if getattr(self, "_latent_mask_pil", None) is not None:
            self._latent_mask_from_pil()
        if self.latent_mask is not None and self.inpainting_fill in (2, 3):
            dev = self.init_latent_all.device
            keep = self.latent_mask.to(dev, torch.float32).expand_as(self.init_latent_all).contiguous()
            if self.inpainting_fill == 2:         # init * mask + create_random_tensors(shape, all_seeds[:N]) * nmask
                fill = ImageRNG(tuple(self.init_latent_all.shape[1:]), all_seeds[0:self.init_latent_all.shape[0]], device=dev).next()
                self.extra_generation_params["Masked content"] = 'latent noise'
            else:                                 # init * mask
                fill = torch.zeros_like(self.init_latent_all)
                self.extra_generation_params["Masked content"] = 'latent nothing'
            self.init_latent_all = ops.mask_blend(fill.contiguous(), self.init_latent_all, keep, (1.0 - keep).contiguous())
        elif self.inpainting_fill not in (0, 1, 2, 3):
            raise ValueError(f"inpainting_fill {self.inpainting_fill!r}")
        elif self.inpainting_fill == 0 and self.latent_mask is not None and self.mask_for_overlay is None:
            raise ValueError("inpainting_fill 0 ('fill') repaints the IMAGE before encoding: pass PIL init_images / mask_image")
        self.image_conditioning_all = self.img2img_image_conditioning(image, self.init_latent_all, self.image_mask, self.mask_round)

    def _init_from_pil(self):
        """modules/processing.py:1608-1745, the image side: binary mask -> invert -> separable Gaussian blur -> either the "only
        masked" crop (region grown to the processing aspect ratio, image and mask resized to it, paste_to remembered) or the plain resize
        (mask doubled for the overlay), overlay images (the unmasked original with the mask as transparency), the "fill" repaint for
        every masked-content mode but "original", uint8 -> [0, 1] tensors, and the latent mask (mask resized to the latent grid,
        rounded, 1 = keep).  Leaves tensors in init_images / latent_mask / image_mask: the tensor path above continues from there."""
        from PIL import Image, ImageOps
        from . import masking, upscaler
        if self.mask_blur is not None:
            self.mask_blur_x = self.mask_blur_y = int(self.mask_blur)
        crop_region = None
        image_mask = self.mask_image
        if image_mask is not None:
            image_mask = masking.create_binary_mask(image_mask, round=self.mask_round)
            if self.inpainting_mask_invert:
                image_mask = ImageOps.invert(image_mask)
                self.extra_generation_params["Mask mode"] = "Inpaint not masked"
            image_mask = masking.blur_mask(image_mask, self.mask_blur_x, self.mask_blur_y)
            if self.mask_blur_x > 0 or self.mask_blur_y > 0:
                self.extra_generation_params["Mask blur"] = self.mask_blur_x if self.mask_blur_x == self.mask_blur_y else None
            if self.inpaint_full_res:
                self.mask_for_overlay = image_mask
                mask = image_mask.convert('L')
                crop_region = masking.get_crop_region_v2(mask, self.inpaint_full_res_padding)
                if crop_region:
                    crop_region = masking.expand_crop_region(crop_region, self.width, self.height, mask.width, mask.height)
                    x1, y1, x2, y2 = crop_region
                    image_mask = upscaler.resize_image(2, mask.crop(crop_region), self.width, self.height)
                    self.paste_to = (x1, y1, x2 - x1, y2 - y1)
                    self.extra_generation_params["Inpaint area"] = "Only masked"
                    self.extra_generation_params["Masked area padding"] = self.inpaint_full_res_padding
                else:                                         # blank mask: plain img2img (:1647-1654)
                    crop_region, image_mask, self.mask_for_overlay, self.inpaint_full_res = None, None, None, False
            else:
                image_mask = upscaler.resize_image(self.resize_mode, image_mask, self.width, self.height)
                doubled = np.clip(np.array(image_mask).astype(np.float32) * 2, 0, 255).astype(np.uint8)
                self.mask_for_overlay = Image.fromarray(doubled)
            self.overlay_images = []
        latent_mask = self.latent_mask_image if self.latent_mask_image is not None else image_mask
        imgs = []
        for img in self.init_images:
            image = img
            if image.mode == "RGBA":                          # images.flatten (modules/images.py:841-849): transparency -> background colour
                flat = Image.new('RGBA', image.size, getattr(shared.opts, "img2img_background_color", "#ffffff"))
                flat.paste(image, mask=image)
                image = flat
            image = image.convert('RGB')
            if crop_region is None and self.resize_mode != 3:
                image = upscaler.resize_image(self.resize_mode, image, self.width, self.height)
            if image_mask is not None:
                if self.mask_for_overlay.size != (image.width, image.height):
                    self.mask_for_overlay = upscaler.resize_image(self.resize_mode, self.mask_for_overlay, image.width, image.height)
                kept = Image.new('RGBa', (image.width, image.height))
                kept.paste(image.convert("RGBA").convert("RGBa"), mask=ImageOps.invert(self.mask_for_overlay.convert('L')))
                self.overlay_images.append(kept.convert('RGBA'))
            if crop_region is not None:
                image = upscaler.resize_image(2, image.crop(crop_region), self.width, self.height)
            if image_mask is not None and self.inpainting_fill != 1:
                image = masking.fill(image, latent_mask)
                if self.inpainting_fill == 0:
                    self.extra_generation_params["Masked content"] = 'fill'
            imgs.append(np.moveaxis(np.array(image).astype(np.float32) / 255.0, 2, 0))
        if len(imgs) == 1:
            batch = np.expand_dims(imgs[0], axis=0).repeat(self.batch_size * self.n_iter, axis=0)
            if self.overlay_images is not None:
                self.overlay_images = self.overlay_images * (self.batch_size * self.n_iter)
        elif len(imgs) <= self.batch_size:                    # :1718-1720: fewer images than the batch size shrink the batch; every
            self.batch_size = len(imgs)                       # iteration then starts from the same images
            batch = np.array(imgs).repeat(1, axis=0)
            batch = np.concatenate([batch] * self.n_iter, axis=0)
            if self.overlay_images is not None:
                self.overlay_images = self.overlay_images * self.n_iter
        elif len(imgs) == self.batch_size * self.n_iter:      # (extension: one init image per image of the whole job)
            batch = np.array(imgs)
        else:
            raise RuntimeError(f"bad number of images passed: {len(imgs)}; expecting {self.batch_size * self.n_iter} or less")
        self.init_images = torch.from_numpy(batch)
        self._latent_mask_pil = None
        if image_mask is not None:
            self._latent_mask_pil = latent_mask                 # resized to the latent grid once the init latent exists (init)
            cond = np.array(image_mask.convert("L")).astype(np.float32) / 255.0
            self.image_mask = torch.from_numpy(cond[None, None])
        elif self.mask_image is not None:                      # a mask was given but is blank after the inversion / blur: plain img2img
            self.latent_mask = self.image_mask = None
        # (no PIL mask at all: tensor latent_mask / image_mask a caller set on the object stay as they are)

    def _latent_mask_from_pil(self):
        """modules/processing.py:1733-1742: the mask image resized (PIL default filter) to the init latent's grid, red channel / 255,
        rounded when mask_round, tiled over the latent channels; stored as 1 = keep (the reference's self.mask = 1 - latmask)."""
        n, c, h, w = self.init_latent_all.shape
        latmask = self._latent_mask_pil.convert('RGB').resize((w, h))
        latmask = np.moveaxis(np.array(latmask, dtype=np.float32), 2, 0)[0] / 255
        if self.mask_round:
            latmask = np.around(latmask)
        self.latent_mask = torch.from_numpy(1.0 - np.tile(latmask[None, None], (n, c, 1, 1))).float()

    def sample(self, conditioning, unconditional_conditioning, seeds, subseeds, subseed_strength, prompts):
        """modules/processing.py:1759-1789"""
        lo = self.iteration * self.batch_size
        self.init_latent = self.init_latent_all[lo:lo + self.batch_size].contiguous()
        if self.latent_mask is not None:
            dev = self.init_latent.device
            self.mask = self.latent_mask[lo:lo + self.batch_size].to(dev, torch.float32).expand_as(self.init_latent).contiguous()
            self.nmask = (1.0 - self.mask).contiguous()
        x = self.rng.next()
        if tuple(x.shape) != tuple(self.init_latent.shape):
            # the reference draws noise at (4, height // 8, width // 8) and adds it to the encoded image: a VAE with another downscale
            # factor, or init tensors of another size than width x height, is a shape error there (torch broadcast) and must be one here
            raise ValueError(f"img2img: init latent {tuple(self.init_latent.shape)} does not match the noise shape {tuple(x.shape)} "
                             f"of a {self.width}x{self.height} job (init_images must encode to (4, height // {opt_f}, width // {opt_f}))")
        if self.initial_noise_multiplier != 1.0:                                # :1762-1764
            self.extra_generation_params["Noise multiplier"] = self.initial_noise_multiplier
            x = ops.lincomb(x, [x], [float(self.initial_noise_multiplier)])
        self.sampler = sd_samplers.create_sampler(self.sampler_name, self.sd_model)
        samples = self.sampler.sample_img2img(self, self.init_latent, x, conditioning, unconditional_conditioning,
                                              image_conditioning=self.image_conditioning_all[lo:lo + self.batch_size].contiguous())
        if self.mask is not None:                                               # :1776-1784 final blend (+ Script.on_mask_blend, is_final_blend)
            blended = ops.mask_blend(samples.contiguous().clone(), self.init_latent, self.mask, self.nmask)
            runner = getattr(self, "scripts", None)
            if runner is not None and hasattr(runner, "on_mask_blend"):
                mba = shared.MaskBlendArgs(samples, self.nmask, self.init_latent, self.mask, blended)
                runner.on_mask_blend(self, mba)
                blended = mba.blended_latent
            samples = blended
        return samples


class Processed:
    """modules/processing.py:516-613 (fields the callers on this path read)."""

    def __init__(self, p, images_list, seed=-1, all_seeds=None, latents=None, images_device=None):
        self.images = images_list
        self.images_device = images_device                   # [n, H, W, 3] uint8 on the GPU: the same images before the host copy (None
                                                             # when a host-side overlay was composited over them) — parallel.py gathers from it
        self.seed = seed
        self.all_seeds = all_seeds or [seed]
        self.width, self.height = p.width, p.height
        self.sampler_name, self.cfg_scale, self.steps = p.sampler_name, p.cfg_scale, p.steps
        self.batch_size = p.batch_size
        self.latents = latents
        # the sampling-side fields of the reference's Processed (:517-563) a caller may read back; the text / file / model-name fields
        # (prompt, info, infotexts, sd_model_hash ...) belong to layers outside this path
        self.prompt = p.prompt if not isinstance(p.prompt, list) else (p.prompt[0] if p.prompt else "")
        self.negative_prompt = p.negative_prompt if not isinstance(p.negative_prompt, list) else (p.negative_prompt[0] if p.negative_prompt else "")
        self.all_subseeds = list(getattr(p, "all_subseeds", None) or [])
        self.subseed = self.all_subseeds[0] if self.all_subseeds else -1
        self.subseed_strength = p.subseed_strength
        self.seed_resize_from_w, self.seed_resize_from_h = p.seed_resize_from_w, p.seed_resize_from_h
        self.image_cfg_scale = getattr(p, "image_cfg_scale", None)
        self.denoising_strength = getattr(p, "denoising_strength", None)
        self.extra_generation_params = p.extra_generation_params
        self.index_of_first_image = 0
        self.eta, self.s_churn, self.s_tmin, self.s_tmax, self.s_noise, self.s_min_uncond = p.eta, p.s_churn, p.s_tmin, p.s_tmax, p.s_noise, p.s_min_uncond
        self.sampler_noise_scheduler_override = p.sampler_noise_scheduler_override


def decode_latent_batch(model, batch, target_device=None, check_for_nans=False):
    """modules/processing.py:625-672.  The reference decodes one image at a time; the engine decodes the whole batch in one batched
    pass (same per-image arithmetic, images are independent).  NaN handling follows the reference: the latents are probed first
    (NansException "unet"); a decoded image that probes NaN triggers the automatic precision fallback when
    opts.auto_vae_precision(_bfloat16) is on — the reference converts the VAE to fp32 / bf16 because fp16 activations of the SD
    decoders overflow on some images; the engine's equivalent is its RANGE-EXTENDED decode (sdmi option "vae_range_extend": the
    decoder's residual stream is carried at 1/64 scale, fp16 range x64, GroupNorm eps rescaled — same function, no overflow) —
    and the whole batch is decoded again; without the option the NansException propagates, as in the reference."""
    from . import devices
    if check_for_nans:
        devices.test_for_nans(batch, "unet")
    out = model.decode_first_stage(batch)
    if check_for_nans:
        try:
            # the reference probes element [0, 0, 0] of every decoded image; all B probes travel in ONE 4 * B byte read here
            if not shared.cmd_opts.disable_nan_check:
                bad = torch.isnan(out[:, 0, 0, 0]).nonzero()
                if bad.numel():
                    devices.test_for_nans(out[int(bad[0])], "vae")
        except devices.NansException as e:
            if not (shared.opts.auto_vae_precision_bfloat16 or shared.opts.auto_vae_precision):
                raise e
            if getattr(model, "vae_range_extended", False):          # already in the fallback mode (devices.dtype_vae == autofix_dtype)
                raise e
            print("A tensor with all NaNs was produced in VAE.\nThe engine will now switch the VAE decoder to its range-extended mode and retry.\n"
                  "To disable this behavior, disable the 'Automatically revert VAE to 32-bit floats' setting.")
            model.set_vae_range_extended(True)
            out = model.decode_first_stage(batch)
    if target_device is not None:
        out = out.to(target_device)
    return out


def job_devices() -> list:
    """``opts.mi355x_devices`` ("0,1,2,3"; empty = the model's own device): the devices of THIS process a job is spread over."""
    spec = getattr(shared.opts, "mi355x_devices", "") or ""
    if isinstance(spec, (list, tuple)):
        return [int(d) for d in spec]
    return [int(d) for d in str(spec).replace(";", ",").split(",") if d.strip() != ""]


def process_images(p: StableDiffusionProcessing) -> Processed:
    """modules/processing.py:819 — the entry point.  One process, N devices (the webui front-end stays a single
    process behind queue_lock, modules/call_queue.py:8-13): with more than one device in ``opts.mi355x_devices`` the job's
    batch_size x n_iter images are cut into contiguous ranges, one per device, each run by a worker thread on that device's own engine
    (parallel.DevicePool — the split of parallel.shard_job, so the images are the single-device job's, bit for bit)."""
    devs = job_devices()
    if len(devs) > 1 and p.batch_size * p.n_iter > 1:
        from . import parallel
        serial = bool(getattr(shared.opts, "mi355x_devices_serial", False))
        return parallel.process_images_devices(p, devs, runner=process_images_one_device, serial=serial)
    return process_images_one_device(p)


def process_images_one_device(p: StableDiffusionProcessing) -> Processed:
    """modules/processing.py:819-857: the job's option overrides are set for its duration and restored afterwards (the checkpoint / VAE
    entries belong to the model loader and are not options of this path), the sampler / scheduler names are brought to the table's spelling
    ("DPM++ 2M Karras" -> "DPM++ 2M" + "Karras"), then the job itself."""
    opts = shared.opts
    overrides = {k: v for k, v in (p.override_settings or {}).items() if k not in ("sd_model_checkpoint", "sd_vae")}
    unknown = [k for k in overrides if not hasattr(opts, k)]
    if unknown:
        raise KeyError(unknown[0])                            # modules/options.py:151: opts.set looks the key up in data_labels
    stored = {k: getattr(opts, k) for k in overrides}
    try:
        for k, v in overrides.items():
            setattr(opts, k, v)
        sd_samplers.fix_p_invalid_sampler_and_scheduler(p)
        return _process_images_inner(p)
    finally:
        if p.override_settings_restore_afterwards:
            for k, v in stored.items():
                setattr(opts, k, v)


def _process_images_inner(p: StableDiffusionProcessing) -> Processed:
    """modules/processing.py:860-1150 (process_images_inner) for the engine path."""
    assert p.c is not None and p.uc is not None, "conditioning tensors p.c / p.uc are required (text encoder is out of scope)"
    n_total = p.batch_size * p.n_iter
    seed = 1000 if p.seed is None or isinstance(p.seed, (list, tuple)) or p.seed == -1 else int(p.seed)
    # :901-909 — with a variation seed every image keeps the SAME seed and the subseeds count up instead
    p.all_seeds = list(p.seed) if isinstance(p.seed, (list, tuple)) else [seed + (i if p.subseed_strength == 0 else 0) for i in range(n_total)]
    subseed = 2000 if p.subseed is None or isinstance(p.subseed, (list, tuple)) or p.subseed == -1 else int(p.subseed)
    p.all_subseeds = list(p.subseed) if isinstance(p.subseed, (list, tuple)) else [subseed + i for i in range(n_total)]
    if p.tiling is None:                                             # :879-880
        p.tiling = shared.opts.tiling
    for m in (p.sd_model, getattr(p, "refiner_sd_model", None), getattr(p, "hr_sd_model", None)):
        if m is not None:                                            # :895 model_hijack.apply_circular(p.tiling): every padded conv
            m.engine.set_option("tiling", 1 if p.tiling else 0)      # of the UNet and the VAE wraps around (sd_hijack.py:311-318)
            # --no-half (modules/cmd_args.py; the reference then runs the model in fp32) / the engine's own accuracy switch: the UNet's
            # residual stream is carried with ~22 bits instead of fp16 (engine option "residual_fp32")
            if hasattr(m, "set_accuracy_mode"):
                m.set_accuracy_mode(bool(shared.cmd_opts.no_half or getattr(shared.opts, "sdmi_accuracy_mode", False)))
    sd_models.apply_alpha_schedule_override(p.sd_model, p)           # :930
    p.init(None, p.all_seeds, None)
    images, latents, device_u8 = [], [], []
    dev = p.sd_model.device
    for n in range(p.n_iter):
        p.iteration = n
        if shared.state.skipped:                                     # :935-939 — "Skip" ends one batch, "Interrupt" the whole job
            shared.state.skipped = False
        if shared.state.interrupted or getattr(shared.state, "stopping_generation", False):
            break
        shared.sd_model = p.sd_model                                 # :941 — the previous iteration may have ended on the refiner
        lo, hi = n * p.batch_size, (n + 1) * p.batch_size
        p.seeds = p.all_seeds[lo:hi]
        p.subseeds = p.all_subseeds[lo:hi]
        p.rng = ImageRNG((opt_C, p.height // opt_f, p.width // opt_f), p.seeds, subseeds=p.subseeds, subseed_strength=p.subseed_strength,
                         seed_resize_from_h=p.seed_resize_from_h, seed_resize_from_w=p.seed_resize_from_w,
                         eta_noise_seed_delta=shared.opts.eta_noise_seed_delta, device=dev)      # :949
        c, uc = prompt_parser.slice_conds(p.c, lo, hi, dev), prompt_parser.slice_conds(p.uc, lo, hi, dev)
        p_y_all, p_uy_all = p.y, p.uy
        try:
            if p_y_all is not None:
                p.y, p.uy = p_y_all[lo:hi].to(dev), p_uy_all[lo:hi].to(dev)
            samples = p.sample(conditioning=c, unconditional_conditioning=uc, seeds=p.seeds, subseeds=None,
                               subseed_strength=0, prompts=None)                                  # :987-988
        finally:                                                     # an interrupt / error must not leave p holding the batch slice
            p.y, p.uy = p_y_all, p_uy_all
        # the model that finished the sampling decodes (a hires checkpoint or the refiner is still "loaded" at :1002 / :1459)
        decode_model = p.sampler.sd_model if getattr(p, "sampler", None) is not None else p.sd_model
        x_samples = decode_latent_batch(decode_model, samples, check_for_nans=True)             # :1002 (hires: decoded inside sample_hr_pass, :1459)
        u8 = ops.image_to_u8(x_samples)                                                           # :1004-1005, 1034-1035
        device_u8.append(u8)
        overlays = getattr(p, "overlay_images", None)
        if not overlays and not getattr(p, "images_to_host", True):   # a non-zero rank of a sharded job: rank 0 gets them over RCCL
            if p.keep_latents:                                       # (the latents stay available on this rank either way)
                latents.append(samples)
            continue
        batch_u8 = list(u8.cpu().numpy())
        if overlays:                                                 # :1063-1068, 1086: paste the generated crop back, composite the unmasked original over it
            from PIL import Image
            from . import masking
            for i, arr in enumerate(batch_u8):
                ov = overlays[lo + i] if lo + i < len(overlays) and getattr(shared.opts, "overlay_inpaint", True) else None      # :1055-1060
                batch_u8[i] = np.array(masking.apply_overlay(Image.fromarray(arr), getattr(p, "paste_to", None), ov)[0])
        images.extend(batch_u8)
        if p.keep_latents:
            latents.append(samples)
    p.close()
    shared.sd_model = p.sd_model                                     # :941 reload_model_weights(): back from a refiner
    same_size = len(device_u8) > 0 and len({tuple(t.shape[1:]) for t in device_u8}) == 1 and not getattr(p, "overlay_images", None)
    return Processed(p, images, seed, p.all_seeds, torch.cat(latents) if latents else None,
                     images_device=(device_u8[0] if len(device_u8) == 1 else torch.cat(device_u8)) if same_size else None)
