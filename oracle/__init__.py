"""CPU oracle for the txt2img / img2img hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain fp32 PyTorch / numpy restatement of the arithmetic on the reference's hot path:
UNet forward, VAE decode / encode, the k-diffusion denoiser wrappers and samplers, the in-repo timestep samplers, the schedulers,
the classifier-free-guidance denoiser, the conditioning containers, the refiner switch, image conditioning of inpainting / edit
checkpoints, the image-space hires hand-off, the Philox ("NV") noise source with variation seeds and seed-resize, LoRA / LyCORIS
merging and the CLIP text transformer.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product path
(``stable-diffusion-webui_amd``) never does and fails loudly when the HIP library is missing.

Parity status.  PINNED by ``tests/test_oracle_pins.py`` against fixtures that ``tests/golden/make_golden.py`` generates by executing
the reference's own files (loaded by path with the webui modules they import stubbed, or — where a file cannot be imported — the
named functions exec'd from its text):
  * Philox randn, ImageRNG (subseeds, seed-resize, ENSD)   <- modules/rng_philox.py, modules/rng.py
  * attention                                             <- modules/sub_quadratic_attention.py, modules/hypernetworks/hypernetwork.py:382-407
  * timestep embedding, SpatialTransformer forward        <- modules/sd_hijack_unet.py:56-102
  * VAE decoder / encoder                                 <- modules/models/sd3/sd3_impls.py VAEDecoder / VAEEncoder
  * CLIP text transformer                                 <- modules/models/sd3/other_impls.py (+ the installed transformers CLIPTextModel)
  * DDIM, DDIM CFG++, PLMS, UniPC, Restart, LCM, Euler    <- modules/sd_samplers_timesteps_impl.py, modules/models/diffusion/uni_pc/uni_pc.py,
                                                             modules/sd_samplers_extra.py, modules/sd_samplers_lcm.py, sd3_impls.py:145-163
  * every in-repo scheduler                               <- modules/sd_schedulers.py
  * CFGDenoiser.forward (20 scenarios), apply_refiner     <- modules/sd_samplers_cfg_denoiser.py, modules/sd_samplers_common.py:158-202
  * conditioning containers / per-step reconstruction     <- modules/prompt_parser.py:136-349
  * image conditioning, resize_image + Upscaler loop      <- modules/processing.py:100-133, 321-374; modules/images.py:252-291, modules/upscaler.py
  * LoRA layer naming, every LyCORIS module's calc_updown  <- extensions-builtin/Lora/networks.py:56-120, network*.py, lyco_helpers.py
  * structural checksums                                   <- parameter counts in SURVEY.md section 8(c)
PARITY UNPINNED: the remainder of the UNet forward (ResBlock, GEGLU feed-forward, block wiring) and the k-diffusion samplers other
than Euler, with get_sigmas_karras / _exponential / _polyexponential — the reference's tests hold no numeric vector for them and
the arithmetic lives in un-vendored third-party repos (ldm @ cf1d67a6, sgm @ 45c443b3, k-diffusion @ ab527a9a); they are restated
from the published algorithms and anchored on the in-tree call sites and state-dict layout.
"""
