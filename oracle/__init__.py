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
  * DDIM, DDIM CFG++, PLMS, UniPC (bh1 / bh2 / vary_coeff  <- modules/sd_samplers_timesteps_impl.py, modules/models/diffusion/uni_pc/uni_pc.py,
    at batch 1: the reference breaks beyond), Restart,        modules/sd_samplers_extra.py, modules/sd_samplers_lcm.py, sd3_impls.py:145-163
    LCM, Euler
  * every in-repo scheduler                               <- modules/sd_schedulers.py
  * CFGDenoiser.forward (20 scenarios), apply_refiner     <- modules/sd_samplers_cfg_denoiser.py, modules/sd_samplers_common.py:158-202
  * conditioning containers / per-step reconstruction     <- modules/prompt_parser.py:136-349
  * image conditioning, resize_image + Upscaler loop      <- modules/processing.py:100-133, 321-374; modules/images.py:252-291, modules/upscaler.py
  * LoRA layer naming, every LyCORIS module's calc_updown  <- extensions-builtin/Lora/networks.py:56-120, network*.py, lyco_helpers.py
    incl. the dense bias entry (ex_bias)
  * hypernetwork modules and their chained application     <- modules/hypernetworks/hypernetwork.py:25-113, 358-379
  * alpha-schedule overrides (zero terminal SNR, fp16)     <- modules/sd_models.py:553-589
The img2img / inpainting front-end (masking.py of the PACKAGE, not part of this oracle) is pinned the same way by
tests/golden/img2img_frontend.npz <- modules/masking.py, modules/images.py:252-291, modules/processing.py:70-98; its Gaussian mask blur is
unpinned (the reference calls cv2.GaussianBlur; cv2 is not installed).
Full-size outputs of this oracle at the benchmarked shapes are committed as tests/golden/fullsize_*.npz (make_fullsize_golden.py) so that
the GPU parity tests at those shapes cost no oracle time on the GPU box.
  * structural checksums                                   <- parameter counts in SURVEY.md section 8(c)
PARITY UNPINNED: the remainder of the UNet forward (ResBlock, GEGLU feed-forward, block wiring) and the k-diffusion samplers other
than Euler, with get_sigmas_karras / _exponential / _polyexponential — the reference's tests hold no numeric vector for them and
the arithmetic lives in un-vendored third-party repos (ldm @ cf1d67a6, sgm @ 45c443b3, k-diffusion @ ab527a9a); they are restated
from the published algorithms and anchored on the in-tree call sites and state-dict layout.
What CAN be checked without those packages is (tests/test_oracle_properties.py): on Gaussian data the ideal denoiser, the probability-flow
solution and the SDE's marginals are closed-form, so every k-diffusion restatement must converge to the exact flow at its published order
(or keep the marginal variance), DPM++ SDE must be second order only on one Brownian path, and the Brownian tree must be a Brownian
motion with consistent nested increments.  A transcription error fails those tests; they are not a pin, and the status above stands.
"""
