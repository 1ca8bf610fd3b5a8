"""CPU oracle for the txt2img hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain fp32 PyTorch/numpy restatement of the arithmetic on the
reference's txt2img/img2img hot path: UNet forward, VAE decode/encode,
the k-diffusion denoiser wrapper + Euler-a / Euler / DPM++ 2M / DDIM samplers,
classifier-free-guidance combine and the Philox ("NV") noise source.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path (``stable-diffusion-webui_amd``) never does and fails loudly when the HIP
library is missing.

Parity status: **parity unpinned for UNet / samplers** (the reference's tests hold no
numeric vector for them and the arithmetic lives in un-vendored third-party repos:
ldm @ cf1d67a6, sgm @ 45c443b3, k-diffusion @ ab527a9a).  Pinned pieces, checked by
``tests/test_oracle_pins.py`` against fixtures generated from the importable reference
files by ``tests/golden/make_golden.py``:
  * Philox randn            <- modules/rng_philox.py (docstring golden vector + generated)
  * attention               <- modules/sub_quadratic_attention.py
  * VAE decoder / encoder   <- modules/models/sd3/sd3_impls.py VAEDecoder/VAEEncoder
  * DDIM                    <- modules/sd_samplers_timesteps_impl.py:12-40
  * structural checksums    <- parameter counts in SURVEY.md section 8(c)
"""
