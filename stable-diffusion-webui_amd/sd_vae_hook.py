"""Boundary B4: replace ``sd_model.first_stage_model.decode`` with the engine's batched VAE decoder.

The reference has no VAE plugin API; the hook point is the method itself, the way lowvram swaps
``first_stage_model.encode/decode`` (modules/lowvram.py:65-75, 136-137).  ``decode_first_stage``
(modules/sd_samplers_common.py:73-76) keeps calling ``model.decode_first_stage(z)`` -> ``first_stage_model.decode(z / scale)``.
"""
from __future__ import annotations

import torch

from . import schema
from .engine import Engine


def install(sd_model, device_index: int = 0):
    """Idempotent: the webui fires on_model_loaded again on the SAME sd_model object after every in-place checkpoint reload
    (modules/sd_models.py:994) and every VAE switch (modules/sd_vae.py:280).  A re-install closes the previous engine, keeps the
    ORIGINAL torch decode as the thing uninstall() restores, and packs the (possibly new) first-stage weights afresh."""
    fsm = sd_model.first_stage_model
    if hasattr(fsm, "_mi355x_engine"):
        uninstall(sd_model)
    sd = {schema.VAE_PREFIX + k: v for k, v in fsm.state_dict().items()}
    cfg = schema.sdxl_vae() if getattr(sd_model, "is_sdxl", False) else schema.sd15_vae()
    cfg.scale_factor = 1.0                         # the caller already divided by scale_factor (ddpm_edit.py:734 twin)
    eng = Engine(device_index)
    eng.load_vae(cfg, sd, decoder_only=True)
    fsm._mi355x_engine = eng
    fsm._torch_decode = fsm.decode

    def decode(z, *a, **kw):
        from .webui_bridge import hypertile_active
        if hypertile_active(fsm):                  # opts.hypertile_enable_vae tiles the torch mid-block attention: that IS the torch decode
            return fsm._torch_decode(z, *a, **kw)
        return eng.vae_decode(z if z.dtype in (torch.float16, torch.float32) else z.float()).to(z.dtype)

    fsm.decode = decode
    return eng


def uninstall(sd_model):
    fsm = sd_model.first_stage_model
    if hasattr(fsm, "_torch_decode"):
        fsm.decode = fsm._torch_decode
        fsm._mi355x_engine.close()
        del fsm._torch_decode, fsm._mi355x_engine
