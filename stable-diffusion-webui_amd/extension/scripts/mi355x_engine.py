"""webui extension entry (boundary B0): drop this directory into ``extensions/mi355x-engine/`` of a webui checkout with the
repo root on ``sys.path``.  Registration is idempotent: the webui clears all callbacks and
re-imports scripts on "Reload UI" (modules/scripts.py:491).
"""
import importlib

from modules import script_callbacks, sd_models, shared

pkg = importlib.import_module("stable-diffusion-webui_amd")
sd_unet_amd = importlib.import_module("stable-diffusion-webui_amd.sd_unet")
sd_opt_amd = importlib.import_module("stable-diffusion-webui_amd.sd_hijack_optimizations")


def _list_unets(options):
    for info in sd_models.checkpoints_list.values():
        provider = (lambda info=info: sd_models.read_state_dict(info.filename, map_location="cpu"))
        options.append(sd_unet_amd.Mi355xUnetOption(info.model_name, provider))


def _list_optimizers(options):
    options.append(sd_opt_amd.SdOptimizationMi355x())


def _model_loaded(sd_model):
    """Boundary B4: route decode_first_stage through the engine (same hook style as modules/lowvram.py:65-75)."""
    vae_amd = importlib.import_module("stable-diffusion-webui_amd.sd_vae_hook")
    vae_amd.install(sd_model)


script_callbacks.on_list_unets(_list_unets)
script_callbacks.on_list_optimizers(_list_optimizers)
script_callbacks.on_model_loaded(_model_loaded)
