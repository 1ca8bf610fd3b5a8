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
    """Boundary B4: route decode_first_stage through the engine (same hook style as modules/lowvram.py:65-75); boundary B6: the
    text encoder's encode_with_transformers (modules/sd_hijack_clip.py:351-360) when shared.opts.mi355x_clip is set."""
    vae_amd = importlib.import_module("stable-diffusion-webui_amd.sd_vae_hook")
    vae_amd.install(sd_model)
    if getattr(getattr(shared, "opts", None), "mi355x_clip", False):
        bridge.install_clip_hook(sd_model)


bridge = importlib.import_module("stable-diffusion-webui_amd.webui_bridge")


def _register_samplers_and_lora():
    """Boundaries B3 / B5.  Imported lazily and tolerant of a webui without those modules (API-only forks, Lora extension disabled):
    the three callbacks above are the minimum, these two widen the drop-in to the fused CFG / sampler kernels and the GPU LoRA merge."""
    done = {}
    try:
        from modules import sd_samplers, sd_unet
        try:                                                                  # store_latent (live preview) and MaskBlendArgs (soft inpainting)
            from modules import sd_samplers_common, scripts as webui_scripts
        except ImportError:
            sd_samplers_common = webui_scripts = None
        bridge.bind_shared(shared, sd_samplers_common, webui_scripts)         # modules.shared.opts / state / cmd_opts reach the engine samplers
        done["samplers"] = bridge.install_samplers(sd_samplers, sd_unet, script_callbacks)   # modules/sd_samplers.py:11-16 rows, same names
        try:                                                                  # refiner checkpoint switch on the engine path (sd_samplers_common.py:158-202)
            from modules import devices as webui_devices
        except ImportError:
            webui_devices = None
        if hasattr(sd_models, "reload_model_weights"):
            bridge.install_refiner_switch(sd_models, shared, sd_unet, webui_devices)
            done["refiner"] = True
    except ImportError:
        sd_unet = None
    try:
        import networks as lora_networks                                      # extensions-builtin/Lora/networks.py (its dir is on sys.path)
        from modules import sd_unet
        bridge.install_lora_hook(lora_networks, sd_models, shared, sd_unet)   # wraps load_networks (extra_networks_lora.py:18-45)
        done["lora"] = True
    except ImportError:
        pass
    return done


script_callbacks.on_list_unets(_list_unets)
script_callbacks.on_list_optimizers(_list_optimizers)
script_callbacks.on_model_loaded(_model_loaded)
registered = _register_samplers_and_lora()
