"""Webui-side registrations beyond the SdUnet / SdOptimization / VAE hooks (boundaries B3, B5, B6 of INTEGRATION.md), as functions
the shipped extension script calls with the webui's OWN modules as arguments — so a CPU test can drive them with stubs.

  B3  install_samplers      every row of ``modules.sd_samplers.all_samplers`` (modules/sd_samplers.py:11-16) whose name the engine's
                            sampler table carries gets a constructor that returns the engine sampler (fused CFG / step kernels) WHILE
                            the engine UNet is the active ``sd_unet.current_unet``, and the stock torch sampler otherwise: same names,
                            aliases and options, so scripts / API / infotext are untouched.
  B5  install_lora_hook     wraps ``networks.load_networks`` of extensions-builtin/Lora (called from ExtraNetworkLora.activate,
                            extra_networks_lora.py:18-45): the stock function keeps the text-encoder part and the bookkeeping, then the
                            UNet part of the same network files is merged on the GPU into the engine's packed weights.
  B6  install_clip_hook     points ``FrozenCLIPEmbedderWithCustomWords.encode_with_transformers`` (modules/sd_hijack_clip.py:351-360)
                            of the loaded model at the engine's CLIP tower; tokenisation, chunking, emphasis and textual inversion stay.

``EngineModelView`` is what those see as "the model": the webui's LatentDiffusion object supplies the schedule and the flags, the
active ``Mi355xUnet`` supplies the engine and the checkpoint weights (the LoRA "weights backup").
"""
from __future__ import annotations

import types
from typing import Optional


class EngineModelView:
    """The attribute set stable-diffusion-webui_amd's samplers / networks read from a model (sd_models.SdModel), served from the
    webui's ``shared.sd_model`` + the active engine UNet."""

    def __init__(self, sd_model, unet):
        import torch
        self._sd_model, self._unet = sd_model, unet
        self.engine = unet.engine
        self.device = torch.device("cuda", unet.engine.device)
        self.unet_cfg = unet.unet_cfg
        self.is_sdxl = bool(getattr(sd_model, "is_sdxl", False))
        self.is_sdxl_inpaint = bool(getattr(sd_model, "is_sdxl_inpaint", False))
        self.parameterization = getattr(sd_model, "parameterization", "eps")
        self.cond_stage_key = getattr(sd_model, "cond_stage_key", "txt")
        inner = getattr(sd_model, "model", None)
        self.model = types.SimpleNamespace(conditioning_key=getattr(inner, "conditioning_key", "crossattn"))
        self.scale_factor = getattr(sd_model, "scale_factor", 0.18215)

    @property
    def alphas_cumprod(self):                                 # read per job: the webui rewrites it for the schedule overrides
        return self._sd_model.alphas_cumprod

    @property
    def cond_stage_model_empty_prompt(self):
        return getattr(self._sd_model, "cond_stage_model_empty_prompt", None)

    def unet_checkpoint_tensor(self, engine_key: str):
        from . import schema
        return self._unet.checkpoint()[schema.UNET_PREFIX + engine_key]

    def decode_first_stage(self, z):
        return self._sd_model.decode_first_stage(z)

    def encode_first_stage(self, x):
        return self._sd_model.encode_first_stage(x)

    def get_first_stage_encoding(self, e):
        return self._sd_model.get_first_stage_encoding(e)


def active_engine_unet(sd_unet_module):
    """The engine UNet if it is the one the webui currently routes UNetModel.forward to (modules/sd_unet.py:86-93), else None."""
    from .sd_unet import Mi355xUnet
    cur = getattr(sd_unet_module, "current_unet", None)
    return cur if isinstance(cur, Mi355xUnet) and cur.engine is not None else None


_view_cache = {}


def engine_model_view(sd_model, sd_unet_module) -> Optional[EngineModelView]:
    unet = active_engine_unet(sd_unet_module)
    if unet is None:
        return None
    key = (id(sd_model), id(unet), id(unet.engine))
    view = _view_cache.get(key)
    if view is None:
        _view_cache.clear()
        view = _view_cache[key] = EngineModelView(sd_model, unet)
    return view


# ---- B3 ---------------------------------------------------------------------------------------------------------------------------
def install_samplers(webui_sd_samplers, sd_unet_module) -> list:
    """Idempotent (the webui re-imports extension scripts on "Reload UI").  Returns the names that now dispatch."""
    from . import sd_samplers as amd
    replaced = []
    rows = list(webui_sd_samplers.all_samplers)
    for i, row in enumerate(rows):
        mine = amd.all_samplers_map.get(row.name)
        if mine is None:
            continue
        stock = getattr(row.constructor, "_mi355x_stock", row.constructor)

        def constructor(model, stock=stock, mine=mine):
            view = engine_model_view(model, sd_unet_module)
            return stock(model) if view is None else mine.constructor(view)
        constructor._mi355x_stock = stock
        rows[i] = type(row)(row.name, constructor, row.aliases, row.options)
        replaced.append(row.name)
    webui_sd_samplers.all_samplers[:] = rows
    webui_sd_samplers.all_samplers_map.clear()
    webui_sd_samplers.all_samplers_map.update({x.name: x for x in rows})
    if hasattr(webui_sd_samplers, "set_samplers"):
        webui_sd_samplers.set_samplers()                      # rebuilds samplers / samplers_for_img2img / samplers_map from all_samplers
    return replaced


# ---- B5 ---------------------------------------------------------------------------------------------------------------------------
def install_lora_hook(webui_networks, webui_sd_models, webui_shared, sd_unet_module):
    """``webui_networks`` = the built-in Lora extension's ``networks`` module."""
    from . import networks as amd_networks
    stock = getattr(webui_networks.load_networks, "_mi355x_stock", webui_networks.load_networks)

    def load_networks(names, te_multipliers=None, unet_multipliers=None, dyn_dims=None):
        stock(names, te_multipliers, unet_multipliers, dyn_dims)          # text encoder + loaded_networks bookkeeping as before
        view = engine_model_view(webui_shared.sd_model, sd_unet_module)
        if view is None:
            return
        sds = []
        for n in names:
            on_disk = webui_networks.available_network_aliases.get(n) or webui_networks.available_networks.get(n)
            if on_disk is None:                               # the stock loader already reported it
                continue
            sds.append((n, webui_sd_models.read_state_dict(on_disk.filename)))
        idx = {n: i for i, n in enumerate(names)}
        pick = lambda xs: None if not xs else [xs[idx[n]] for n, _ in sds]
        amd_networks.load_networks(view, [n for n, _ in sds], [sd for _, sd in sds], pick(te_multipliers), pick(unet_multipliers), pick(dyn_dims))
    load_networks._mi355x_stock = stock
    webui_networks.load_networks = load_networks
    return load_networks


# ---- B6 ---------------------------------------------------------------------------------------------------------------------------
def install_clip_hook(sd_model, device_index: int = 0):
    """SD 1.x checkpoints (FrozenCLIPEmbedderWithCustomWords over transformers' CLIPTextModel): the CLIP-L tower is packed into an engine
    and ``encode_with_transformers`` of THIS model's embedder is rebound to it.  Token embeddings still come from the webui's
    (textual-inversion patched) embedding layer and enter as ``inputs_embeds``.  Returns the encoder, or None when the checkpoint's text
    encoder is not that class (SD 2.x / SDXL keep the torch towers: their hooks live in sd_hijack_clip.Mi355xClipTextEncoder and are
    bound the same way once the webui exposes the wrapped towers)."""
    from . import schema
    from .engine import Engine
    from .sd_hijack_clip import Mi355xClipTextEncoder
    csm = getattr(sd_model, "cond_stage_model", None)
    wrapped = getattr(csm, "wrapped", None)
    transformer = getattr(wrapped, "transformer", None)
    text_model = getattr(transformer, "text_model", None)
    if csm is None or text_model is None or getattr(sd_model, "is_sdxl", False):
        return None
    if hasattr(csm, "_mi355x_clip"):
        uninstall_clip_hook(sd_model)
    sd = {schema.CLIP_PREFIX + k: v for k, v in text_model.state_dict().items()}
    eng = Engine(device_index)
    enc = Mi355xClipTextEncoder(eng, schema.sd15_clip(), sd)
    csm._mi355x_clip = enc
    csm._torch_encode_with_transformers = csm.encode_with_transformers

    def encode_with_transformers(tokens):
        emb = text_model.embeddings.token_embedding(tokens)   # EmbeddingsWithFixes: textual-inversion vectors spliced in
        return enc.encode_with_transformers(tokens, inputs_embeds=emb)
    csm.encode_with_transformers = encode_with_transformers
    return enc


def uninstall_clip_hook(sd_model):
    csm = getattr(sd_model, "cond_stage_model", None)
    if csm is not None and hasattr(csm, "_torch_encode_with_transformers"):
        csm.encode_with_transformers = csm._torch_encode_with_transformers
        csm._mi355x_clip.engine.close()
        del csm._torch_encode_with_transformers, csm._mi355x_clip
