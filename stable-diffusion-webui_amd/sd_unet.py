"""Boundary B1: the engine as an ``SdUnet`` (modules/sd_unet.py:63-93).

Inside the webui, ``Mi355xUnetOption`` is appended by an ``on_list_unets`` callback
(modules/script_callbacks.py:602-606); ``apply_unet`` (modules/sd_unet.py:33-60) then calls ``create_unet()`` /
``activate()`` and the patched ``UNetModel.forward`` (modules/sd_unet.py:86-93) routes every UNet call to
``Mi355xUnet.forward(x, timesteps, context, *args, **kwargs)``.  Standalone, the same classes derive from local stand-ins.
"""
from __future__ import annotations

import torch

try:                                            # inside the webui: subclass the real plugin base classes
    from modules import sd_unet as _ref_sd_unet
    SdUnetOption, SdUnet = _ref_sd_unet.SdUnetOption, _ref_sd_unet.SdUnet
except Exception:                               # standalone: same interface (modules/sd_unet.py:63-83)
    class SdUnetOption:
        model_name = None
        label = None

        def create_unet(self):
            raise NotImplementedError()

    class SdUnet(torch.nn.Module):
        def forward(self, x, timesteps, context, *args, **kwargs):
            raise NotImplementedError()

        def activate(self):
            pass

        def deactivate(self):
            pass


class Mi355xUnet(SdUnet):
    """forward(x, timesteps, context, y=None): x [2B,C,h,w] in dtype_unet already scaled by c_in, timesteps [2B],
    context [2B,77k,ctx_dim]; returns eps [2B,4,h,w] in x.dtype on x.device (modules/sd_hijack_unet.py:40-54 contract)."""

    def __init__(self, state_dict_provider, unet_cfg=None, device_index: int = 0):
        super().__init__()
        self._provider = state_dict_provider
        self._cfg = unet_cfg
        self._device_index = device_index
        self.engine = None
        self.unet_cfg = unet_cfg
        self._sd = None
        self._ctx_key = None
        self._last_ctx = None

    def checkpoint(self) -> dict:
        """The state dict the engine was packed from (kept by reference while active: the "weights backup" a LoRA rewrite starts
        from, extensions-builtin/Lora/networks.py:423-432)."""
        return self._sd

    def activate(self):
        from . import schema
        from .engine import Engine
        sd = self._provider()
        cfg = self._cfg
        if cfg is None:
            from .sd_models import guess_unet_config
            cfg = guess_unet_config(sd)
        self.engine = Engine(self._device_index)
        self.engine.load_unet(cfg, sd, prefix=schema.UNET_PREFIX)
        self.unet_cfg, self._sd = cfg, sd

    def deactivate(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None
        self._sd = None
        self._ctx_key = self._last_ctx = None

    def forward(self, x, timesteps, context, *args, **kwargs):
        from . import shared
        if shared.webui is not None:                          # ToMe / Hypertile patch the torch UNet this adapter replaces: never ignored —
            from .webui_bridge import patched_unet_reason     # the call goes to the patched torch UNet
            why = patched_unet_reason(getattr(shared.webui, "sd_model", None))
            if why is not None:
                return self._torch_unet_forward(why, x, timesteps, context, *args, **kwargs)
        # extra inputs (modules/sd_unet.py:76-77, 87-91 pass *args / **kwargs through): ``y`` and the ControlNet residuals of ldm's
        # ControlledUnetModel.forward (``control`` / ``only_mid_control``, cldm.py) are the engine's; anything else is refused loudly
        extra = set(kwargs) - {"y", "control", "only_mid_control"}
        if args or extra:
            raise NotImplementedError(f"extra UNet inputs {sorted(extra) or 'given positionally'} are not supported by the engine UNet")
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        y = kwargs.get("y", None)
        # The context is step-invariant unless prompt editing swaps it, so its K / V projections are cached — validated by
        # CONTENT, on the DEVICE: the webui re-catenates cond | uncond every step (sd_samplers_cfg_denoiser.py:246) and the caching
        # allocator hands the next step's (or the next job's) tensor the same address, so an address / version key would go stale,
        # and a host-side torch.equal would cost a device -> host synchronisation per UNet evaluation.  The engine compares the
        # new rows with its cached fp16 copy in a kernel and predicates the re-projection launches on the result.
        self.engine.set_context_cached(context)
        ctx = None
        # This adapter gets an anonymous batch from the webui's own CFG denoiser: whether it is [x | x] at one timestep (the plain CFG
        # batch, for which the engine shares the layers in front of the first cross-attention) only the data says.  opts.mi355x_auto_cfg_pairs
        # lets the engine look (a synchronising compare per evaluation; the engine's own samplers know and never need it).
        auto = bool(getattr(shared.opts, "mi355x_auto_cfg_pairs", True))
        return self.engine.unet_forward(x, timesteps, ctx, y, auto_promises=auto, control=kwargs.get("control"),
                                        only_mid_control=bool(kwargs.get("only_mid_control", False)))


    def _torch_unet_forward(self, why, x, timesteps, context, *args, **kwargs):
        """This one call through the webui's OWN UNet, with whatever patched it (tomesd around attn1, Hypertile's tiled self-attention)
        in force.  modules/sd_unet.py:86-93 routes UNetModel.forward to ``current_unet`` while one is set, so the option steps aside for
        the duration of the call; apply_unet parked the torch UNet on the CPU when the option was activated (:55), so it is brought back
        to the device first (and stays: 288 GB of HBM hold both).  The mi355x cross-attention optimization keeps working inside it."""
        from . import shared, webui_bridge
        ref = webui_bridge.webui_sd_unet_module()
        sd_model = getattr(shared.webui, "sd_model", None)
        unet = getattr(getattr(sd_model, "model", None), "diffusion_model", None)
        if ref is None or unet is None:
            raise NotImplementedError(webui_bridge.REFUSAL.format(why=why))
        first = next(unet.parameters(), None)
        if first is not None and first.device != x.device:
            unet.to(x.device)
        self.torch_fallback_reason = why                      # what tests and the infotext hook can read back
        saved = ref.current_unet
        ref.current_unet = None
        try:
            return unet(x, timesteps, context, *args, **kwargs)
        finally:
            ref.current_unet = saved


class Mi355xUnetOption(SdUnetOption):
    def __init__(self, model_name, state_dict_provider, unet_cfg=None, device_index=0):
        self.model_name = model_name
        self.label = f"[MI355X] {model_name}"
        self._provider = state_dict_provider
        self._cfg = unet_cfg
        self._device_index = device_index

    def create_unet(self):
        return Mi355xUnet(self._provider, self._cfg, self._device_index)
