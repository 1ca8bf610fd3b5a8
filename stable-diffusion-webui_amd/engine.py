"""Python face of the C++ engine (sdmi_engine): weight hand-over from a checkpoint state dict, UNet forward,
VAE decode / encode.  Torch tensors are used for storage only; every computation is a HIP kernel behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr, dtype_code
from .schema import UNET_PREFIX, VAE_PREFIX, UNetConfig, VAEConfig, unet_schema, vae_schema
import os

# SDMI_CFG_PAIRS=0 keeps the CFG denoiser's [cond | uncond] batch on the per-row path everywhere (same-box A/B, tools/gpu/knob_sweep.py)
CFG_PAIRS = os.environ.get("SDMI_CFG_PAIRS", "1") != "0"


def _unet_cfg_c(cfg: UNetConfig) -> _lib.UNetConfigC:
    c = _lib.UNetConfigC()
    c.in_channels, c.out_channels, c.model_channels = cfg.in_channels, cfg.out_channels, cfg.model_channels
    c.num_levels = len(cfg.channel_mult)
    ds = 1
    for i, m in enumerate(cfg.channel_mult):
        c.channel_mult[i] = m
        c.attn_level[i] = 1 if ds in cfg.attention_resolutions else 0
        c.transformer_depth[i] = cfg.depth_at(i)
        ds *= 2
    c.num_res_blocks = cfg.num_res_blocks
    c.num_heads = cfg.num_heads
    c.num_head_channels = cfg.num_head_channels
    c.context_dim = cfg.context_dim
    c.adm_in_channels = cfg.adm_in_channels or 0
    return c


def _vae_cfg_c(cfg: VAEConfig) -> _lib.VAEConfigC:
    c = _lib.VAEConfigC()
    c.ch, c.num_levels = cfg.ch, len(cfg.ch_mult)
    for i, m in enumerate(cfg.ch_mult):
        c.ch_mult[i] = m
    c.num_res_blocks, c.in_channels, c.out_ch, c.z_channels = cfg.num_res_blocks, cfg.in_channels, cfg.out_ch, cfg.z_channels
    c.scale_factor = cfg.scale_factor
    return c


class Engine:
    """One engine per GPU.  Not thread-safe by design: the webui serialises GPU work behind one FIFO lock."""

    def __init__(self, device: int = 0):
        _lib.require_device()
        self.device = int(device)
        self.handle = lib.sdmi_engine_create(self.device)
        if not self.handle:
            raise _lib.SdmiError("sdmi_engine_create failed: " + _lib.last_error())
        self.unet_cfg: Optional[UNetConfig] = None
        self.vae_cfg: Optional[VAEConfig] = None
        self._ctx_key = None

    def close(self):
        if getattr(self, "handle", None):
            lib.sdmi_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------
    def set_option(self, name: str, value: int):
        check(lib.sdmi_engine_set_option(self.handle, name.encode(), int(value)), "set_option")

    def _load(self, fn, key: str, t: torch.Tensor):
        if t.dtype not in (torch.float16, torch.float32):
            t = t.float()
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        check(fn(self.handle, key.encode(), C.c_void_p(t.data_ptr()), dtype_code(t), t.dim(), shape, 1 if t.is_cuda else 0),
              f"load_tensor({key})")

    def load_unet(self, cfg: UNetConfig, state_dict: dict, prefix: str = UNET_PREFIX):
        """Stream ``state_dict[prefix + key]`` for every key of the UNet schema into the engine and pack.
        ``state_dict`` is what modules/sd_models.py:312-329 read_state_dict returns."""
        self.unet_cfg = cfg
        c = _unet_cfg_c(cfg)
        check(lib.sdmi_unet_configure(self.handle, C.byref(c)), "unet_configure")
        for key, shape, _ in unet_schema(cfg):
            t = state_dict[prefix + key]
            if tuple(t.shape) != tuple(shape):
                raise _lib.SdmiError(f"checkpoint tensor {prefix + key} has shape {tuple(t.shape)}, expected {shape}")
            self._load(lib.sdmi_unet_load_tensor, key, t)
        check(lib.sdmi_unet_finalize(self.handle), "unet_finalize")
        self._ctx_key = None

    def update_unet_weight(self, key: str, t: torch.Tensor):
        """Replace one conv / linear weight of the loaded UNet (LoRA rewrite, networks.network_apply_weights) and drop the
        cached cross-attention projections."""
        self._load(lib.sdmi_unet_update_weight, key, t)
        self._ctx_key = None
        self.weights_version = getattr(self, "weights_version", 0) + 1

    def update_unet_vector(self, key: str, t: torch.Tensor):
        """Replace one bias / norm gain / norm shift of the loaded UNet (LyCORIS norm modules, bias deltas)."""
        if t.dtype not in (torch.float16, torch.float32):
            t = t.float()
        t = t.contiguous().reshape(-1)
        check(lib.sdmi_unet_update_vector(self.handle, key.encode(), C.c_void_p(t.data_ptr()), dtype_code(t), t.numel(), 1 if t.is_cuda else 0),
              f"update_vector({key})")
        self.weights_version = getattr(self, "weights_version", 0) + 1

    def load_vae(self, cfg: VAEConfig, state_dict: dict, prefix: str = VAE_PREFIX, decoder_only: bool = False):
        self.vae_cfg = cfg
        c = _vae_cfg_c(cfg)
        check(lib.sdmi_vae_configure(self.handle, C.byref(c)), "vae_configure")
        for key, shape, _ in vae_schema(cfg):
            if decoder_only and (key.startswith("encoder.") or key.startswith("quant_conv.")):
                continue
            self._load(lib.sdmi_vae_load_tensor, key, state_dict[prefix + key])
        check(lib.sdmi_vae_finalize(self.handle), "vae_finalize")

    def load_clip(self, cfg, state_dict: dict, prefix: str = None, slot: int = 0):
        """Stream a transformers-layout CLIP text model (keys below ``prefix`` = ".text_model.") into the engine."""
        from .schema import CLIP_PREFIX, clip_schema
        prefix = CLIP_PREFIX if prefix is None else prefix
        c = _lib.ClipConfigC(cfg.vocab_size, cfg.max_positions, cfg.hidden, cfg.layers, cfg.heads, cfg.intermediate,
                             {"quick_gelu": 0, "gelu": 1}[cfg.act], cfg.eps)
        check(lib.sdmi_clip_configure(self.handle, slot, C.byref(c)), "clip_configure")
        for key, shape, _ in clip_schema(cfg):
            t = state_dict[prefix + key]
            if tuple(t.shape) != tuple(shape):
                raise _lib.SdmiError(f"checkpoint tensor {prefix + key} has shape {tuple(t.shape)}, expected {shape}")
            if t.dtype not in (torch.float16, torch.float32):
                t = t.float()
            t = t.contiguous()
            shp = (C.c_int64 * t.dim())(*t.shape)
            check(lib.sdmi_clip_load_tensor(self.handle, slot, key.encode(), C.c_void_p(t.data_ptr()), dtype_code(t), t.dim(), shp,
                                            1 if t.is_cuda else 0), f"clip_load_tensor({key})")
        check(lib.sdmi_clip_finalize(self.handle, slot), "clip_finalize")
        self.clip_cfg = getattr(self, "clip_cfg", {})
        self.clip_cfg[slot] = cfg

    def clip_forward(self, tokens: torch.Tensor, skip: int = 1, apply_final_ln: bool = True, inputs_embeds: torch.Tensor = None,
                     slot: int = 0, return_pooled: bool = False):
        """tokens [B, L] integer tensor on the engine's device -> hidden states [B, L, hidden] fp32 (see sdmi_clip_forward)."""
        dev = torch.device("cuda", self.device)
        tok = tokens.to(dev, torch.int32).contiguous()
        b, l = tok.shape
        hidden = self.clip_cfg[slot].hidden
        emb = None
        if inputs_embeds is not None:
            emb = inputs_embeds.to(dev, torch.float32).contiguous()
            assert emb.shape == (b, l, hidden)
        out = torch.empty((b, l, hidden), dtype=torch.float32, device=dev)
        pdim = self.clip_cfg[slot].proj_dim or hidden
        pooled = torch.empty((b, pdim), dtype=torch.float32, device=dev) if return_pooled else None
        check(lib.sdmi_clip_forward(self.handle, slot, ptr(tok), ptr(emb), b, l, int(skip), 1 if apply_final_ln else 0, ptr(out),
                                    ptr(pooled), stream_ptr()), "clip_forward")
        return (out, pooled) if return_pooled else out

    # ------------------------------------------------------------------------------------------------------
    def set_context(self, context: torch.Tensor):
        """Project the (step-invariant) text conditioning to every cross-attention layer's K / V^T once."""
        context = context.contiguous()
        bn, l, _ = context.shape
        check(lib.sdmi_unet_set_context(self.handle, ptr(context), dtype_code(context), bn, l, stream_ptr()), "set_context")
        self._ctx_shape = (bn, l)

    def set_context_cached(self, context: torch.Tensor):
        """set_context for callers that cannot know whether the context changed (SdUnet.forward inside the webui): compared with
        the cached copy on the device, re-projected only if different — no host synchronisation (sdmi_unet_set_context_cached)."""
        if context.dtype not in (torch.float16, torch.float32):
            context = context.float()
        context = context.contiguous()
        bn, l, _ = context.shape
        check(lib.sdmi_unet_set_context_cached(self.handle, ptr(context), dtype_code(context), bn, l, stream_ptr()), "set_context_cached")
        self._ctx_shape = (bn, l)

    def unet_forward(self, x: torch.Tensor, timesteps: torch.Tensor, context: Optional[torch.Tensor] = None,
                     y: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, uniform_t: bool = False,
                     cfg_pairs: bool = False, auto_promises: bool = False, control=None, only_mid_control: bool = False) -> torch.Tensor:
        """eps = UNet(x, timesteps, context[, y]); x [Bn,Cin,h,w]; context None reuses the cached projections.  ``uniform_t``: the
        caller guarantees that all rows share one timestep (the samplers' CFG batch): the timestep-embedding path then runs for one row
        (engine option "uniform_t"; same bits)."""
        # ``cfg_pairs``: the caller guarantees rows [Bn/2, Bn) repeat the latent AND the timestep of rows [0, Bn/2) (the CFG denoiser's
        # [cond | uncond] batch): the layers in front of the first cross-attention then run for one half.
        # ``auto_promises``: the caller knows neither — the engine derives both from x and timesteps for this call (one synchronising
        # device -> host compare per forward); for the stock CFG denoiser behind Mi355xUnet.forward.
        # All three are arguments of THIS call (sdmi_unet_forward_ex, round 6): nothing sticky is left in the engine for a later caller.
        flags = (1 if uniform_t else 0) | (2 if (cfg_pairs and CFG_PAIRS) else 0) | (4 if (auto_promises and CFG_PAIRS) else 0)
        x = x.contiguous()
        dt = x.dtype
        timesteps = timesteps.to(dt).contiguous()
        bn, _, h, w = x.shape
        if context is not None:
            context = context.to(dt).contiguous()
            l = context.shape[1]
            self._ctx_shape = (bn, l)
        else:
            l = self._ctx_shape[1]
        if y is not None:
            y = y.to(dt).contiguous()
        if out is None:
            out = torch.empty((bn, self.unet_cfg.out_channels, h, w), dtype=dt, device=x.device)
        if control is not None:
            # ControlNet residuals (ldm cldm.py ControlledUnetModel.forward): one tensor per input block output + the middle block's,
            # handed to the engine for THIS call (sdmi_unet_set_control); kept alive here until the launch is enqueued
            control = [c.to(device=x.device, dtype=dt).contiguous() for c in control]
            ptrs = (C.c_void_p * len(control))(*[c.data_ptr() for c in control])
            numel = (C.c_int64 * len(control))(*[c.numel() for c in control])
            check(lib.sdmi_unet_set_control(self.handle, ptrs, numel, len(control), 1 if only_mid_control else 0), "unet_set_control")
        check(lib.sdmi_unet_forward_ex(self.handle, ptr(x), ptr(timesteps), ptr(context), ptr(y), ptr(out), dtype_code(x),
                                       bn, h, w, l, flags, stream_ptr()), "unet_forward")
        return out

    def vae_decode(self, z: torch.Tensor) -> torch.Tensor:
        """decode_first_stage for a whole batch: fp32 NCHW image in [-1, 1]."""
        z = z.contiguous()
        b, _, h, w = z.shape
        nlev = len(self.vae_cfg.ch_mult)
        f = 2 ** (nlev - 1)
        out = torch.empty((b, self.vae_cfg.out_ch, h * f, w * f), dtype=torch.float32, device=z.device)
        check(lib.sdmi_vae_decode(self.handle, ptr(z), dtype_code(z), ptr(out), b, h, w, stream_ptr()), "vae_decode")
        return out

    def vae_encode_moments(self, x: torch.Tensor) -> torch.Tensor:
        """quant_conv(encoder(x)): fp32 NCHW [B, 2*z, H/f, W/f] (mean | logvar); x in [-1, 1]."""
        x = x.contiguous()
        b, _, hh, ww = x.shape
        nlev = len(self.vae_cfg.ch_mult)
        f = 2 ** (nlev - 1)
        out = torch.empty((b, 2 * self.vae_cfg.z_channels, hh // f, ww // f), dtype=torch.float32, device=x.device)
        check(lib.sdmi_vae_encode(self.handle, ptr(x), dtype_code(x), ptr(out), b, hh, ww, stream_ptr()), "vae_encode")
        return out

    def taps(self) -> dict:
        """Block outputs of the last UNet forward / VAE decode recorded under option "trace": {reference module name: fp16 NCHW
        tensor} (parity error budget; see sdmi_engine_tap_*)."""
        out = {}
        dev = torch.device("cuda", self.device)
        for i in range(int(lib.sdmi_engine_tap_count(self.handle))):
            name = C.create_string_buffer(256)
            dims = (C.c_int64 * 4)()
            check(lib.sdmi_engine_tap_info(self.handle, i, name, 256, dims), "tap_info")
            b, h, w, c = (int(d) for d in dims)
            t = torch.empty((b, h, w, c), dtype=torch.float16, device=dev)
            check(lib.sdmi_engine_tap_read(self.handle, i, ptr(t), stream_ptr()), "tap_read")
            out[name.value.decode()] = t.permute(0, 3, 1, 2)
        return out

    def arena_bytes(self) -> int:
        return int(lib.sdmi_engine_arena_bytes(self.handle))
