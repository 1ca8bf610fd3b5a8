"""fp32 CPU restatement of the eps-prediction UNet on the reference hot path (oracle; tests only).

The reference does not own this arithmetic: it instantiates ``ldm.modules.diffusionmodules.
openaimodel.UNetModel`` (Stability-AI/stablediffusion @ cf1d67a6) or the ``sgm`` twin
(generative-models @ 45c443b3) from ``configs/v1-inference.yaml:29-44`` /
``configs/sd_xl_inpaint.yaml:19-37`` and monkey-patches pieces of it.  What is pinned in-tree
and followed here:
  * module / state-dict layout .......... extensions-builtin/Lora/networks.py:43-98
  * timestep embedding (cos first, fp32)  modules/sd_hijack_unet.py:58-78
  * SpatialTransformer.forward order .... modules/sd_hijack_unet.py:83-102
  * BasicTransformerBlock._forward ...... modules/sd_hijack_checkpoint.py:7-8
  * baseline CrossAttention.forward ..... modules/hypernetworks/hypernetwork.py:382-407
  * GroupNorm32 = fp32 GroupNorm(32) .... modules/devices.py:284-295
  * SDXL block indices / depths ......... modules/sd_hijack.py:191-203
  * structural checksum: 859,520,964 params (SD1.5), 2,567,463,684 (SDXL base).
Parity status for this file: unpinned numerically (no golden vector exists in the reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    channel_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    attention_resolutions: Sequence[int] = (4, 2, 1)
    num_heads: int = 8                      # used when num_head_channels == -1 (SD1.x)
    num_head_channels: int = -1             # SDXL / SD2: 64
    transformer_depth: Union[int, Sequence[int]] = 1
    context_dim: int = 768
    use_linear_in_transformer: bool = False
    adm_in_channels: Optional[int] = None   # SDXL: 2816

    def depth_at(self, level: int) -> int:
        td = self.transformer_depth
        return td if isinstance(td, int) else td[level]


def sd15_config() -> UNetConfig:
    """configs/v1-inference.yaml:29-44"""
    return UNetConfig()


def sdxl_base_config() -> UNetConfig:
    """configs/sd_xl_inpaint.yaml:19-37 with in_channels 4 (base model)."""
    return UNetConfig(model_channels=320, channel_mult=(1, 2, 4), attention_resolutions=(4, 2),
                      num_heads=-1, num_head_channels=64, transformer_depth=(1, 2, 10),
                      context_dim=2048, use_linear_in_transformer=True, adm_in_channels=2816)


def tiny_config(**kw) -> UNetConfig:
    """Small SD1.x-shaped config for fast tests (channels stay multiples of 64 so the MFMA path is used)."""
    base = dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(1, 2),
                num_heads=-1, num_head_channels=64, transformer_depth=1, context_dim=64)
    base.update(kw)
    return UNetConfig(**base)


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """modules/sd_hijack_unet.py:58-78 (cos first, then sin; computed in fp32)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class ResBlock(nn.Module):
    """ldm ResBlock; names per extensions-builtin/Lora/networks.py:43-53."""

    def __init__(self, ch, emb_ch, out_ch):
        super().__init__()
        self.in_layers = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(ch, out_ch, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, out_ch))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_ch), nn.SiLU(), nn.Dropout(0.0),
                                        nn.Conv2d(out_ch, out_ch, 3, padding=1))
        self.skip_connection = nn.Identity() if out_ch == ch else nn.Conv2d(ch, out_ch, 1)

    def forward(self, x, emb):
        h = self.in_layers(x)
        h = h + self.emb_layers(emb)[:, :, None, None]
        h = self.out_layers(h)
        return self.skip_connection(x) + h


LOADED_HYPERNETWORKS: list = []       # oracle.hypernetwork.Hypernetwork objects (shared.loaded_hypernetworks of the reference)
# Rows of the score matrix evaluated at a time (None = all at once).  Softmax rows are independent, so the result is the one of the
# unchunked product; the full-size fixtures (N = 16384 tokens at a 128x128 latent: 17 GB of fp32 scores per forward) are generated
# with a bound — the same memory bound modules/sub_quadratic_attention.py:54-113 puts on the reference's own product.
QUERY_CHUNK = None


class CrossAttention(nn.Module):
    """modules/hypernetworks/hypernetwork.py:382-407 (baseline forward incl. apply_hypernetworks on the context; no mask)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))

    def forward(self, x, context=None):
        h = self.heads
        context = x if context is None else context
        context_k = context_v = context
        if LOADED_HYPERNETWORKS:
            from .hypernetwork import apply_hypernetworks
            context_k, context_v = apply_hypernetworks(LOADED_HYPERNETWORKS, context)
        q, k, v = self.to_q(x), self.to_k(context_k), self.to_v(context_v)
        b, n, _ = q.shape
        split = lambda t: t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)
        q, k, v = split(q), split(k), split(v)
        if QUERY_CHUNK and n > QUERY_CHUNK:
            out = torch.cat([torch.einsum('bij,bjd->bid', (torch.einsum('bid,bjd->bij', q[:, i:i + QUERY_CHUNK], k) * self.scale).softmax(dim=-1), v)
                             for i in range(0, n, QUERY_CHUNK)], dim=1)
        else:
            sim = torch.einsum('bid,bjd->bij', q, k) * self.scale
            attn = sim.softmax(dim=-1)
            out = torch.einsum('bij,bjd->bid', attn, v)
        out = out.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.to_out(out)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        x = self.ff(self.norm3(x)) + x
        return x


class SpatialTransformer(nn.Module):
    """forward order: modules/sd_hijack_unet.py:83-102."""

    def __init__(self, in_ch, n_heads, d_head, depth, context_dim, use_linear):
        super().__init__()
        inner = n_heads * d_head
        self.use_linear = use_linear
        self.norm = nn.GroupNorm(32, in_ch, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_ch, inner) if use_linear else nn.Conv2d(in_ch, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(inner, in_ch) if use_linear else nn.Conv2d(inner, in_ch, 1)

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        x_in = x
        x = self.norm(x)
        if not self.use_linear:
            x = self.proj_in(x)
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        if self.use_linear:
            x = self.proj_in(x)
        for block in self.transformer_blocks:
            x = block(x, context=context)
        if self.use_linear:
            x = self.proj_out(x)
        x = x.view(b, h, w, -1).permute(0, 3, 1, 2)
        if not self.use_linear:
            x = self.proj_out(x)
        return x + x_in


class Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.op = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))


class TimestepEmbedSequential(nn.Sequential):
    def forward(self, x, emb, context=None):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    """State-dict keys equal the ``model.diffusion_model.*`` namespace probed at modules/sd_models.py:392."""

    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        mc = cfg.model_channels
        ted = mc * 4
        self.time_embed = nn.Sequential(nn.Linear(mc, ted), nn.SiLU(), nn.Linear(ted, ted))
        if cfg.adm_in_channels is not None:
            self.label_emb = nn.Sequential(
                nn.Sequential(nn.Linear(cfg.adm_in_channels, ted), nn.SiLU(), nn.Linear(ted, ted)))

        def heads_for(ch):
            if cfg.num_head_channels == -1:
                return cfg.num_heads, ch // cfg.num_heads
            return ch // cfg.num_head_channels, cfg.num_head_channels

        def make_st(ch, level):
            nh, dh = heads_for(ch)
            return SpatialTransformer(ch, nh, dh, cfg.depth_at(level), cfg.context_dim,
                                      cfg.use_linear_in_transformer)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(cfg.in_channels, mc, 3, padding=1))])
        chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(cfg.channel_mult):
            for _ in range(cfg.num_res_blocks):
                layers = [ResBlock(ch, ted, mult * mc)]
                ch = mult * mc
                if ds in cfg.attention_resolutions:
                    layers.append(make_st(ch, level))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(cfg.channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        last = len(cfg.channel_mult) - 1
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, ch), make_st(ch, last), ResBlock(ch, ted, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
            for i in range(cfg.num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, mc * mult)]
                ch = mc * mult
                if ds in cfg.attention_resolutions:
                    layers.append(make_st(ch, level))
                if level and i == cfg.num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(mc, cfg.out_channels, 3, padding=1))

    def forward(self, x, timesteps, context, y=None, control=None, only_mid_control=False):
        """``control`` (ControlNet residuals; the webui's SdUnet.forward hands extra inputs through, modules/sd_unet.py:76-77, 87-91): the
        published ControlledUnetModel.forward (lllyasviel/ControlNet cldm/cldm.py — third party, absent here: restated) — a list with one
        tensor per input block output plus one for the middle block, consumed from the END: the last entry is added to the middle block's
        output, the others to the skip connections as the output blocks pop them."""
        emb = self.time_embed(timestep_embedding(timesteps, self.cfg.model_channels))
        if self.cfg.adm_in_channels is not None:
            emb = emb + self.label_emb(y)
        hs = []
        h = x
        for module in self.input_blocks:
            h = module(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        control = list(control) if control is not None else None
        if control is not None:
            h = h + control.pop()
        for module in self.output_blocks:
            if only_mid_control or control is None:
                h = torch.cat([h, hs.pop()], dim=1)
            else:
                h = torch.cat([h, hs.pop() + control.pop()], dim=1)
            h = module(h, emb, context)
        return self.out(h)


def build_unet(cfg: UNetConfig, state_dict: dict, prefix: str = "model.diffusion_model.") -> UNetModel:
    """Instantiate the oracle UNet and load fp32 copies of ``state_dict[prefix + key]``."""
    with torch.device("meta"):          # skip the (slow) random init; parameters are assigned from the checkpoint
        net = UNetModel(cfg)
    own = {}
    for k in net.state_dict().keys():
        own[k] = state_dict[prefix + k].float()
    net.load_state_dict(own, strict=True, assign=True)
    return net.eval().requires_grad_(False)
