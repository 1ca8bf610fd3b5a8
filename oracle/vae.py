"""fp32 CPU restatement of AutoencoderKL decode / encode (oracle; tests only).

The reference calls ``ldm.models.autoencoder.AutoencoderKL`` (third-party, not vendored) through
``modules/sd_samplers_common.py:55-58`` (decode) and ``:87-112`` (encode).  Structure follows the
in-tree plain-torch twin ``modules/models/sd3/sd3_impls.py:171-355`` (ResnetBlock, AttnBlock,
Downsample with (0,1,0,1) padding, Upsample nearest x2, VAEEncoder, VAEDecoder) with
``z_channels=4`` per ``configs/v1-inference.yaml:51-65``, plus ``quant_conv`` / ``post_quant_conv``
and the latent ``scale_factor`` 0.18215 (``configs/v1-inference.yaml:17``; division shown in-tree at
``modules/models/diffusion/ddpm_edit.py:734``).  State-dict keys equal ``first_stage_model.*``.
Pinned by tests/test_oracle_pins.py against fixtures produced by the reference's own VAEDecoder /
VAEEncoder classes (tests/golden/make_golden.py).  Checksums: decoder 49,490,179 params, whole
AutoencoderKL 83,653,863.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    ch: int = 128
    ch_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    in_channels: int = 3
    out_ch: int = 3
    z_channels: int = 4
    scale_factor: float = 0.18215


def sd15_vae_config() -> VAEConfig:
    return VAEConfig()


def tiny_vae_config(**kw) -> VAEConfig:
    base = dict(ch=64, ch_mult=(1, 2), num_res_blocks=1)
    base.update(kw)
    return VAEConfig(**base)


QUERY_CHUNK = None                     # see oracle/unet.py


def Normalize(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)   # sd3_impls.py:171-172


class ResnetBlock(nn.Module):                            # sd3_impls.py:175-202
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = Normalize(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = Normalize(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)
        self.cin, self.cout = cin, cout

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.cin != self.cout:
            x = self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):                              # sd3_impls.py:205-224; scale c^-0.5
    def __init__(self, c):
        super().__init__()
        self.norm = Normalize(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)

    def forward(self, x):
        h = self.norm(x)
        q, k, v = self.q(h), self.k(h), self.v(h)
        b, c, hh, ww = q.shape
        q, k, v = [t.reshape(b, c, hh * ww).permute(0, 2, 1) for t in (q, k, v)]
        n = q.shape[1]
        if QUERY_CHUNK and n > QUERY_CHUNK:                  # row blocks of the same product (softmax rows are independent)
            o = torch.cat([torch.bmm(torch.softmax(torch.bmm(q[:, i:i + QUERY_CHUNK], k.transpose(1, 2)) * (int(c) ** -0.5), dim=-1), v)
                           for i in range(0, n, QUERY_CHUNK)], dim=1)
        else:
            w = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (int(c) ** -0.5), dim=-1)
            o = torch.bmm(w, v)
        o = o.permute(0, 2, 1).reshape(b, c, hh, ww)
        return x + self.proj_out(o)


class Downsample(nn.Module):                             # sd3_impls.py:227-236
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Upsample(nn.Module):                               # sd3_impls.py:239-247
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Encoder(nn.Module):                                # sd3_impls.py:250-302
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch, nres = cfg.ch, len(cfg.ch_mult)
        self.nres, self.nrb = nres, cfg.num_res_blocks
        self.conv_in = nn.Conv2d(cfg.in_channels, ch, 3, padding=1)
        in_mult = (1,) + tuple(cfg.ch_mult)
        self.down = nn.ModuleList()
        for i in range(nres):
            bi, bo = ch * in_mult[i], ch * cfg.ch_mult[i]
            lvl = nn.Module()
            lvl.block = nn.ModuleList()
            lvl.attn = nn.ModuleList()
            for _ in range(cfg.num_res_blocks):
                lvl.block.append(ResnetBlock(bi, bo))
                bi = bo
            if i != nres - 1:
                lvl.downsample = Downsample(bi)
            self.down.append(lvl)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(bi, bi)
        self.mid.attn_1 = AttnBlock(bi)
        self.mid.block_2 = ResnetBlock(bi, bi)
        self.norm_out = Normalize(bi)
        self.conv_out = nn.Conv2d(bi, 2 * cfg.z_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for i in range(self.nres):
            for j in range(self.nrb):
                h = self.down[i].block[j](h)
            if i != self.nres - 1:
                h = self.down[i].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(F.silu(self.norm_out(h)))


class Decoder(nn.Module):                                # sd3_impls.py:305-355
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch, nres = cfg.ch, len(cfg.ch_mult)
        self.nres, self.nrb = nres, cfg.num_res_blocks
        bi = ch * cfg.ch_mult[nres - 1]
        self.conv_in = nn.Conv2d(cfg.z_channels, bi, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(bi, bi)
        self.mid.attn_1 = AttnBlock(bi)
        self.mid.block_2 = ResnetBlock(bi, bi)
        self.up = nn.ModuleList()
        for i in reversed(range(nres)):
            bo = ch * cfg.ch_mult[i]
            lvl = nn.Module()
            lvl.block = nn.ModuleList()
            for _ in range(cfg.num_res_blocks + 1):
                lvl.block.append(ResnetBlock(bi, bo))
                bi = bo
            if i != 0:
                lvl.upsample = Upsample(bi)
            self.up.insert(0, lvl)
        self.norm_out = Normalize(bi)
        self.conv_out = nn.Conv2d(bi, cfg.out_ch, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i in reversed(range(self.nres)):
            for j in range(self.nrb + 1):
                h = self.up[i].block[j](h)
            if i != 0:
                h = self.up[i].upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.z_channels, 2 * cfg.z_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.z_channels, cfg.z_channels, 1)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def decode_first_stage(self, z):
        """LatentDiffusion.decode_first_stage: z / scale_factor then decode (ddpm_edit.py:726-735)."""
        return self.decode(z / self.cfg.scale_factor)

    def encode_moments(self, x):
        return self.quant_conv(self.encoder(x))

    def encode_first_stage_mean(self, x):
        """encode + get_first_stage_encoding with the posterior *mean* (deterministic variant used by the
        measurement plan); the reference samples (sd3_impls.py:369-374)."""
        mean, _ = torch.chunk(self.encode_moments(x), 2, dim=1)
        return mean * self.cfg.scale_factor


def build_vae(cfg: VAEConfig, state_dict: dict, prefix: str = "first_stage_model.") -> AutoencoderKL:
    with torch.device("meta"):          # skip the (slow) random init; parameters are assigned from the checkpoint
        net = AutoencoderKL(cfg)
    own = {k: state_dict[prefix + k].float() for k in net.state_dict().keys()}
    net.load_state_dict(own, strict=True, assign=True)
    return net.eval().requires_grad_(False)


def to_uint8_hwc(x_decoded: torch.Tensor):
    """modules/processing.py:1004-1005, 1034-1035: clamp((x+1)/2,0,1) then 255*x truncated to uint8, HWC."""
    import numpy as np
    x = torch.clamp((x_decoded.float() + 1.0) / 2.0, min=0.0, max=1.0)
    out = []
    for img in x:
        a = 255. * np.moveaxis(img.cpu().numpy(), 0, 2)
        out.append(a.astype(np.uint8))
    return np.stack(out)
