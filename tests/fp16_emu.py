"""Emulation of the reference's *own* default GPU numerics on the CPU oracle (test infrastructure).

The webui's default GPU configuration is ``model.half()`` under ``torch.autocast("cuda")`` (SURVEY.md section 3.5;
/root/reference/modules/sd_models.py:482-491, modules/devices.py:210-231): every conv / linear / einsum accumulates in fp32 and
*stores an fp16 result*; GroupNorm32 computes in fp32 and casts back to fp16 (modules/devices.py:284-295 +
``.type(x.dtype)``); residual adds, SiLU, GELU and the GEGLU product run on fp16 tensors; the split-attention forward keeps the
score matrix and the softmax output in ``q.dtype`` (modules/sd_hijack_optimizations.py:262-268: ``s2 = s1.softmax(dim=-1,
dtype=q.dtype)``).  ``fp16_storage(net)`` reproduces exactly that rounding pattern on the fp32 oracle modules: arithmetic stays
fp32 (= fp32 accumulation), every tensor a GPU kernel would *write* is rounded to binary16.

Used by the C1 parity tests to measure how far the reference's fp16 path itself sits from its fp32 CPU path — the yardstick the
engine's own distance is compared with (profiles/r02_parity.json).
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import unet as ou
from oracle import vae as ov


def r16(t: torch.Tensor) -> torch.Tensor:
    return t.half().float()


def _res_forward(self, x, emb):
    h = self.in_layers(x)                                       # GroupNorm / SiLU / Conv outputs rounded by the leaf hooks
    h = r16(h + self.emb_layers(emb)[:, :, None, None])
    h = self.out_layers(h)
    return r16(self.skip_connection(x) + h)


def _attn_forward(self, x, context=None):
    h = self.heads
    context = x if context is None else context
    q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
    b, n, _ = q.shape
    split = lambda t: t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)
    q, k, v = split(q), split(k), split(v)
    sim = r16(r16(torch.einsum('bid,bjd->bij', q, k)) * self.scale)
    attn = r16(sim.softmax(dim=-1))
    out = r16(torch.einsum('bij,bjd->bid', attn, v))
    out = out.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
    return self.to_out(out)


def _geglu_forward(self, x):
    x, gate = self.proj(x).chunk(2, dim=-1)
    return r16(x * r16(F.gelu(gate)))


def _tblock_forward(self, x, context=None):
    x = r16(self.attn1(self.norm1(x)) + x)
    x = r16(self.attn2(self.norm2(x), context=context) + x)
    x = r16(self.ff(self.norm3(x)) + x)
    return x


def _st_forward(self, x, context=None):
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
    return r16(x + x_in)


def _vae_res_forward(self, x):
    h = self.conv1(r16(F.silu(self.norm1(x))))
    h = self.conv2(r16(F.silu(self.norm2(h))))
    if self.cin != self.cout:
        x = self.nin_shortcut(x)
    return r16(x + h)


def _vae_attn_forward(self, x):
    h = self.norm(x)
    q, k, v = self.q(h), self.k(h), self.v(h)
    b, c, hh, ww = q.shape
    q, k, v = [t.reshape(b, c, hh * ww).permute(0, 2, 1) for t in (q, k, v)]
    w = r16(torch.softmax(r16(r16(torch.bmm(q, k.transpose(1, 2))) * (int(c) ** -0.5)), dim=-1))
    o = r16(torch.bmm(w, v)).permute(0, 2, 1).reshape(b, c, hh, ww)
    return r16(x + self.proj_out(o))


def _vae_decoder_forward(self, z):
    h = self.conv_in(z)
    h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
    for i in reversed(range(self.nres)):
        for j in range(self.nrb + 1):
            h = self.up[i].block[j](h)
        if i != 0:
            h = self.up[i].upsample(h)
    return self.conv_out(r16(F.silu(self.norm_out(h))))


_PATCHES = [(ou.ResBlock, _res_forward), (ou.CrossAttention, _attn_forward), (ou.GEGLU, _geglu_forward),
            (ou.BasicTransformerBlock, _tblock_forward), (ou.SpatialTransformer, _st_forward),
            (ov.ResnetBlock, _vae_res_forward), (ov.AttnBlock, _vae_attn_forward), (ov.Decoder, _vae_decoder_forward)]
_LEAVES = (nn.Conv2d, nn.Linear, nn.GroupNorm, nn.LayerNorm, nn.SiLU)


@contextlib.contextmanager
def fp16_storage(net: nn.Module):
    """Inside the block ``net`` (an oracle UNetModel / AutoencoderKL / Decoder) computes with fp32 arithmetic and fp16-rounded
    stores at every point where the reference's half-precision autocast path materialises a tensor."""
    saved = [(cls, cls.forward) for cls, _ in _PATCHES]
    handles = []
    try:
        for cls, fn in _PATCHES:
            cls.forward = fn
        for m in net.modules():
            if isinstance(m, _LEAVES):
                handles.append(m.register_forward_hook(lambda mod, inp, out: r16(out)))
        yield net
    finally:
        for h in handles:
            h.remove()
        for cls, fn in saved:
            cls.forward = fn


def capture_outputs(net: nn.Module, names) -> tuple:
    """Forward hooks that record the output of the named sub-modules (error budget).  Returns (dict, handles)."""
    got, handles = {}, []
    mods = dict(net.named_modules())
    for n in names:
        if n in mods:
            handles.append(mods[n].register_forward_hook(lambda mod, inp, out, n=n: got.__setitem__(n, out.detach().clone())))
    return got, handles
