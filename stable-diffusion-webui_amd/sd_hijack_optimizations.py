"""Boundary B2: the fused attention kernel as an ``SdOptimization`` (modules/sd_hijack_optimizations.py:25-48).

``SdOptimizationMi355x.apply()`` assigns ``mi355x_attention_forward`` to ``CrossAttention.forward`` of ldm / sgm exactly
like the in-tree optimizations do (modules/sd_hijack_optimizations.py:50-143), so the kernel runs inside the UNMODIFIED
torch UNet; registration goes through ``script_callbacks.on_list_optimizers`` (modules/script_callbacks.py:594-599).
The forward keeps the reference structure: to_q / to_k / to_v (+ hypernetworks hook, :227) in torch, attention math
(:236-272) in the HIP kernel, to_out in torch.
"""
from __future__ import annotations

import torch

from . import ops, shared

try:
    from modules import sd_hijack_optimizations as _ref
    SdOptimization = _ref.SdOptimization
except Exception:
    class SdOptimization:                      # modules/sd_hijack_optimizations.py:25-48
        name: str = None
        label = None
        cmd_opt = None
        priority: int = 0

        def title(self):
            return self.name if self.label is None else f"{self.name} - {self.label}"

        def is_available(self):
            return True

        def apply(self):
            pass

        def undo(self):
            pass


def mi355x_attention_forward(self, x, context=None, mask=None, **kwargs):
    """Drop-in for CrossAttention.forward(self, x, context=None, mask=None) — same contract as
    split_cross_attention_forward (modules/sd_hijack_optimizations.py:221-281)."""
    h = self.heads
    q_in = self.to_q(x)
    context = x if context is None else context
    try:                                        # standalone (no webui on the path) there are no hypernetworks to apply ...
        from modules import shared as _shared
        from modules.hypernetworks import hypernetwork as _hn
    except ImportError:
        context_k, context_v = context, context
    else:                                       # ... inside the webui their errors are the user's to see, as with every in-tree optimizer (:227)
        context_k, context_v = _hn.apply_hypernetworks(_shared.loaded_hypernetworks, context)
    k_in = self.to_k(context_k)
    v_in = self.to_v(context_v)
    dtype = q_in.dtype
    out = ops.attention(q_in.half(), k_in.half(), v_in.half(), heads=h, scale=getattr(self, "scale", None))
    return self.to_out(out.to(dtype))


def mi355x_attnblock_forward(self, x):
    """Drop-in for ldm.modules.diffusionmodules.model.AttnBlock.forward(self, x) — the single-head VAE mid-block attention
    (N = h*w tokens, d = C = 512) that every in-tree optimizer replaces too (modules/sd_hijack_optimizations.py:554-610, 613-676):
    norm / q / k / v / proj_out stay torch modules, softmax(q k^T * C^-0.5) v runs in sdmi_attention_wide."""
    h_ = self.norm(x)
    q, k, v = self.q(h_), self.k(h_), self.v(h_)
    b, c, h, w = q.shape
    if q.dtype == torch.float32 or getattr(shared.opts, "upcast_attn", False):
        #  a VAE running in fp32 — --no-half-vae, or the webui's automatic fp32 retry after a NaN decode
        # (modules/processing.py:636-665) — must not get fp16 attention operands: the in-tree optimizers keep the module dtype or upcast
        # (modules/sd_hijack_optimizations.py:232-233, 572-580).  torch's fp32 attention on these 4096 x 512 tokens instead.
        qt, kt, vt = (t.reshape(b, c, h * w).transpose(1, 2).float() for t in (q, k, v))
        out = torch.softmax(qt @ kt.transpose(1, 2) * (int(c) ** (-0.5)), dim=-1) @ vt
        out = out.to(q.dtype).transpose(1, 2).reshape(b, c, h, w)
        return x + self.proj_out(out)
    tokens = lambda t: t.reshape(b, c, h * w).transpose(1, 2).half().contiguous()
    out = ops.attention(tokens(q), tokens(k), tokens(v), heads=1, scale=int(c) ** (-0.5))
    out = out.to(q.dtype).transpose(1, 2).reshape(b, c, h, w)
    return x + self.proj_out(out)


class SdOptimizationMi355x(SdOptimization):
    name = "mi355x"
    label = "MFMA flash attention (gfx950)"
    cmd_opt = "opt_mi355x_attention"
    priority = 110                              # above xformers (100): modules/sd_hijack_optimizations.py:50-64

    def is_available(self):
        from . import _lib
        return _lib.device_ok()

    def apply(self):
        import ldm.modules.attention
        import ldm.modules.diffusionmodules.model
        ldm.modules.attention.CrossAttention.forward = mi355x_attention_forward
        ldm.modules.diffusionmodules.model.AttnBlock.forward = mi355x_attnblock_forward      # as :62-63, 76-77 ... of the in-tree rows
        try:
            import sgm.modules.attention
            import sgm.modules.diffusionmodules.model
            sgm.modules.attention.CrossAttention.forward = mi355x_attention_forward
            sgm.modules.diffusionmodules.model.AttnBlock.forward = mi355x_attnblock_forward
        except ImportError:
            pass
