"""Boundary B2: the fused attention kernel as an ``SdOptimization`` (modules/sd_hijack_optimizations.py:25-48).

``SdOptimizationMi355x.apply()`` assigns ``mi355x_attention_forward`` to ``CrossAttention.forward`` of ldm / sgm exactly
like the in-tree optimizations do (modules/sd_hijack_optimizations.py:50-143), so the kernel runs inside the UNMODIFIED
torch UNet; registration goes through ``script_callbacks.on_list_optimizers`` (modules/script_callbacks.py:594-599).
The forward keeps the reference structure: to_q / to_k / to_v (+ hypernetworks hook, :227) in torch, attention math
(:236-272) in the HIP kernel, to_out in torch.
"""
from __future__ import annotations

import torch

from . import ops

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
    try:
        from modules import shared as _shared
        from modules.hypernetworks import hypernetwork as _hn
        context_k, context_v = _hn.apply_hypernetworks(_shared.loaded_hypernetworks, context)
    except Exception:
        context_k, context_v = context, context
    k_in = self.to_k(context_k)
    v_in = self.to_v(context_v)
    dtype = q_in.dtype
    out = ops.attention(q_in.half(), k_in.half(), v_in.half(), heads=h, scale=getattr(self, "scale", None))
    return self.to_out(out.to(dtype))


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
        ldm.modules.attention.CrossAttention.forward = mi355x_attention_forward
        try:
            import sgm.modules.attention
            sgm.modules.attention.CrossAttention.forward = mi355x_attention_forward
        except Exception:
            pass
