"""Text-encoder side of the drop-in: the engine behind `encode_with_transformers`.

Reference: /root/reference/modules/sd_hijack_clip.py — `FrozenCLIPEmbedderWithCustomWords` keeps tokenisation, 75-token
chunking, emphasis and textual-inversion fix-ups on the host and sends every 77-token chunk through
`encode_with_transformers(tokens)` (:351-360), which calls transformers' CLIPTextModel.  This module provides that method on
top of the engine (`sdmi_clip_forward`), with the same clip-skip semantics (`opts.CLIP_stop_at_last_layers`), plus the SDXL
CLIP-L variant (:369-377: hidden_states[-2], no final norm).  INTEGRATION.md shows the two-line monkeypatch that routes the
webui's embedder here; token embeddings patched by textual inversion (modules/sd_hijack.py EmbeddingsWithFixes) enter through
``inputs_embeds``."""
from __future__ import annotations

from typing import Optional

import torch

from . import schema, shared
from .engine import Engine


class Mi355xClipTextEncoder:
    def __init__(self, engine: Engine, cfg: schema.ClipConfig, state_dict: dict, prefix: Optional[str] = None, slot: int = 0,
                 layer: str = "last", layer_idx: Optional[int] = None):
        self.engine, self.cfg, self.slot = engine, cfg, slot
        self.layer, self.layer_idx = layer, layer_idx            # sgm FrozenCLIPEmbedder fields the SDXL wrapper reads
        engine.load_clip(cfg, state_dict, prefix=prefix, slot=slot)

    def encode_with_transformers(self, tokens: torch.Tensor, inputs_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sd_hijack_clip.py:351-360"""
        skip = int(getattr(shared.opts, "CLIP_stop_at_last_layers", 1))
        if skip > 1:
            return self.engine.clip_forward(tokens, skip=skip, apply_final_ln=True, inputs_embeds=inputs_embeds, slot=self.slot)
        return self.engine.clip_forward(tokens, skip=1, apply_final_ln=True, inputs_embeds=inputs_embeds, slot=self.slot)

    def encode_with_transformers_sdxl(self, tokens: torch.Tensor, inputs_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sd_hijack_clip.py:369-377 (FrozenCLIPEmbedderForSDXLWithCustomWords)"""
        if getattr(shared.opts, "sdxl_clip_l_skip", False) is True:
            skip = int(getattr(shared.opts, "CLIP_stop_at_last_layers", 1))
            return self.engine.clip_forward(tokens, skip=skip, apply_final_ln=False, inputs_embeds=inputs_embeds, slot=self.slot)
        if self.layer == "last":
            return self.engine.clip_forward(tokens, skip=1, apply_final_ln=True, inputs_embeds=inputs_embeds, slot=self.slot)
        # hidden_states[layer_idx]: index 0 = embeddings ... layers = after the last block; -k = after block layers-k+1
        idx = self.layer_idx if self.layer_idx < 0 else self.layer_idx - (self.cfg.layers + 1)
        return self.engine.clip_forward(tokens, skip=-idx, apply_final_ln=False, inputs_embeds=inputs_embeds, slot=self.slot)

    def encode_with_transformer_openclip(self, tokens: torch.Tensor, inputs_embeds: Optional[torch.Tensor] = None):
        """SD 2.x: ldm FrozenOpenCLIPEmbedder.encode_with_transformer with layer="penultimate" (called from
        modules/sd_hijack_open_clip.py:26-30): the last block is skipped, ln_final applied."""
        return self.engine.clip_forward(tokens, skip=2, apply_final_ln=True, inputs_embeds=inputs_embeds, slot=self.slot)

    def encode_with_transformer_openclip2(self, tokens: torch.Tensor, inputs_embeds: Optional[torch.Tensor] = None):
        """SDXL: sgm FrozenOpenCLIPEmbedder2 (legacy=False) as consumed at modules/sd_hijack_open_clip.py:57-66: z = the
        penultimate hidden state WITHOUT ln_final, z.pooled = ln_final(last)[EOS] @ text_projection."""
        z, pooled = self.engine.clip_forward(tokens, skip=2, apply_final_ln=False, inputs_embeds=inputs_embeds, slot=self.slot,
                                             return_pooled=True)
        z.pooled = pooled
        return z
