"""fp32 CPU restatement of the CLIP text transformer (oracle; tests only).

What `FrozenCLIPEmbedderWithCustomWords.encode_with_transformers` runs (/root/reference/modules/sd_hijack_clip.py:351-360):
transformers' CLIPTextModel (third-party; the reference pins transformers==4.30.2, requirements_versions.txt) — token +
position embeddings, pre-LN blocks with CAUSAL self-attention (q scaled by head_dim**-0.5), MLP with quick_gelu, final
LayerNorm; `hidden_states[-k]` = residual stream after block N-k+1 (index 0 = embeddings).  The reference carries a
plain-torch twin of the same network for SD3 (modules/models/sd3/other_impls.py:61-150), which tests/golden/make_golden.py
executes to produce tests/golden/clip_text.npz; tests/test_oracle_pins.py additionally compares against the installed
`transformers` CLIPTextModel when it is importable.  State-dict keys are transformers' (below "text_model.")."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class ClipConfig:
    vocab_size: int = 49408
    max_positions: int = 77
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    act: str = "quick_gelu"
    eps: float = 1e-5
    proj_dim: int = None


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


class ClipAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.heads = cfg.heads
        self.q_proj = nn.Linear(cfg.hidden, cfg.hidden)
        self.k_proj = nn.Linear(cfg.hidden, cfg.hidden)
        self.v_proj = nn.Linear(cfg.hidden, cfg.hidden)
        self.out_proj = nn.Linear(cfg.hidden, cfg.hidden)

    def forward(self, x, mask):
        b, l, c = x.shape
        d = c // self.heads
        q = self.q_proj(x) * d ** -0.5
        k, v = self.k_proj(x), self.v_proj(x)
        sp = lambda t: t.view(b, l, self.heads, d).transpose(1, 2)
        w = sp(q) @ sp(k).transpose(-1, -2) + mask
        o = w.softmax(dim=-1) @ sp(v)
        return self.out_proj(o.transpose(1, 2).reshape(b, l, c))


class ClipMLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.hidden, cfg.intermediate)
        self.fc2 = nn.Linear(cfg.intermediate, cfg.hidden)
        self.act = quick_gelu if cfg.act == "quick_gelu" else F.gelu

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class ClipLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(cfg.hidden, eps=cfg.eps)
        self.self_attn = ClipAttention(cfg)
        self.layer_norm2 = nn.LayerNorm(cfg.hidden, eps=cfg.eps)
        self.mlp = ClipMLP(cfg)

    def forward(self, x, mask):
        x = x + self.self_attn(self.layer_norm1(x), mask)
        return x + self.mlp(self.layer_norm2(x))


class ClipEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden)
        self.position_embedding = nn.Embedding(cfg.max_positions, cfg.hidden)

    def forward(self, tokens, inputs_embeds=None):
        e = self.token_embedding(tokens) if inputs_embeds is None else inputs_embeds
        return e + self.position_embedding.weight[: e.shape[1]]


class ClipEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([ClipLayer(cfg) for _ in range(cfg.layers)])


class ClipTextModel(nn.Module):
    """Module tree == transformers' CLIPTextModel.text_model, so the checkpoint keys load unchanged."""

    def __init__(self, cfg: ClipConfig):
        super().__init__()
        self.cfg = cfg
        self.embeddings = ClipEmbeddings(cfg)
        self.encoder = ClipEncoder(cfg)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden, eps=cfg.eps)
        self.text_projection = nn.Linear(cfg.hidden, cfg.proj_dim, bias=False) if cfg.proj_dim else None

    def hidden_states(self, tokens, inputs_embeds=None):
        x = self.embeddings(tokens, inputs_embeds)
        l = x.shape[1]
        mask = torch.full((l, l), float("-inf"), dtype=x.dtype).triu_(1)
        hs = [x]
        for layer in self.encoder.layers:
            x = layer(x, mask)
            hs.append(x)
        return hs

    @torch.no_grad()
    def forward(self, tokens, skip: int = 1, apply_final_ln: bool = True, inputs_embeds=None, return_pooled: bool = False):
        """= encode_with_transformers with opts.CLIP_stop_at_last_layers = skip (sd_hijack_clip.py:351-360); pooled = the row
        at the EOS (argmax id) position, as transformers' pooler_output / other_impls.py:146."""
        hs = self.hidden_states(tokens, inputs_embeds)
        z = hs[-skip]
        if apply_final_ln:
            z = self.final_layer_norm(z)
        if return_pooled:
            # always from the LAST block + final norm (transformers pooler_output; open_clip pool(ln_final(x)) @ text_projection)
            last = self.final_layer_norm(hs[-1])
            pooled = last[torch.arange(last.shape[0]), tokens.to(torch.int).argmax(dim=-1)]
            if self.text_projection is not None:
                pooled = self.text_projection(pooled)
            return z, pooled
        return z


def build_clip(cfg: ClipConfig, state_dict: dict, prefix: str = "cond_stage_model.transformer.text_model.") -> ClipTextModel:
    m = ClipTextModel(cfg)
    sd = {k[len(prefix):]: v.float() for k, v in state_dict.items() if k.startswith(prefix) and not k.endswith("position_ids")}
    m.load_state_dict(sd, strict=True)
    return m.eval()
