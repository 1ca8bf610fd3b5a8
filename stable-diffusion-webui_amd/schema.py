"""State-dict schemas (names + shapes) of the models on the hot path, and the synthetic-weight generator.

Names follow the checkpoint namespaces the reference loader probes: ``model.diffusion_model.*``
(/root/reference/modules/sd_models.py:392) and ``first_stage_model.*`` (:452-454); the UNet module layout is
pinned in-tree by extensions-builtin/Lora/networks.py:43-98.  Architectures: configs/v1-inference.yaml:29-67
(SD1.5), configs/sd_xl_inpaint.yaml:19-98 (SDXL, in_channels 4 for base).

Synthetic weights follow SURVEY.md section 8(d): seed 0x5D15, Conv/Linear ~ N(0, 1/fan_in), norm weight
1 + N(0, 0.02^2), every bias N(0, 0.02^2); ``alphas_cumprod`` from the linear-sqrt-beta schedule.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

# torch is imported where tensors are made (synthetic_state_dict, make_alphas_cumprod): the schemas themselves are plain Python, so the
# torch-free kernel harness (tools/gpu/fwd_ab.py) can import this module without paying for `import torch` on a fresh GPU box

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    channel_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    attention_resolutions: Sequence[int] = (4, 2, 1)
    num_heads: int = 8
    num_head_channels: int = -1
    transformer_depth: Union[int, Sequence[int]] = 1
    context_dim: int = 768
    use_linear_in_transformer: bool = False
    adm_in_channels: Optional[int] = None

    def depth_at(self, level: int) -> int:
        td = self.transformer_depth
        return td if isinstance(td, int) else td[level]

    def heads_for(self, ch: int) -> Tuple[int, int]:
        if self.num_head_channels == -1:
            return self.num_heads, ch // self.num_heads
        return ch // self.num_head_channels, self.num_head_channels


@dataclass
class VAEConfig:
    ch: int = 128
    ch_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    in_channels: int = 3
    out_ch: int = 3
    z_channels: int = 4
    scale_factor: float = 0.18215


@dataclass
class ClipConfig:
    """transformers CLIPTextConfig fields the text transformer uses (SD1.x CLIP-L: configs of openai/clip-vit-large-patch14)."""
    vocab_size: int = 49408
    max_positions: int = 77
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    act: str = "quick_gelu"          # "quick_gelu" (OpenAI CLIP) | "gelu" (OpenCLIP)
    eps: float = 1e-5
    proj_dim: Optional[int] = None   # text_projection output width (pooled vector), None = no projection


CLIP_PREFIX = "cond_stage_model.transformer.text_model."     # where an SD1.x checkpoint keeps the text encoder


def sd15_clip() -> ClipConfig:
    return ClipConfig()


def openclip_h() -> ClipConfig:
    """OpenCLIP ViT-H/14 text tower (SD 2.x cond_stage_model, configs/v2-inference*.yaml: FrozenOpenCLIPEmbedder, penultimate)."""
    return ClipConfig(hidden=1024, layers=24, heads=16, intermediate=4096, act="gelu")


def openclip_bigg() -> ClipConfig:
    """OpenCLIP ViT-bigG/14 text tower (SDXL conditioner.embedders.1, FrozenOpenCLIPEmbedder2: penultimate + pooled)."""
    return ClipConfig(hidden=1280, layers=32, heads=20, intermediate=5120, act="gelu", proj_dim=1280)


def openclip_to_transformers_keys(sd: dict, prefix: str, out_prefix: str = CLIP_PREFIX) -> dict:
    """open_clip text-tower state dict (keys below ``prefix``, e.g. "cond_stage_model.model." for SD 2.x or
    "conditioner.embedders.1.model." for SDXL) -> the transformers CLIPTextModel layout the engine loads:
    nn.MultiheadAttention's packed in_proj_{weight,bias} [3C, ...] split into q / k / v, ln_1 / ln_2 / c_fc / c_proj renamed,
    positional_embedding -> position_embedding.weight, text_projection [C, P] (used as x @ W) -> nn.Linear weight [P, C]."""
    out = {}
    g = lambda k: sd[prefix + k]
    out[out_prefix + "embeddings.token_embedding.weight"] = g("token_embedding.weight")
    out[out_prefix + "embeddings.position_embedding.weight"] = g("positional_embedding")
    i = 0
    while prefix + f"transformer.resblocks.{i}.ln_1.weight" in sd:
        src, dst = f"transformer.resblocks.{i}.", out_prefix + f"encoder.layers.{i}."
        w, b = g(src + "attn.in_proj_weight"), g(src + "attn.in_proj_bias")
        c = w.shape[1]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            out[dst + f"self_attn.{n}.weight"] = w[j * c:(j + 1) * c].contiguous()
            out[dst + f"self_attn.{n}.bias"] = b[j * c:(j + 1) * c].contiguous()
        for a, bname in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                         ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            out[dst + bname + ".weight"] = g(src + a + ".weight")
            out[dst + bname + ".bias"] = g(src + a + ".bias")
        i += 1
    out[out_prefix + "final_layer_norm.weight"] = g("ln_final.weight")
    out[out_prefix + "final_layer_norm.bias"] = g("ln_final.bias")
    if prefix + "text_projection" in sd:
        out[out_prefix + "text_projection.weight"] = g("text_projection").t().contiguous()
    return out


def tiny_clip(**kw) -> ClipConfig:
    base = dict(vocab_size=1000, max_positions=77, hidden=128, layers=3, heads=2, intermediate=256, act="quick_gelu")
    base.update(kw)
    return ClipConfig(**base)


def clip_schema(cfg: ClipConfig):
    """[(key below "text_model.", shape, kind)] of transformers' CLIPTextModel state dict (position_ids buffer excluded)."""
    C, I = cfg.hidden, cfg.intermediate
    out = [("embeddings.token_embedding.weight", (cfg.vocab_size, C), "e"),
           ("embeddings.position_embedding.weight", (cfg.max_positions, C), "e")]
    for i in range(cfg.layers):
        b = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out += [(b + f"self_attn.{n}.weight", (C, C), "w"), (b + f"self_attn.{n}.bias", (C,), "b")]
        out += [(b + "layer_norm1.weight", (C,), "g"), (b + "layer_norm1.bias", (C,), "b"),
                (b + "mlp.fc1.weight", (I, C), "w"), (b + "mlp.fc1.bias", (I,), "b"),
                (b + "mlp.fc2.weight", (C, I), "w"), (b + "mlp.fc2.bias", (C,), "b"),
                (b + "layer_norm2.weight", (C,), "g"), (b + "layer_norm2.bias", (C,), "b")]
    out += [("final_layer_norm.weight", (C,), "g"), ("final_layer_norm.bias", (C,), "b")]
    if cfg.proj_dim:
        out += [("text_projection.weight", (cfg.proj_dim, C), "w")]
    return out


def sd15_unet() -> UNetConfig:
    return UNetConfig()


def sdxl_unet() -> UNetConfig:
    return UNetConfig(channel_mult=(1, 2, 4), attention_resolutions=(4, 2), num_heads=-1, num_head_channels=64,
                      transformer_depth=(1, 2, 10), context_dim=2048, use_linear_in_transformer=True,
                      adm_in_channels=2816)


def sd21_unet() -> UNetConfig:
    """SD 2.x (configs/v2-inference(-v).yaml of Stability-AI/stablediffusion, selected at modules/sd_models_config.py:86-94):
    OpenCLIP-H context (1024), 64-channel heads, linear proj_in / proj_out."""
    return UNetConfig(num_heads=-1, num_head_channels=64, context_dim=1024, use_linear_in_transformer=True)


def tiny_unet(**kw) -> UNetConfig:
    """Small SD-shaped UNet for tests / smoke (channels multiples of 64, head size 64 so the MFMA kernels are used)."""
    base = dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(1, 2), num_heads=-1,
                num_head_channels=64, transformer_depth=1, context_dim=64)
    base.update(kw)
    return UNetConfig(**base)


def tiny_vae(**kw) -> VAEConfig:
    base = dict(ch=64, ch_mult=(1, 2), num_res_blocks=1)
    base.update(kw)
    return VAEConfig(**base)


def sd15_vae() -> VAEConfig:
    return VAEConfig()


def sdxl_vae() -> VAEConfig:
    return VAEConfig(scale_factor=0.13025)


Entry = Tuple[str, Tuple[int, ...], str]   # (key, shape, kind) kind in {"w", "b", "g"} (weight / bias / norm gain)


def _conv(out: List[Entry], name, cin, cout, k):
    out.append((name + ".weight", (cout, cin, k, k), "w"))
    out.append((name + ".bias", (cout,), "b"))


def _lin(out: List[Entry], name, cin, cout, bias=True):
    out.append((name + ".weight", (cout, cin), "w"))
    if bias:
        out.append((name + ".bias", (cout,), "b"))


def _norm(out: List[Entry], name, c):
    out.append((name + ".weight", (c,), "g"))
    out.append((name + ".bias", (c,), "b"))


def unet_blocks(cfg: UNetConfig):
    """Walk the UNet the way ldm's constructor does and yield a structural description.

    Returns dict with 'input' / 'middle' / 'output': lists of blocks; each block is a list of layer tuples:
      ("conv_in", cin, cout) | ("res", cin, cout) | ("st", ch, heads, dhead, depth) | ("down", ch) | ("up", ch)
    """
    mc = cfg.model_channels
    inp = [[("conv_in", cfg.in_channels, mc)]]
    chans = [mc]
    ch, ds = mc, 1
    nlev = len(cfg.channel_mult)
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                nh, dh = cfg.heads_for(ch)
                layers.append(("st", ch, nh, dh, cfg.depth_at(level)))
            inp.append(layers)
            chans.append(ch)
        if level != nlev - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    nh, dh = cfg.heads_for(ch)
    mid = [("res", ch, ch), ("st", ch, nh, dh, cfg.depth_at(nlev - 1)), ("res", ch, ch)]
    outb = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult, ch, ich)]   # extra: split of the concat (h, skip)
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                nh, dh = cfg.heads_for(ch)
                layers.append(("st", ch, nh, dh, cfg.depth_at(level)))
            if level and i == cfg.num_res_blocks:
                layers.append(("up", ch))
                ds //= 2
            outb.append(layers)
    return {"input": inp, "middle": mid, "output": outb, "final_ch": ch}


def unet_schema(cfg: UNetConfig) -> List[Entry]:
    mc, ted = cfg.model_channels, cfg.model_channels * 4
    out: List[Entry] = []
    _lin(out, "time_embed.0", mc, ted)
    _lin(out, "time_embed.2", ted, ted)
    if cfg.adm_in_channels is not None:
        _lin(out, "label_emb.0.0", cfg.adm_in_channels, ted)
        _lin(out, "label_emb.0.2", ted, ted)

    def res(name, cin, cout):
        _norm(out, name + ".in_layers.0", cin)
        _conv(out, name + ".in_layers.2", cin, cout, 3)
        _lin(out, name + ".emb_layers.1", ted, cout)
        _norm(out, name + ".out_layers.0", cout)
        _conv(out, name + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            _conv(out, name + ".skip_connection", cin, cout, 1)

    def st(name, ch, nh, dh, depth):
        inner = nh * dh
        _norm(out, name + ".norm", ch)
        if cfg.use_linear_in_transformer:
            _lin(out, name + ".proj_in", ch, inner)
        else:
            _conv(out, name + ".proj_in", ch, inner, 1)
        for d in range(depth):
            tb = f"{name}.transformer_blocks.{d}"
            _lin(out, tb + ".attn1.to_q", inner, inner, bias=False)
            _lin(out, tb + ".attn1.to_k", inner, inner, bias=False)
            _lin(out, tb + ".attn1.to_v", inner, inner, bias=False)
            _lin(out, tb + ".attn1.to_out.0", inner, inner)
            _lin(out, tb + ".ff.net.0.proj", inner, inner * 8)
            _lin(out, tb + ".ff.net.2", inner * 4, inner)
            _lin(out, tb + ".attn2.to_q", inner, inner, bias=False)
            _lin(out, tb + ".attn2.to_k", cfg.context_dim, inner, bias=False)
            _lin(out, tb + ".attn2.to_v", cfg.context_dim, inner, bias=False)
            _lin(out, tb + ".attn2.to_out.0", inner, inner)
            _norm(out, tb + ".norm1", inner)
            _norm(out, tb + ".norm2", inner)
            _norm(out, tb + ".norm3", inner)
        if cfg.use_linear_in_transformer:
            _lin(out, name + ".proj_out", inner, ch)
        else:
            _conv(out, name + ".proj_out", inner, ch, 1)

    def walk(prefix, blocks):
        for bi, layers in enumerate(blocks):
            for li, layer in enumerate(layers):
                name = f"{prefix}.{bi}.{li}" if prefix != "middle_block" else f"{prefix}.{li}"
                kind = layer[0]
                if kind == "conv_in":
                    _conv(out, name, layer[1], layer[2], 3)
                elif kind == "res":
                    res(name, layer[1], layer[2])
                elif kind == "st":
                    st(name, *layer[1:])
                elif kind == "down":
                    _conv(out, name + ".op", layer[1], layer[1], 3)
                elif kind == "up":
                    _conv(out, name + ".conv", layer[1], layer[1], 3)

    b = unet_blocks(cfg)
    walk("input_blocks", b["input"])
    walk("middle_block", [b["middle"]])
    walk("output_blocks", b["output"])
    _norm(out, "out.0", b["final_ch"])
    _conv(out, "out.2", mc, cfg.out_channels, 3)
    return out


def vae_schema(cfg: VAEConfig) -> List[Entry]:
    out: List[Entry] = []

    def res(name, cin, cout):
        _norm(out, name + ".norm1", cin)
        _conv(out, name + ".conv1", cin, cout, 3)
        _norm(out, name + ".norm2", cout)
        _conv(out, name + ".conv2", cout, cout, 3)
        if cin != cout:
            _conv(out, name + ".nin_shortcut", cin, cout, 1)

    def attn(name, c):
        _norm(out, name + ".norm", c)
        for k in ("q", "k", "v", "proj_out"):
            _conv(out, f"{name}.{k}", c, c, 1)

    ch, nres = cfg.ch, len(cfg.ch_mult)
    # encoder
    _conv(out, "encoder.conv_in", cfg.in_channels, ch, 3)
    in_mult = (1,) + tuple(cfg.ch_mult)
    bi = ch
    for i in range(nres):
        bi, bo = ch * in_mult[i], ch * cfg.ch_mult[i]
        for j in range(cfg.num_res_blocks):
            res(f"encoder.down.{i}.block.{j}", bi, bo)
            bi = bo
        if i != nres - 1:
            _conv(out, f"encoder.down.{i}.downsample.conv", bi, bi, 3)
    res("encoder.mid.block_1", bi, bi)
    attn("encoder.mid.attn_1", bi)
    res("encoder.mid.block_2", bi, bi)
    _norm(out, "encoder.norm_out", bi)
    _conv(out, "encoder.conv_out", bi, 2 * cfg.z_channels, 3)
    # decoder
    bi = ch * cfg.ch_mult[nres - 1]
    _conv(out, "decoder.conv_in", cfg.z_channels, bi, 3)
    res("decoder.mid.block_1", bi, bi)
    attn("decoder.mid.attn_1", bi)
    res("decoder.mid.block_2", bi, bi)
    for i in reversed(range(nres)):
        bo = ch * cfg.ch_mult[i]
        for j in range(cfg.num_res_blocks + 1):
            res(f"decoder.up.{i}.block.{j}", bi, bo)
            bi = bo
        if i != 0:
            _conv(out, f"decoder.up.{i}.upsample.conv", bi, bi, 3)
    _norm(out, "decoder.norm_out", bi)
    _conv(out, "decoder.conv_out", bi, cfg.out_ch, 3)
    _conv(out, "quant_conv", 2 * cfg.z_channels, 2 * cfg.z_channels, 1)
    _conv(out, "post_quant_conv", cfg.z_channels, cfg.z_channels, 1)
    return out


def make_alphas_cumprod(linear_start=0.00085, linear_end=0.0120, n=1000) -> torch.Tensor:
    """ldm 'linear' schedule (configs/v1-inference.yaml:5-9; restated in-tree at ddpm_edit.py:133-154)."""
    import torch
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2
    return torch.tensor(np.cumprod(1.0 - betas.numpy(), axis=0), dtype=torch.float32)


def synthetic_state_dict(unet_cfg: Optional[UNetConfig] = None, vae_cfg: Optional[VAEConfig] = None,
                         seed: int = 0x5D15, dtype=None, device="cpu", clip_cfg: Optional["ClipConfig"] = None) -> dict:
    """Seeded synthetic checkpoint in the reference's state-dict schema (no checkpoint exists offline).

    Values are generated in fp32 on ``device`` then cast to ``dtype`` (fp16 = what ``model.half()`` leaves in a
    loaded checkpoint, modules/sd_models.py:482-486); ``alphas_cumprod`` stays fp32 as in the reference.
    """
    import torch
    dtype = torch.float16 if dtype is None else dtype
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}

    def fill(prefix, entries):
        for key, shape, kind in entries:
            if kind == "w":
                fan_in = int(np.prod(shape[1:]))
                t = torch.randn(shape, generator=g, dtype=torch.float32) * (fan_in ** -0.5)
            elif kind == "e":
                t = 0.5 * torch.randn(shape, generator=g, dtype=torch.float32)
            elif kind == "g":
                t = 1.0 + 0.02 * torch.randn(shape, generator=g, dtype=torch.float32)
            else:
                t = 0.02 * torch.randn(shape, generator=g, dtype=torch.float32)
            sd[prefix + key] = t.to(dtype).to(device)

    if unet_cfg is not None:
        fill(UNET_PREFIX, unet_schema(unet_cfg))
    if vae_cfg is not None:
        fill(VAE_PREFIX, vae_schema(vae_cfg))
    if clip_cfg is not None:
        fill(CLIP_PREFIX, clip_schema(clip_cfg))
    sd["alphas_cumprod"] = make_alphas_cumprod().to(device)
    return sd
