"""fp32 CPU restatement of LoRA weight merging (oracle; tests only).

Follows /root/reference/extensions-builtin/Lora: key grouping and lookup networks.py:183-240, diffusers -> compvis layer
names networks.py:56-120 (pinned by tests/golden/lora_names.json, produced by executing that function), module creation
network_lora.py:9-35, delta network_lora.py:65-80 + lyco_helpers.py:9-15, scaling network.py:161-173 / 196-216, weight
rewrite networks.py:455-472 (W + updown per loaded network, in list order).  UNet layers only (the text encoder is not
on the engine's path).  Every LyCORIS module type of the reference is restated (LoRA / LoCon incl. cp-decomposition and DoRA,
LoHa, LoKr, GLoRA, IA3, full, norm, OFT / COFT / BOFT) and pinned by tests/golden/lyco.npz, which make_golden produces by
loading the reference's network_*.py files."""
from __future__ import annotations

import re

import torch

_SUFFIX = {
    "attentions": {},
    "resnets": {"conv1": "in_layers_2", "conv2": "out_layers_3", "norm1": "in_layers_0", "norm2": "out_layers_0",
                "time_emb_proj": "emb_layers_1", "conv_shortcut": "skip_connection"},
}


def convert_diffusers_name_to_compvis(key: str, is_sd2: bool = False) -> str:
    def grp(rx):
        r = re.match(rx, key)
        if not r:
            return None
        return [int(x) if re.match(r"\d+", x) else x for x in r.groups()]

    m = grp(r"lora_unet_conv_in(.*)")
    if m:
        return f"diffusion_model_input_blocks_0_0{m[0]}"
    m = grp(r"lora_unet_conv_out(.*)")
    if m:
        return f"diffusion_model_out_2{m[0]}"
    m = grp(r"lora_unet_time_embedding_linear_(\d+)(.*)")
    if m:
        return f"diffusion_model_time_embed_{m[0] * 2 - 2}{m[1]}"
    m = grp(r"lora_unet_down_blocks_(\d+)_(attentions|resnets)_(\d+)_(.+)")
    if m:
        suffix = _SUFFIX.get(m[1], {}).get(m[3], m[3])
        return f"diffusion_model_input_blocks_{1 + m[0] * 3 + m[2]}_{1 if m[1] == 'attentions' else 0}_{suffix}"
    m = grp(r"lora_unet_mid_block_(attentions|resnets)_(\d+)_(.+)")
    if m:
        suffix = _SUFFIX.get(m[0], {}).get(m[2], m[2])
        return f"diffusion_model_middle_block_{1 if m[0] == 'attentions' else m[1] * 2}_{suffix}"
    m = grp(r"lora_unet_up_blocks_(\d+)_(attentions|resnets)_(\d+)_(.+)")
    if m:
        suffix = _SUFFIX.get(m[1], {}).get(m[3], m[3])
        return f"diffusion_model_output_blocks_{m[0] * 3 + m[2]}_{1 if m[1] == 'attentions' else 0}_{suffix}"
    m = grp(r"lora_unet_down_blocks_(\d+)_downsamplers_0_conv")
    if m:
        return f"diffusion_model_input_blocks_{3 + m[0] * 3}_0_op"
    m = grp(r"lora_unet_up_blocks_(\d+)_upsamplers_0_conv")
    if m:
        return f"diffusion_model_output_blocks_{2 + m[0] * 3}_{2 if m[0] > 0 else 1}_conv"
    m = grp(r"lora_te_text_model_encoder_layers_(\d+)_(.+)")
    if m:
        if is_sd2:
            if 'mlp_fc1' in m[1]:
                return f"model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc1', 'mlp_c_fc')}"
            if 'mlp_fc2' in m[1]:
                return f"model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc2', 'mlp_c_proj')}"
            return f"model_transformer_resblocks_{m[0]}_{m[1].replace('self_attn', 'attn')}"
        return f"transformer_text_model_encoder_layers_{m[0]}_{m[1]}"
    m = grp(r"lora_te2_text_model_encoder_layers_(\d+)_(.+)")
    if m:
        if 'mlp_fc1' in m[1]:
            return f"1_model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc1', 'mlp_c_fc')}"
        if 'mlp_fc2' in m[1]:
            return f"1_model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc2', 'mlp_c_proj')}"
        return f"1_model_transformer_resblocks_{m[0]}_{m[1].replace('self_attn', 'attn')}"
    return key


def layer_mapping(unet_state_dict: dict, prefix: str = "model.diffusion_model.") -> dict:
    """network layer name -> checkpoint key of its weight, as networks.py:141-144 names compvis modules
    (``name.replace(".", "_")`` of the module path under ``sd_model.model``)."""
    out = {}
    for k, v in unet_state_dict.items():
        # conv / linear weights, and the gains of GroupNorm / LayerNorm layers (LyCORIS norm modules, network_norm.py): every
        # nn.Module under sd_model.model gets a network_layer_name (networks.py:141-144), 1-D "weight"s belong to norm layers
        if k.startswith(prefix) and k.endswith(".weight") and (v.dim() in (2, 4) or (v.dim() == 1 and k[:-len("weight")] + "bias" in unet_state_dict)):
            mod = k[len("model."):-len(".weight")]
            out[mod.replace(".", "_")] = k
    return out


def group_network(lora_sd: dict, mapping: dict, is_sd2: bool = False):
    matched, failed = {}, {}
    for key_network, weight in lora_sd.items():
        base, _, part = key_network.partition(".")
        key = convert_diffusers_name_to_compvis(base, is_sd2)
        if key not in mapping and "lora_unet" in base:
            key = base.replace("lora_unet", "diffusion_model")       # SDXL LoRAs already carry compvis names (:216-218)
        if key not in mapping:
            failed[key_network] = key
            continue
        matched.setdefault(key, {})[part] = weight
    return matched, failed


def _cp(t, wa, wb):
    """lyco_helpers.make_weight_cp (:4-6): contract the Tucker core t[i,j,k,l] with wa[i,r] and wb[j,r']."""
    temp = torch.einsum('i j k l, j r -> i r k l', t, wb)
    return torch.einsum('i j k l, i r -> r j k l', temp, wa)


def _conventional(up, down, shape, dyn_dim=None):
    """lyco_helpers.rebuild_conventional (:9-15)"""
    up, down = up.reshape(up.size(0), -1), down.reshape(down.size(0), -1)
    if dyn_dim is not None:
        up, down = up[:, :dyn_dim], down[:dyn_dim, :]
    return (up @ down).reshape(shape)


def module_kind(w: dict) -> str:
    """Dispatch of networks.py:26-36 (module_types order: lora, hada, ia3, lokr, full, norm, glora, oft)."""
    if all(x in w for x in ("lora_up.weight", "lora_down.weight")) or all(x in w for x in ("lora_A.weight", "lora_B.weight")):
        return "lora"
    if all(x in w for x in ("hada_w1_a", "hada_w1_b", "hada_w2_a", "hada_w2_b")):
        return "hada"
    if "weight" in w:
        return "ia3"
    if ("lokr_w1" in w or ("lokr_w1_a" in w and "lokr_w1_b" in w)) and ("lokr_w2" in w or ("lokr_w2_a" in w and "lokr_w2_b" in w)):
        return "lokr"
    if "diff" in w:
        return "full"
    if all(x in w for x in ("w_norm", "b_norm")):
        return "norm"
    if all(x in w for x in ("a1.weight", "a2.weight", "alpha", "b1.weight", "b2.weight")):
        return "glora"
    if "oft_blocks" in w or "oft_diag" in w:
        return "oft"
    raise AssertionError(f"Could not find a module type that would accept those keys: {', '.join(w)}")


def _oft_updown(w: dict, orig):
    """network_oft.py:26-118 — orthogonal fine-tuning: the weight's output rows are rotated block-wise, updown = R W - W.
    kohya / LyCORIS "oft_blocks" [k, n, n]: R = (I + Q)(I - Q)^-1 (Cayley) of the skew part Q = B - B^T, its norm clamped to
    alpha * out_dim when alpha is given (COFT); old LyCORIS "oft_diag": the blocks are R already; BOFT (4-D blocks [m, k, n, n]):
    m butterfly factors, factor i acting on rows regrouped with stride 2^i * n / 2; optional per-row "rescale"."""
    out_dim = orig.shape[0]
    is_r = "oft_blocks" not in w
    blocks = w["oft_diag"] if is_r else w["oft_blocks"]
    boft = blocks.dim() == 4
    n = blocks.shape[1] if is_r else blocks.shape[2] if boft else out_dim // blocks.shape[0]
    eye = torch.eye(n)
    if not is_r:
        q = blocks - blocks.transpose(-1, -2)
        constraint = (0 if w.get("alpha") is None else w["alpha"]) * out_dim
        if constraint != 0:
            norm_q = torch.norm(q.flatten())
            q = q * ((torch.clamp(norm_q, max=constraint) + 1e-8) / (norm_q + 1e-8))
        blocks = torch.matmul(eye + q, (eye - q).float().inverse())
    rest = orig.shape[1:]
    if not boft:
        k = out_dim // n
        merged = torch.einsum('k n m, k n r -> k m r', blocks, orig.reshape(k, n, -1)).reshape(out_dim, *rest)
    else:
        merged = orig
        for i in range(blocks.shape[0]):
            stride = 2 ** i * (n // 2)
            c = out_dim // (2 * stride)
            x = merged.reshape(c, 2, stride, -1).transpose(1, 2).reshape(out_dim // n, n, -1)        # "(c g k) -> (c k g)", then blocks of n
            x = torch.einsum("b i j, b j r -> b i r", blocks[i], x)
            merged = x.reshape(c, stride, 2, -1).transpose(1, 2).reshape(out_dim, *rest)               # back to "(c g k)"
    if w.get("rescale") is not None:
        merged = w["rescale"].reshape(-1, *[1] * (orig.dim() - 1)) * merged
    return merged - orig


def calc_updown(w: dict, orig_weight, multiplier: float, dyn_dim=None, with_bias: bool = False):
    """NetworkModule*.calc_updown + finalize_updown for every module type (network_lora.py:65-80, network_hada.py:28-55,
    network_lokr.py:37-64, network_glora.py:21-33, network_ia3.py:18-30, network_full.py:17-27, network_norm.py:17-28,
    network_oft.py:66-118, network.py:175-216), pinned by tests/golden/lyco.npz.  ``orig_weight`` is the layer's current weight (a shape is accepted
    where the type does not read it).  Returns updown, or (updown, ex_bias) with ``with_bias``."""
    if not torch.is_tensor(orig_weight):
        orig_weight = torch.zeros(tuple(orig_weight))
    orig = orig_weight.float()
    w = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in w.items()}
    kind = module_kind(w)
    dim, ex_bias = None, None
    if kind == "lora":
        if "lora_A.weight" in w:
            w = dict(w, **{"lora_up.weight": w["lora_B.weight"], "lora_down.weight": w["lora_A.weight"]})
        up, down = w["lora_up.weight"], w["lora_down.weight"]
        dim = down.shape[0]
        if orig.dim() == 4 and down.dim() == 2:                         # conv layers keep their factors as 1x1 convs (:36-52)
            down = down.reshape(down.shape[0], -1, 1, 1)
        if orig.dim() == 4:
            up = up.reshape(up.shape[0], -1, 1, 1)
        output_shape = [up.size(0), down.size(1)]
        if "lora_mid.weight" in w:                                       # cp-decomposition (lyco_helpers.py:18-21)
            mid = w["lora_mid.weight"]
            updown = torch.einsum('n m k l, i n, m j -> i j k l', mid, up.reshape(up.size(0), -1), down.reshape(down.size(0), -1))
            output_shape += mid.shape[2:]
        else:
            if down.dim() == 4:
                output_shape += down.shape[2:]
            updown = _conventional(up, down, output_shape, dyn_dim)
    elif kind == "hada":
        w1a, w1b, w2a, w2b = w["hada_w1_a"], w["hada_w1_b"], w["hada_w2_a"], w["hada_w2_b"]
        dim = w1b.shape[0]
        output_shape = [w1a.size(0), w1b.size(1)]
        if w.get("hada_t1") is not None:
            output_shape = [w1a.size(1), w1b.size(1)]
            updown1 = _cp(w["hada_t1"], w1a, w1b)
            output_shape += w["hada_t1"].shape[2:]
        else:
            if w1b.dim() == 4:
                output_shape += w1b.shape[2:]
            updown1 = _conventional(w1a, w1b, output_shape)
        updown2 = _cp(w["hada_t2"], w2a, w2b) if w.get("hada_t2") is not None else _conventional(w2a, w2b, output_shape)
        updown = updown1 * updown2
    elif kind == "lokr":
        if w.get("lokr_w1_b") is not None:
            dim = w["lokr_w1_b"].shape[0]
        if w.get("lokr_w2_b") is not None:
            dim = w["lokr_w2_b"].shape[0]
        w1 = w["lokr_w1"] if w.get("lokr_w1") is not None else w["lokr_w1_a"] @ w["lokr_w1_b"]
        if w.get("lokr_w2") is not None:
            w2 = w["lokr_w2"]
        elif w.get("lokr_t2") is None:
            w2 = w["lokr_w2_a"] @ w["lokr_w2_b"]
        else:
            w2 = _cp(w["lokr_t2"], w["lokr_w2_a"], w["lokr_w2_b"])
        output_shape = [w1.size(0) * w2.size(0), w1.size(1) * w2.size(1)]
        if orig.dim() == 4:
            output_shape = orig.shape
        if w2.dim() == 4:
            w1 = w1.unsqueeze(2).unsqueeze(2)
        updown = torch.kron(w1, w2.contiguous()).reshape(tuple(output_shape))
    elif kind == "glora":
        w1a, w1b, w2a, w2b = w["a1.weight"], w["b1.weight"], w["a2.weight"], w["b2.weight"]
        output_shape = [w1a.size(0), w1b.size(1)]
        updown = (w2b @ w1b) + ((orig @ w2a) @ w1a)
    elif kind == "ia3":
        ww = w["weight"]
        output_shape = [ww.size(0), orig.size(1)]
        if w["on_input"].item():
            output_shape.reverse()
        else:
            ww = ww.reshape(-1, 1)
        updown = orig * ww
    elif kind == "full":
        updown, output_shape, ex_bias = w["diff"], w["diff"].shape, w.get("diff_b")
    elif kind == "norm":
        updown, output_shape, ex_bias = w["w_norm"], w["w_norm"].shape, w.get("b_norm")
    elif kind == "oft":
        updown, output_shape = _oft_updown(w, orig), orig.shape
        w = dict(w, scale=torch.tensor(1.0))                              # NetworkModuleOFT fixes self.scale = 1.0 (:23)
    else:
        raise NotImplementedError(kind)

    # finalize_updown (network.py:196-216)
    if w.get("bias") is not None:
        updown = updown.reshape(w["bias"].shape) + w["bias"]
        updown = updown.reshape(tuple(output_shape))
    if len(output_shape) == 4:
        updown = updown.reshape(tuple(output_shape))
    if orig.numel() == updown.numel():
        updown = updown.reshape(orig.shape)
    if ex_bias is not None:
        ex_bias = ex_bias * multiplier
    if "scale" in w:
        scale = w["scale"].item()
    elif dim is not None and "alpha" in w:
        scale = w["alpha"].item() / dim
    else:
        scale = 1.0
    updown = updown * scale
    if w.get("dora_scale") is not None:                                  # apply_weight_decompose (network.py:175-194)
        merged = updown + orig
        norm = (merged.transpose(0, 1).reshape(merged.shape[1], -1).norm(dim=1, keepdim=True)
                .reshape(merged.shape[1], *[1] * (orig.dim() - 1)).transpose(0, 1))
        updown = merged * (w["dora_scale"] / norm) - orig
    updown = updown * multiplier
    return (updown, ex_bias) if with_bias else updown


def merge(state_dict: dict, networks, prefix: str = "model.diffusion_model.") -> dict:
    """``networks`` = [(lora_state_dict, unet_multiplier), ...]; returns a copy of ``state_dict`` with every matched UNet
    weight replaced by W + sum of deltas (fp32)."""
    mapping = layer_mapping(state_dict, prefix)
    out = dict(state_dict)
    for lora_sd, mult in networks:
        matched, _ = group_network(lora_sd, mapping)
        for key, w in matched.items():
            ck = mapping[key]
            base = out[ck].float()
            updown, ex_bias = calc_updown(w, base, mult, with_bias=True)
            out[ck] = base + updown
            if ex_bias is not None:                                       # networks.py:455-462: self.bias += ex_bias
                bk = ck[:-len("weight")] + "bias"
                out[bk] = out[bk].float() + ex_bias.reshape(out[bk].shape)
    return out
