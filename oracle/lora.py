"""fp32 CPU restatement of LoRA weight merging (oracle; tests only).

Follows /root/reference/extensions-builtin/Lora: key grouping and lookup networks.py:183-240, diffusers -> compvis layer
names networks.py:56-120 (pinned by tests/golden/lora_names.json, produced by executing that function), module creation
network_lora.py:9-35, delta network_lora.py:65-80 + lyco_helpers.py:9-15, scaling network.py:161-173 / 196-216, weight
rewrite networks.py:455-472 (W + updown per loaded network, in list order).  UNet layers only (the text encoder is not
on the engine's path); cp-decomposition (lora_mid), DoRA, bias and the other LyCORIS module types are not restated."""
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
        if k.startswith(prefix) and k.endswith(".weight") and v.dim() in (2, 4):
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


def calc_updown(w: dict, orig_shape, multiplier: float) -> torch.Tensor:
    if "lora_A.weight" in w:
        w = dict(w, **{"lora_up.weight": w["lora_B.weight"], "lora_down.weight": w["lora_A.weight"]})
    up, down = w["lora_up.weight"].float(), w["lora_down.weight"].float()
    if "lora_mid.weight" in w:
        raise NotImplementedError("cp-decomposition")
    dim = down.shape[0]
    updown = (up.reshape(up.size(0), -1) @ down.reshape(down.size(0), -1)).reshape(orig_shape)
    if "scale" in w:
        scale = w["scale"].item()
    elif "alpha" in w:
        scale = w["alpha"].item() / dim
    else:
        scale = 1.0
    return updown * scale * multiplier


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
            out[ck] = base + calc_updown(w, base.shape, mult)
    return out
