"""LoRA for the fused UNet — host-side mirror of /root/reference/extensions-builtin/Lora (networks.py, network.py,
network_lora.py), same names and argument meaning where the reference has them.

The reference rewrites ``module.weight`` lazily inside ``torch.nn.Linear/Conv2d.forward`` (networks.py:411-480, patches at
lora_patches.py:7-18) — hooks a fused UNet never runs.  Here ``load_networks`` + ``network_apply_weights`` do the same
rewrite eagerly: for every UNet layer a loaded network touches, start from the checkpoint weight (the "backup",
networks.py:423-432), add each network's delta ``up @ down * alpha/dim * multiplier`` on the GPU (sdmi_lora_merge) and
re-pack that one layer inside the engine (sdmi_unet_update_weight).  Layers no network touches any more are restored.
Text-encoder keys are reported in ``keys_failed_to_match`` (the text encoder is outside the engine); LyCORIS module types
other than plain LoRA, cp-decomposition (lora_mid), DoRA and bias deltas raise NotImplementedError."""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr, dtype_code
from .schema import UNET_PREFIX, unet_schema

re_digits = re.compile(r"\d+")
re_compiled = {}

suffix_conversion = {                                    # networks.py:43-53
    "attentions": {},
    "resnets": {
        "conv1": "in_layers_2",
        "conv2": "out_layers_3",
        "norm1": "in_layers_0",
        "norm2": "out_layers_0",
        "time_emb_proj": "emb_layers_1",
        "conv_shortcut": "skip_connection",
    }
}


def convert_diffusers_name_to_compvis(key, is_sd2):       # networks.py:56-120
    def match(match_list, regex_text):
        regex = re_compiled.get(regex_text)
        if regex is None:
            regex = re.compile(regex_text)
            re_compiled[regex_text] = regex
        r = re.match(regex, key)
        if not r:
            return False
        match_list.clear()
        match_list.extend([int(x) if re.match(re_digits, x) else x for x in r.groups()])
        return True

    m = []
    if match(m, r"lora_unet_conv_in(.*)"):
        return f'diffusion_model_input_blocks_0_0{m[0]}'
    if match(m, r"lora_unet_conv_out(.*)"):
        return f'diffusion_model_out_2{m[0]}'
    if match(m, r"lora_unet_time_embedding_linear_(\d+)(.*)"):
        return f"diffusion_model_time_embed_{m[0] * 2 - 2}{m[1]}"
    if match(m, r"lora_unet_down_blocks_(\d+)_(attentions|resnets)_(\d+)_(.+)"):
        suffix = suffix_conversion.get(m[1], {}).get(m[3], m[3])
        return f"diffusion_model_input_blocks_{1 + m[0] * 3 + m[2]}_{1 if m[1] == 'attentions' else 0}_{suffix}"
    if match(m, r"lora_unet_mid_block_(attentions|resnets)_(\d+)_(.+)"):
        suffix = suffix_conversion.get(m[0], {}).get(m[2], m[2])
        return f"diffusion_model_middle_block_{1 if m[0] == 'attentions' else m[1] * 2}_{suffix}"
    if match(m, r"lora_unet_up_blocks_(\d+)_(attentions|resnets)_(\d+)_(.+)"):
        suffix = suffix_conversion.get(m[1], {}).get(m[3], m[3])
        return f"diffusion_model_output_blocks_{m[0] * 3 + m[2]}_{1 if m[1] == 'attentions' else 0}_{suffix}"
    if match(m, r"lora_unet_down_blocks_(\d+)_downsamplers_0_conv"):
        return f"diffusion_model_input_blocks_{3 + m[0] * 3}_0_op"
    if match(m, r"lora_unet_up_blocks_(\d+)_upsamplers_0_conv"):
        return f"diffusion_model_output_blocks_{2 + m[0] * 3}_{2 if m[0]>0 else 1}_conv"
    if match(m, r"lora_te_text_model_encoder_layers_(\d+)_(.+)"):
        if is_sd2:
            if 'mlp_fc1' in m[1]:
                return f"model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc1', 'mlp_c_fc')}"
            elif 'mlp_fc2' in m[1]:
                return f"model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc2', 'mlp_c_proj')}"
            else:
                return f"model_transformer_resblocks_{m[0]}_{m[1].replace('self_attn', 'attn')}"
        return f"transformer_text_model_encoder_layers_{m[0]}_{m[1]}"
    if match(m, r"lora_te2_text_model_encoder_layers_(\d+)_(.+)"):
        if 'mlp_fc1' in m[1]:
            return f"1_model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc1', 'mlp_c_fc')}"
        elif 'mlp_fc2' in m[1]:
            return f"1_model_transformer_resblocks_{m[0]}_{m[1].replace('mlp_fc2', 'mlp_c_proj')}"
        else:
            return f"1_model_transformer_resblocks_{m[0]}_{m[1].replace('self_attn', 'attn')}"
    return key


@dataclass
class NetworkWeights:                                     # network.py:96-101 (sd_module -> engine weight key)
    network_key: str
    sd_key: str
    w: dict
    engine_key: str


class NetworkModuleLora:                                  # network_lora.py:25-80 + network.py:111-216
    def __init__(self, net: "Network", weights: NetworkWeights, shape):
        self.network = net
        self.network_key = weights.network_key
        self.sd_key = weights.sd_key
        self.engine_key = weights.engine_key
        self.shape = tuple(shape)
        w = weights.w
        if "lora_mid.weight" in w:
            raise NotImplementedError(f"{self.network_key}: cp-decomposition (lora_mid) is not implemented")
        if "dora_scale" in w or "bias" in w:
            raise NotImplementedError(f"{self.network_key}: DoRA / bias deltas are not implemented")
        self.up = w["lora_up.weight"]
        self.down = w["lora_down.weight"]
        self.dim = self.down.shape[0]
        self.alpha = w["alpha"].item() if "alpha" in w else None
        self.scale = w["scale"].item() if "scale" in w else None
        rows, cols = self.shape[0], 1
        for d in self.shape[1:]:
            cols *= d
        if self.up.reshape(self.up.shape[0], -1).shape != (rows, self.dim) or self.down.reshape(self.dim, -1).shape[1] != cols:
            raise AssertionError(f"Lora layer {self.network_key}: up {tuple(self.up.shape)} @ down {tuple(self.down.shape)} "
                                 f"does not rebuild a weight of shape {self.shape}")

    def multiplier(self):                                 # network.py:161-165
        if 'transformer' in self.sd_key[:20]:
            return self.network.te_multiplier
        return self.network.unet_multiplier

    def calc_scale(self):                                 # network.py:167-173
        if self.scale is not None:
            return self.scale
        if self.dim is not None and self.alpha is not None:
            return self.alpha / self.dim
        return 1.0


class ModuleTypeLora:                                     # network_lora.py:9-22
    def create_module(self, net, weights: NetworkWeights, shape):
        if all(x in weights.w for x in ["lora_up.weight", "lora_down.weight"]):
            return NetworkModuleLora(net, weights, shape)
        if all(x in weights.w for x in ["lora_A.weight", "lora_B.weight"]):
            w = weights.w.copy()
            weights.w.clear()
            weights.w.update({"lora_up.weight": w["lora_B.weight"], "lora_down.weight": w["lora_A.weight"]})
            return NetworkModuleLora(net, weights, shape)
        return None


module_types = [ModuleTypeLora()]


@dataclass
class Network:                                            # network.py:104-118
    name: str
    te_multiplier: float = 1.0
    unet_multiplier: float = 1.0
    dyn_dim: Optional[int] = None
    modules: Dict[str, NetworkModuleLora] = field(default_factory=dict)
    keys_failed_to_match: Dict[str, str] = field(default_factory=dict)


loaded_networks: List[Network] = []


def assign_network_names_to_compvis_modules(sd_model):    # networks.py:122-147 (UNet part)
    """network layer name -> (engine weight key, shape): `name.replace(".", "_")` of the module path under sd_model.model."""
    mapping = {}
    for key, shape, _ in unet_schema(sd_model.unet_cfg):
        if key.endswith(".weight") and len(shape) in (2, 4):
            mapping[("diffusion_model." + key[:-len(".weight")]).replace(".", "_")] = (key, tuple(shape))
    sd_model.network_layer_mapping = mapping
    return mapping


def load_network(name, sd: dict, sd_model) -> Network:    # networks.py:150-262
    net = Network(name)
    mapping = getattr(sd_model, "network_layer_mapping", None) or assign_network_names_to_compvis_modules(sd_model)
    is_sd2 = 'model_transformer_resblocks' in mapping
    matched: Dict[str, NetworkWeights] = {}
    for key_network, weight in sd.items():
        key_network_without_network_parts, _, network_part = key_network.partition(".")
        key = convert_diffusers_name_to_compvis(key_network_without_network_parts, is_sd2)
        site = mapping.get(key)
        # SDXL loras seem to already have correct compvis keys (:216-218)
        if site is None and "lora_unet" in key_network_without_network_parts:
            key = key_network_without_network_parts.replace("lora_unet", "diffusion_model")
            site = mapping.get(key)
        if site is None:
            net.keys_failed_to_match[key_network] = key
            continue
        if key not in matched:
            matched[key] = NetworkWeights(network_key=key_network, sd_key=key, w={}, engine_key=site[0])
        matched[key].w[network_part] = weight
    for key, weights in matched.items():
        net_module = None
        for nettype in module_types:
            net_module = nettype.create_module(net, weights, mapping[key][1])
            if net_module is not None:
                break
        if net_module is None:
            raise AssertionError(f"Could not find a module type that would accept those keys: {', '.join(weights.w)}")
        net.modules[key] = net_module
    return net


def _merge_on_device(base: torch.Tensor, module: NetworkModuleLora, device) -> torch.Tensor:
    rows = base.shape[0]
    cols = base.numel() // rows
    up = module.up.to(device).contiguous()
    down = module.down.to(device).contiguous()
    if up.dtype not in (torch.float16, torch.float32):
        up = up.float()
    if down.dtype not in (torch.float16, torch.float32):
        down = down.float()
    if module.network.dyn_dim is not None:                # lyco_helpers.py:12-14
        d = module.network.dyn_dim
        up = up.reshape(rows, -1)[:, :d].contiguous()
        down = down.reshape(down.shape[0], -1)[:d, :].contiguous()
    rank = down.shape[0]
    out = torch.empty(base.shape, dtype=torch.float32, device=device)
    scale = float(module.calc_scale()) * float(module.multiplier())
    check(lib.sdmi_lora_merge(ptr(out), ptr(base), dtype_code(base), ptr(up), dtype_code(up), ptr(down), dtype_code(down),
                              rows, cols, rank, scale, stream_ptr()), "sdmi_lora_merge")
    return out


def network_apply_weights(sd_model):
    """networks.py:411-480 for every UNet layer at once: restore from the checkpoint, add each loaded network's delta in
    list order, hand the result to the engine.  Idempotent for an unchanged set of (name, multipliers, dyn_dim)."""
    _lib.require_device()
    wanted = tuple((x.name, x.te_multiplier, x.unet_multiplier, x.dyn_dim) for x in loaded_networks)
    if getattr(sd_model, "network_current_names", ()) == wanted:
        return
    eng = sd_model.engine
    device = torch.device("cuda", eng.device)
    touched = {}
    for net in loaded_networks:
        for key, module in net.modules.items():
            touched.setdefault(module.engine_key, []).append(module)
    previously = getattr(sd_model, "network_touched_keys", set())
    for engine_key in sorted(set(touched) | previously):
        base = sd_model.unet_checkpoint_tensor(engine_key).to(device)
        if base.dtype not in (torch.float16, torch.float32):
            base = base.float()
        w = base.contiguous()
        for module in touched.get(engine_key, []):
            w = _merge_on_device(w, module, device)
        eng.update_unet_weight(engine_key, w)
    sd_model.network_touched_keys = set(touched)
    sd_model.network_current_names = wanted


def load_networks(sd_model, names, state_dicts, te_multipliers=None, unet_multipliers=None, dyn_dims=None):
    """networks.py:300-372 with the network files already read (``state_dicts``: what sd_models.read_state_dict returns for
    each LoRA file), followed by the eager weight rewrite."""
    loaded_networks.clear()
    for i, (name, sd) in enumerate(zip(names, state_dicts)):
        net = load_network(name, sd, sd_model)
        net.te_multiplier = te_multipliers[i] if te_multipliers else 1.0
        net.unet_multiplier = unet_multipliers[i] if unet_multipliers else 1.0
        net.dyn_dim = dyn_dims[i] if dyn_dims else None
        loaded_networks.append(net)
    network_apply_weights(sd_model)
    return loaded_networks
