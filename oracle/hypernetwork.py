"""fp32 restatement of the webui's hypernetworks (oracle; tests only).

Follows /root/reference/modules/hypernetworks/hypernetwork.py: ``HypernetworkModule`` (:25-113: Sequential of Linear / activation /
LayerNorm / Dropout built from ``layer_structure``; ``forward = x + linear(x) * multiplier``), ``apply_hypernetworks`` (:358-379: every
loaded network transforms the attention context separately for the K and V paths, picking its modules by the context's feature
width) and ``attention_CrossAttention_forward`` (:382-407), which ``oracle.unet.CrossAttention.forward`` already restates without
hypernetworks.  Pinned by tests/golden/hypernetwork.npz, which make_golden produces by exec'ing the reference's own class text.
"""
from __future__ import annotations

import torch
import torch.nn as nn

ACT = {"linear": nn.Identity, "relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "elu": nn.ELU, "swish": nn.Hardswish, "tanh": nn.Tanh,
       "sigmoid": nn.Sigmoid, "silu": nn.SiLU, "gelu": nn.GELU, "mish": nn.Mish, "relu6": nn.ReLU6, "selu": nn.SELU,
       "softplus": nn.Softplus, "softsign": nn.Softsign, "hardtanh": nn.Hardtanh, "hardsigmoid": nn.Hardsigmoid, "hardswish": nn.Hardswish}


class HypernetworkModule(nn.Module):
    def __init__(self, dim, state_dict, layer_structure=(1, 2, 1), activation_func=None, add_layer_norm=False, activate_output=False,
                 dropout_structure=None):
        super().__init__()
        self.multiplier = 1.0
        linears = []
        for i in range(len(layer_structure) - 1):
            linears.append(nn.Linear(int(dim * layer_structure[i]), int(dim * layer_structure[i + 1])))
            if activation_func == "linear" or activation_func is None or (i >= len(layer_structure) - 2 and not activate_output):
                pass
            else:
                linears.append(ACT[activation_func]())
            if add_layer_norm:
                linears.append(nn.LayerNorm(int(dim * layer_structure[i + 1])))
            if dropout_structure is not None and dropout_structure[i + 1] > 0:
                linears.append(nn.Dropout(p=dropout_structure[i + 1]))
        self.linear = nn.Sequential(*linears)
        if state_dict is not None:
            self.load_state_dict({k: v.float() for k, v in state_dict.items()})
        self.eval().requires_grad_(False)

    def forward(self, x):
        return x + self.linear(x) * self.multiplier


class Hypernetwork:
    def __init__(self, state_dict: dict, multiplier: float = 1.0):
        ls = state_dict.get('layer_structure', [1, 2, 1])
        act = state_dict.get('activation_func', None)
        ln = state_dict.get('is_layer_norm', False)
        ao = state_dict.get('activate_output', True)
        ds = state_dict.get('dropout_structure', None)
        self.layers = {}
        for size, sd in state_dict.items():
            if type(size) == int:
                self.layers[size] = tuple(HypernetworkModule(size, sd[j], ls, act, ln, ao, ds) for j in (0, 1))
        for pair in self.layers.values():
            for m in pair:
                m.multiplier = multiplier


def apply_hypernetworks(hypernetworks, context):
    context_k = context_v = context
    for hn in hypernetworks:
        layers = hn.layers.get(context_k.shape[2], None)
        if layers is None:
            continue
        context_k, context_v = layers[0](context_k.float()), layers[1](context_v.float())
    return context_k, context_v
