"""Hypernetworks for the engine UNet — mirror of modules/hypernetworks/hypernetwork.py (the parts inference touches).

A hypernetwork file (``torch.save`` of a dict: integer keys = feature widths -> (K module state dict, V module state dict), plus
'layer_structure', 'activation_func', 'is_layer_norm', 'activate_output', 'dropout_structure' ...; :243-300) describes, per
attention width, two small MLPs.  ``HypernetworkModule`` (:25-113) builds ``torch.nn.Sequential`` of Linear / activation / LayerNorm /
Dropout in that order from ``layer_structure``; at inference ``forward`` is ``x + linear(x) * multiplier`` and
``apply_hypernetworks`` (:358-379) chains every loaded network over the attention context, separately for the K and the V path.

Here the module description is turned into the engine's op list (``sdmi_unet_hypernet_*``); the MLPs then run inside the engine's
attention layers.  ``load_hypernetworks(sd_model, names_or_dicts, multipliers)`` keeps the reference's reuse-if-already-loaded
behaviour (:324-345) in ``shared.loaded_hypernetworks``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, shared
from ._lib import lib, check

# activation_dict of :26-35 (lower-cased torch.nn.modules.activation class names) -> engine codes (hn_act in csrc/elementwise.hip)
ACTIVATIONS = {"relu": 1, "leakyrelu": 2, "elu": 3, "swish": 4, "hardswish": 4, "tanh": 5, "sigmoid": 6, "silu": 7, "gelu": 8, "mish": 9,
               "relu6": 10, "selu": 11, "softplus": 12, "softsign": 13, "hardtanh": 14, "hardsigmoid": 15}


def parse_dropout_structure(layer_structure, use_dropout, last_layer_dropout):       # :127-141
    if layer_structure is None:
        layer_structure = [1, 2, 1]
    if not use_dropout:
        return [0] * len(layer_structure)
    dropout_values = [0]
    dropout_values.extend([0.3] * (len(layer_structure) - 3))
    dropout_values.append(0.3 if last_layer_dropout else 0)
    dropout_values.append(0)
    return dropout_values


def _fix_old_state_dict(sd: dict) -> dict:                                            # :90-102
    sd = dict(sd)
    for fr, to in (('linear1.bias', 'linear.0.bias'), ('linear1.weight', 'linear.0.weight'),
                   ('linear2.bias', 'linear.1.bias'), ('linear2.weight', 'linear.1.weight')):
        if fr in sd:
            sd[to] = sd.pop(fr)
    return sd


def module_ops(dim: int, state_dict: dict, layer_structure, activation_func, add_layer_norm, activate_output, dropout_structure) -> list:
    """The op sequence of HypernetworkModule.__init__ (:38-70) with the weights of ``state_dict`` attached: a list of
    ("linear", W [out, in], b) | ("act", code) | ("ln", gamma, beta).  Sequential indices count Dropout layers too (they hold no
    weights but shift the indices of the layers after them)."""
    assert layer_structure is not None and layer_structure[0] == 1 and layer_structure[-1] == 1, "layer_structure must start and end with 1"
    sd = _fix_old_state_dict(state_dict)
    ops, idx = [], 0
    for i in range(len(layer_structure) - 1):
        n_in, n_out = int(dim * layer_structure[i]), int(dim * layer_structure[i + 1])
        w, b = sd[f"linear.{idx}.weight"], sd[f"linear.{idx}.bias"]
        if tuple(w.shape) != (n_out, n_in):
            raise ValueError(f"hypernetwork layer {idx}: weight {tuple(w.shape)}, expected {(n_out, n_in)}")
        ops.append(("linear", w, b))
        idx += 1
        if activation_func == "linear" or activation_func is None or (i >= len(layer_structure) - 2 and not activate_output):
            pass
        elif activation_func in ACTIVATIONS:
            ops.append(("act", ACTIVATIONS[activation_func]))
            idx += 1
        else:
            raise NotImplementedError(f"hypernetwork uses an activation function the engine does not implement: {activation_func}")
        if add_layer_norm:
            ops.append(("ln", sd[f"linear.{idx}.weight"], sd[f"linear.{idx}.bias"]))
            idx += 1
        if dropout_structure is not None and dropout_structure[i + 1] > 0:
            idx += 1                                           # torch.nn.Dropout: identity at inference
    return ops


class Hypernetwork:
    def __init__(self, name=None):
        self.name, self.filename = name, None
        self.layers: Dict[int, Tuple[list, list]] = {}       # width -> (K ops, V ops)
        self.multiplier = 1.0
        self.layer_structure = self.activation_func = None
        self.add_layer_norm = False
        self.activate_output = True

    def load_state(self, state_dict: dict):
        """:246-300 from the unpickled dict on."""
        self.layer_structure = state_dict.get('layer_structure', [1, 2, 1])
        self.activation_func = state_dict.get('activation_func', None)
        self.add_layer_norm = state_dict.get('is_layer_norm', False)
        dropout_structure = state_dict.get('dropout_structure', None)
        use_dropout = True if dropout_structure is not None and any(dropout_structure) else state_dict.get('use_dropout', False)
        self.activate_output = state_dict.get('activate_output', True)
        last_layer_dropout = state_dict.get('last_layer_dropout', False)
        if dropout_structure is None:
            dropout_structure = parse_dropout_structure(self.layer_structure, use_dropout, last_layer_dropout)
        for size, sd in state_dict.items():
            if type(size) == int:
                self.layers[size] = tuple(module_ops(size, sd[j], self.layer_structure, self.activation_func, self.add_layer_norm,
                                                     self.activate_output, dropout_structure) for j in (0, 1))
        self.name = state_dict.get('name', self.name)
        return self

    def load(self, filename: str):
        self.filename = filename
        if self.name is None:
            self.name = os.path.splitext(os.path.basename(filename))[0]
        # weights_only: a hypernetwork .pt holds tensors, OrderedDicts, ints, strs, lists and bools only — never unpickle arbitrary objects
        # from a user-supplied file (the reference routes torch.load through modules/safe.py's restricted unpickler)
        return self.load_state(torch.load(filename, map_location='cpu', weights_only=True))

    def set_multiplier(self, multiplier):
        self.multiplier = float(multiplier)
        return self


if not hasattr(shared, "loaded_hypernetworks"):
    shared.loaded_hypernetworks = []
if not hasattr(shared, "hypernetworks"):
    shared.hypernetworks = {}                                 # name -> path (list_hypernetworks, :302-309)


def load_hypernetwork(name_or_state) -> Optional[Hypernetwork]:                        # :312-324
    if isinstance(name_or_state, dict):
        return Hypernetwork(name_or_state.get("name")).load_state(name_or_state)
    path = shared.hypernetworks.get(name_or_state, None)
    if path is None:
        return None
    return Hypernetwork().load(path)


def _send(eng_handle, dim, which, ops):
    for op in ops:
        if op[0] == "linear":
            w = op[1].detach().to(torch.float32).contiguous()
            b = op[2].detach().to(torch.float32).contiguous()
            check(lib.sdmi_unet_hypernet_linear(eng_handle, dim, which, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), _lib.F32,
                                                w.shape[0], w.shape[1], 1 if w.is_cuda else 0), "hypernet_linear")
        elif op[0] == "act":
            check(lib.sdmi_unet_hypernet_act(eng_handle, dim, which, int(op[1])), "hypernet_act")
        else:
            g, b = op[1].detach().to(torch.float32).contiguous(), op[2].detach().to(torch.float32).contiguous()
            check(lib.sdmi_unet_hypernet_layernorm(eng_handle, dim, which, C.c_void_p(g.data_ptr()), C.c_void_p(b.data_ptr()), _lib.F32,
                                                   g.numel(), 1 if g.is_cuda else 0), "hypernet_layernorm")


def load_hypernetworks(sd_model, names, multipliers=None) -> List[Hypernetwork]:
    """:327-345, then hands the loaded set to the engine of ``sd_model`` (in list order = application order).  ``names`` entries
    are names registered in shared.hypernetworks or already-unpickled state dicts."""
    _lib.require_device()
    already = {h.name: h for h in shared.loaded_hypernetworks}
    shared.loaded_hypernetworks.clear()
    for i, name in enumerate(names):
        key = name.get("name") if isinstance(name, dict) else name
        hn = already.get(key) if key is not None and key in already else load_hypernetwork(name)
        if hn is None:
            continue
        hn.set_multiplier(multipliers[i] if multipliers else 1.0)
        shared.loaded_hypernetworks.append(hn)
    eng = sd_model.engine
    check(lib.sdmi_unet_hypernet_clear(eng.handle), "hypernet_clear")
    for hn in shared.loaded_hypernetworks:
        check(lib.sdmi_unet_hypernet_begin(eng.handle, float(hn.multiplier)), "hypernet_begin")
        for dim, (k_ops, v_ops) in hn.layers.items():
            _send(eng.handle, int(dim), 0, k_ops)
            _send(eng.handle, int(dim), 1, v_ops)
    eng._ctx_key = None
    eng.weights_version = getattr(eng, "weights_version", 0) + 1      # cached cross-attention K / V depend on the hypernetworks
    return shared.loaded_hypernetworks
