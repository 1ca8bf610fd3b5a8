"""LoRA for the fused UNet — host-side mirror of /root/reference/extensions-builtin/Lora (networks.py, network.py,
network_lora.py), same names and argument meaning where the reference has them.

The reference rewrites ``module.weight`` lazily inside ``torch.nn.Linear/Conv2d.forward`` (networks.py:411-480, patches at
lora_patches.py:7-18) — hooks a fused UNet never runs.  Here ``load_networks`` + ``network_apply_weights`` do the same
rewrite eagerly: for every UNet layer a loaded network touches, start from the checkpoint weight (the "backup",
networks.py:423-432), add each network's delta ``up @ down * alpha/dim * multiplier`` on the GPU (sdmi_lora_merge) and
re-pack that one layer inside the engine (sdmi_unet_update_weight).  Layers no network touches any more are restored.
Text-encoder keys are reported in ``keys_failed_to_match`` (the text encoder is outside the engine).  Module types, in the
reference's dispatch order (networks.py:26-36): LoRA / LoCon (incl. cp-decomposition, dyn_dim, the inpainting conv_in padding),
LoHa, IA3, LoKr, full diff, GLoRA, each optionally weight-decomposed (DoRA); norm, OFT / BOFT and bias deltas raise
NotImplementedError.  Every full-size operation is a HIP kernel (sdmi_lora_merge / sdmi_weight_*); torch only moves data."""
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


class NetworkModule:                                      # network.py:111-216 (the parts a weight rewrite needs)
    kind = None

    def __init__(self, net: "Network", weights: NetworkWeights, shape):
        self.network = net
        self.network_key = weights.network_key
        self.sd_key = weights.sd_key
        self.engine_key = weights.engine_key
        self.shape = tuple(shape)
        self.w = weights.w
        w = weights.w
        self.dim = None
        self.alpha = w["alpha"].item() if "alpha" in w else None
        self.scale = w["scale"].item() if "scale" in w else None
        self.dora_scale = w.get("dora_scale", None)
        # network.py:154, 196-199: an extra dense "bias" entry with as many elements as the weight delta, added to updown BEFORE the
        # scale, the weight decomposition and the multiplier (finalize_updown)
        self.bias = w.get("bias")

    def ex_bias(self, device):
        """Bias delta of this module (network.py:196-216 finalize_updown: ex_bias * multiplier), or None."""
        return None

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

    def add_delta(self, base, scale, device):
        """base + scale * updown as a fresh fp32 tensor of the layer's shape (``base``: fp32, contiguous, on the device)."""
        raise NotImplementedError()


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def _mm(a, b, base=None, scale=1.0):
    """base + scale * (a @ b) for fp32 device matrices through sdmi_lora_merge (zero base when omitted)."""
    rows, rank = a.shape
    cols = b.shape[1]
    assert b.shape[0] == rank, (tuple(a.shape), tuple(b.shape))
    out = torch.empty((rows, cols), dtype=torch.float32, device=a.device)
    base = torch.zeros_like(out) if base is None else base
    check(lib.sdmi_lora_merge(ptr(out), ptr(base), _lib.F32, ptr(a.contiguous()), _lib.F32, ptr(b.contiguous()), _lib.F32,
                              rows, cols, rank, float(scale), stream_ptr()), "sdmi_lora_merge")
    return out


def _cp(t, wa, wb):
    """lyco_helpers.make_weight_cp (:4-6) as two matrix products: out[r, r', k, l] = sum_ij t[i,j,k,l] wa[i,r] wb[j,r']."""
    i, j, k, l = t.shape
    rp = wb.shape[1]
    temp = _mm(t.permute(0, 2, 3, 1).reshape(i * k * l, j).contiguous(), wb)                    # [(i,k,l), r']
    temp = temp.reshape(i, k, l, rp).permute(0, 3, 1, 2).reshape(i, rp * k * l).contiguous()   # [i, (r',k,l)]
    return _mm(wa.t().contiguous(), temp).reshape(wa.shape[1], rp, k, l)


class NetworkModuleLora(NetworkModule):                   # network_lora.py:25-80
    kind = "lora"

    def __init__(self, net, weights, shape):
        super().__init__(net, weights, shape)
        w = weights.w
        self.up = w["lora_up.weight"]
        self.down = w["lora_down.weight"]
        self.mid = w.get("lora_mid.weight")
        self.dim = self.down.shape[0]
        rows, cols = self.shape[0], 1
        for d in self.shape[1:]:
            cols *= d
        self.pad_inpaint = False
        if self.mid is None:
            got_cols = self.down.reshape(self.dim, -1).shape[1]
            if len(self.shape) == 4 and self.shape[1] == 9 and got_cols * 9 == cols * 4:
                self.pad_inpaint = True                   # inpainting conv_in: the 4-channel delta is zero-padded to 9 (networks.py:468-470)
            elif self.up.reshape(self.up.shape[0], -1).shape != (rows, self.dim) or got_cols != cols:
                raise AssertionError(f"Lora layer {self.network_key}: up {tuple(self.up.shape)} @ down {tuple(self.down.shape)} "
                                     f"does not rebuild a weight of shape {self.shape}")

    def add_delta(self, base, scale, device):
        rows = self.shape[0]
        up = _f32(self.up, device).reshape(rows, -1)
        down = _f32(self.down, device)
        down = down.reshape(down.shape[0], -1)
        if self.mid is not None:                          # cp-decomposition (lyco_helpers.py:18-21)
            mid = _f32(self.mid, device)
            n, m, k, l = mid.shape
            t = _mm(mid.permute(0, 2, 3, 1).reshape(n * k * l, m).contiguous(), down)            # [(n,k,l), j]
            j = down.shape[1]
            t = t.reshape(n, k, l, j).permute(0, 3, 1, 2).reshape(n, j * k * l).contiguous()
            return _mm(up, t, base.reshape(rows, -1), scale).reshape(self.shape)
        if self.network.dyn_dim is not None:              # lyco_helpers.py:12-14
            d = self.network.dyn_dim
            up, down = up[:, :d].contiguous(), down[:d, :].contiguous()
        if self.pad_inpaint:
            kk = self.shape[2] * self.shape[3]
            delta = _mm(up, down, None, scale).reshape(rows, 4, self.shape[2], self.shape[3])
            delta = torch.nn.functional.pad(delta, (0, 0, 0, 0, 0, 5)).contiguous()
            from . import ops
            return ops.lincomb(torch.empty_like(base), [base, delta], [1.0, 1.0])
        return _mm(up, down, base.reshape(rows, -1), scale).reshape(self.shape)


class NetworkModuleHada(NetworkModule):                   # network_hada.py:14-55 (LoHa)
    kind = "hada"

    def __init__(self, net, weights, shape):
        super().__init__(net, weights, shape)
        self.dim = weights.w["hada_w1_b"].shape[0]

    def add_delta(self, base, scale, device):
        w = self.w
        w1a, w1b, w2a, w2b = (_f32(w[k], device) for k in ("hada_w1_a", "hada_w1_b", "hada_w2_a", "hada_w2_b"))
        t1, t2 = w.get("hada_t1"), w.get("hada_t2")
        rebuild = lambda a, b: _mm(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1))
        u1 = _cp(_f32(t1, device), w1a, w1b) if t1 is not None else rebuild(w1a, w1b)
        u2 = _cp(_f32(t2, device), w2a, w2b) if t2 is not None else rebuild(w2a, w2b)
        if u1.numel() != base.numel() or u2.numel() != base.numel():
            raise AssertionError(f"LoHa layer {self.network_key} does not rebuild a weight of shape {self.shape}")
        out = torch.empty_like(base)
        check(lib.sdmi_weight_hadamard(ptr(out), ptr(base), ptr(u1.contiguous()), ptr(u2.contiguous()), float(scale), base.numel(),
                                       stream_ptr()), "sdmi_weight_hadamard")
        return out


class NetworkModuleLokr(NetworkModule):                   # network_lokr.py:26-64
    kind = "lokr"

    def __init__(self, net, weights, shape):
        super().__init__(net, weights, shape)
        w = weights.w
        if w.get("lokr_w1_b") is not None:
            self.dim = w["lokr_w1_b"].shape[0]
        if w.get("lokr_w2_b") is not None:
            self.dim = w["lokr_w2_b"].shape[0]

    def add_delta(self, base, scale, device):
        w = self.w
        w1 = _f32(w["lokr_w1"], device) if w.get("lokr_w1") is not None else _mm(_f32(w["lokr_w1_a"], device), _f32(w["lokr_w1_b"], device))
        if w.get("lokr_w2") is not None:
            w2 = _f32(w["lokr_w2"], device)
        elif w.get("lokr_t2") is None:
            w2 = _mm(_f32(w["lokr_w2_a"], device), _f32(w["lokr_w2_b"], device))
        else:
            w2 = _cp(_f32(w["lokr_t2"], device), _f32(w["lokr_w2_a"], device), _f32(w["lokr_w2_b"], device))
        r1, c1 = w1.shape
        r2, c2 = w2.shape[0], w2.shape[1]
        k = w2.numel() // (r2 * c2)
        if r1 * r2 * c1 * c2 * k != base.numel() or r1 * r2 != self.shape[0]:
            raise AssertionError(f"LoKr layer {self.network_key}: kron({tuple(w1.shape)}, {tuple(w2.shape)}) is not a weight of shape {self.shape}")
        out = torch.empty_like(base)
        check(lib.sdmi_weight_kron(ptr(out), ptr(base), ptr(w1.contiguous()), ptr(w2.contiguous()), r1, c1, r2, c2, k, float(scale),
                                   stream_ptr()), "sdmi_weight_kron")
        return out


class NetworkModuleGLora(NetworkModule):                  # network_glora.py:12-33
    kind = "glora"

    def add_delta(self, base, scale, device):
        if len(self.shape) != 2:
            raise NotImplementedError(f"GLoRA layer {self.network_key}: only linear layers")
        w1a, w1b, w2a, w2b = (_f32(self.w[k], device) for k in ("a1.weight", "b1.weight", "a2.weight", "b2.weight"))
        out = _mm(_mm(base, w2a), w1a, base, scale)       # (W @ a2) @ a1 reads the CURRENT weight, like the reference's orig_weight
        return _mm(w2b, w1b, out, scale)


class NetworkModuleIa3(NetworkModule):                    # network_ia3.py:13-30
    kind = "ia3"

    def add_delta(self, base, scale, device):
        if len(self.shape) != 2:
            raise NotImplementedError(f"IA3 layer {self.network_key}: only linear layers")
        v = _f32(self.w["weight"], device).reshape(-1)
        on_input = bool(self.w["on_input"].item())
        rows, cols = self.shape
        if v.numel() != (cols if on_input else rows):
            raise AssertionError(f"IA3 layer {self.network_key}: {v.numel()} scales for a weight of shape {self.shape}")
        out = torch.empty_like(base)
        check(lib.sdmi_weight_ia3(ptr(out), ptr(base), ptr(v), rows, cols, int(on_input), float(scale), stream_ptr()), "sdmi_weight_ia3")
        return out


class NetworkModuleFull(NetworkModule):                   # network_full.py:12-27
    kind = "full"

    def ex_bias(self, device):                            # network_full.py:20-24
        db = self.w.get("diff_b")
        return None if db is None else _f32(db, device).reshape(-1) * float(self.multiplier())

    def add_delta(self, base, scale, device):
        diff = _f32(self.w["diff"], device)
        if diff.numel() != base.numel():
            raise AssertionError(f"full-diff layer {self.network_key}: diff {tuple(diff.shape)} for a weight of shape {self.shape}")
        from . import ops
        return ops.lincomb(torch.empty_like(base), [base, diff.reshape(base.shape).contiguous()], [1.0, float(scale)])


class NetworkModuleNorm(NetworkModule):                   # network_norm.py:13-28
    """w_norm / b_norm: differences of a GroupNorm / LayerNorm layer's gain and shift (the engine keeps them as fp32 vectors beside
    its fused norm kernels; rewritten through sdmi_unet_update_vector)."""
    kind = "norm"

    def add_delta(self, base, scale, device):
        wn = _f32(self.w["w_norm"], device)
        if wn.numel() != base.numel():
            raise AssertionError(f"norm layer {self.network_key}: w_norm {tuple(wn.shape)} for a parameter of shape {self.shape}")
        from . import ops
        return ops.lincomb(torch.empty_like(base), [base, wn.reshape(base.shape).contiguous()], [1.0, float(scale)])

    def ex_bias(self, device):
        bn = self.w.get("b_norm")
        return None if bn is None else _f32(bn, device).reshape(-1) * float(self.multiplier())


class NetworkModuleOFT(NetworkModule):                    # network_oft.py:14-118
    """Orthogonal fine-tuning: the weight's output rows are rotated block-wise, updown = R W - W.  kohya / new LyCORIS "oft_blocks"
    [k, n, n] hold the generators B: Q = B - B^T, optionally norm-clamped to alpha * out_dim (COFT), R = (I + Q)(I - Q)^-1 (Cayley);
    old LyCORIS "oft_diag" holds R itself; 4-D blocks [m, k, n, n] are BOFT's m butterfly factors.  The rotation matrices are a few
    small blocks built once per network load on the host (fp32 torch, like the reference); rotating the layer's rows is the same
    block-by-block matrix product either way and runs through sdmi_lora_merge on the device."""
    kind = "oft"

    def __init__(self, net, weights, shape):
        super().__init__(net, weights, shape)
        w = weights.w
        self.scale = 1.0                                   # :23
        self.out_dim = self.shape[0]
        if "oft_blocks" in w:
            self.blocks, self.is_r = w["oft_blocks"], False
            dim = self.blocks.shape[0]
        else:
            self.blocks, self.is_r = w["oft_diag"], True
            dim = self.blocks.shape[1]
        self.is_boft = self.blocks.dim() == 4
        self.rescale = w.get("rescale")
        self.num_blocks, self.block_size = dim, self.out_dim // dim
        self.constraint = (0.0 if self.alpha is None else float(self.alpha)) * self.out_dim
        if self.is_r:
            self.constraint = None
            self.block_size, self.num_blocks = dim, self.out_dim // dim
        elif self.is_boft:
            self.boft_m, self.num_blocks, self.block_size = self.blocks.shape[0], self.blocks.shape[1], self.blocks.shape[2]

    def calc_scale(self):
        return 1.0

    def rotations(self) -> torch.Tensor:
        blocks = self.blocks.detach().float().cpu()
        if self.is_r:
            return blocks
        eye = torch.eye(self.block_size)
        q = blocks - blocks.transpose(-1, -2)
        if self.constraint != 0:
            norm_q = torch.norm(q.flatten())
            new_norm = torch.clamp(norm_q, max=torch.tensor(self.constraint))
            q = q * ((new_norm + 1e-8) / (norm_q + 1e-8))
        return torch.matmul(eye + q, (eye - q).float().inverse())

    def _rotate_rows(self, r_blocks: torch.Tensor, rows: torch.Tensor, device) -> torch.Tensor:
        """rows [k*n, cols] -> for every block k: out[k] = R[k]^T-applied rows, i.e. out[k, m, :] = sum_n R[k, n, m] rows[k, n, :]."""
        k, n = r_blocks.shape[0], r_blocks.shape[1]
        cols = rows.shape[1]
        out = torch.empty_like(rows)
        rt = r_blocks.transpose(-1, -2).contiguous().to(device)
        for i in range(k):
            out[i * n:(i + 1) * n] = _mm(rt[i], rows[i * n:(i + 1) * n].contiguous()).reshape(n, cols)
        return out

    def add_delta(self, base, scale, device):
        rmat = self.rotations()
        flat = base.reshape(self.out_dim, -1).contiguous()
        if not self.is_boft:
            merged = self._rotate_rows(rmat, flat, device)
        else:
            b = self.block_size
            r_b = b // 2
            merged = flat
            for i in range(self.boft_m):                   # butterfly factor i acts on rows permuted with stride 2^i * r_b (:92-103)
                kk = (2 ** i) * r_b
                c = self.out_dim // (2 * kk)
                idx = torch.arange(self.out_dim, device=device).reshape(c, 2, kk).permute(0, 2, 1).reshape(-1)   # "(c g k) -> (c k g)"
                perm = merged[idx].contiguous()
                bi = rmat[i]                               # [blocks, b, b]: out[d] = bi[d] @ rows[d]
                rot = torch.empty_like(perm)
                bd = bi.contiguous().to(device)
                for d in range(bi.shape[0]):
                    rot[d * b:(d + 1) * b] = _mm(bd[d], perm[d * b:(d + 1) * b].contiguous()).reshape(b, -1)
                inv = torch.empty_like(idx)
                inv[idx] = torch.arange(self.out_dim, device=device)
                merged = rot[inv].contiguous()
        if self.rescale is not None:
            merged = merged * _f32(self.rescale, device).reshape(-1, 1)
        from . import ops
        merged = merged.reshape(base.shape).contiguous()
        # base + scale * (merged - base), accumulated left to right like finalize_updown's updown * (calc_scale * multiplier)
        return ops.lincomb(torch.empty_like(base), [base, merged, base], [1.0, float(scale), -float(scale)])


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


class _ModuleTypeByKeys:
    def __init__(self, cls, accepts):
        self.cls, self.accepts = cls, accepts

    def create_module(self, net, weights: NetworkWeights, shape):
        return self.cls(net, weights, shape) if self.accepts(weights.w) else None


def _unsupported(name):
    def make(net, weights, shape):
        raise NotImplementedError(f"{weights.network_key}: LyCORIS module type {name} is not implemented")
    return make


module_types = [                                          # the order of networks.py:26-36
    ModuleTypeLora(),
    _ModuleTypeByKeys(NetworkModuleHada, lambda w: all(x in w for x in ["hada_w1_a", "hada_w1_b", "hada_w2_a", "hada_w2_b"])),
    _ModuleTypeByKeys(NetworkModuleIa3, lambda w: "weight" in w),
    _ModuleTypeByKeys(NetworkModuleLokr, lambda w: ("lokr_w1" in w or ("lokr_w1_a" in w and "lokr_w1_b" in w))
                      and ("lokr_w2" in w or ("lokr_w2_a" in w and "lokr_w2_b" in w))),
    _ModuleTypeByKeys(NetworkModuleFull, lambda w: "diff" in w),
    _ModuleTypeByKeys(NetworkModuleNorm, lambda w: all(x in w for x in ["w_norm", "b_norm"])),
    _ModuleTypeByKeys(NetworkModuleGLora, lambda w: all(x in w for x in ["a1.weight", "a2.weight", "alpha", "b1.weight", "b2.weight"])),
    _ModuleTypeByKeys(NetworkModuleOFT, lambda w: "oft_blocks" in w or "oft_diag" in w),
]


@dataclass
class Network:                                            # network.py:104-118
    name: str
    te_multiplier: float = 1.0
    unet_multiplier: float = 1.0
    dyn_dim: Optional[int] = None
    modules: Dict[str, NetworkModule] = field(default_factory=dict)
    keys_failed_to_match: Dict[str, str] = field(default_factory=dict)


loaded_networks: List[Network] = []


def assign_network_names_to_compvis_modules(sd_model):    # networks.py:122-147 (UNet part)
    """network layer name -> (engine weight key, shape): `name.replace(".", "_")` of the module path under sd_model.model."""
    mapping = {}
    for key, shape, kind in unet_schema(sd_model.unet_cfg):
        if key.endswith(".weight") and (len(shape) in (2, 4) or kind == "g"):      # conv / linear weights and norm gains
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


def _merge_on_device(base: torch.Tensor, module: NetworkModule, device) -> torch.Tensor:
    """One network module's contribution to one layer, network.py:196-216 finalize_updown folded in:
    W + updown * calc_scale * multiplier, or with a ``dora_scale`` the weight-decomposed form of network.py:175-194."""
    base = _f32(base, device)
    scale, mult = float(module.calc_scale()), float(module.multiplier())
    bias = None
    if module.bias is not None:                            # network.py:196-199: updown.reshape(bias.shape) + bias, then back
        bias = _f32(module.bias, device).reshape(-1)
        if bias.numel() != base.numel():
            raise AssertionError(f"{module.network_key}: 'bias' has {bias.numel()} elements for a weight of {base.numel()}")
        bias = bias.reshape(base.shape).contiguous()
    from . import ops
    if module.dora_scale is None:
        out = module.add_delta(base, scale * mult, device)
        return out if bias is None else ops.lincomb(torch.empty_like(out), [out, bias], [1.0, scale * mult])
    if module.kind in ("ia3", "glora", "oft", "norm"):
        raise NotImplementedError(f"{module.network_key}: DoRA on {module.kind} modules is not implemented")
    delta = module.add_delta(torch.zeros_like(base), scale, device)
    if bias is not None:
        delta = ops.lincomb(torch.empty_like(delta), [delta, bias], [1.0, scale])
    rows, cin = base.shape[0], base.shape[1]
    k = base.numel() // (rows * cin)
    dora_scale = _f32(module.dora_scale, device).reshape(-1)
    if dora_scale.numel() != cin:
        raise AssertionError(f"{module.network_key}: dora_scale has {dora_scale.numel()} entries for {cin} input channels")
    out = torch.empty_like(base)
    check(lib.sdmi_weight_dora(ptr(out), ptr(base), ptr(delta.contiguous()), ptr(dora_scale), rows, cin, k, mult, stream_ptr()), "sdmi_weight_dora")
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
    bias_touched = set()
    for engine_key in sorted(set(touched) | previously):
        base = sd_model.unet_checkpoint_tensor(engine_key).to(device)
        if base.dtype not in (torch.float16, torch.float32):
            base = base.float()
        w = base.contiguous()
        ex = None
        for module in touched.get(engine_key, []):
            w = _merge_on_device(w, module, device)
            eb = module.ex_bias(device)                    # networks.py:455-462: bias += ex_bias
            if eb is not None:
                ex = eb if ex is None else ex + eb
        if w.dim() == 1:
            eng.update_unet_vector(engine_key, w)          # a norm layer's gain
        else:
            eng.update_unet_weight(engine_key, w)
        bias_key = engine_key[:-len("weight")] + "bias"
        if ex is not None:
            try:
                b0 = sd_model.unet_checkpoint_tensor(bias_key)
            except KeyError:
                raise NotImplementedError(f"{engine_key}: the network adds a bias to a layer that has none") from None
            from . import ops
            b0 = _f32(b0, device)
            eng.update_unet_vector(bias_key, ops.lincomb(torch.empty_like(b0), [b0, ex.contiguous()], [1.0, 1.0]))
            bias_touched.add(bias_key)
    for bias_key in sorted(getattr(sd_model, "network_touched_biases", set()) - bias_touched):     # restore biases no network touches any more
        eng.update_unet_vector(bias_key, _f32(sd_model.unet_checkpoint_tensor(bias_key), device))
    sd_model.network_touched_keys = set(touched)
    sd_model.network_touched_biases = bias_touched
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
