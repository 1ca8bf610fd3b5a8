"""Op-level wrappers over the C ABI (torch tensors in, torch tensors out; all compute in HIP kernels).

These are the units the parity tests exercise one by one and what the SdOptimization adapter calls.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr, ConvDesc, F16, F32, EP_OUT_F32, EP_GEGLU, EP_NCHW, EP_BIAS_ROW


def _rup(x, m):
    return (x + m - 1) // m * m


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v per head.  q [B,N,H*D], k/v [B,M,H*D] fp16 (last dim contiguous) -> [B,N,H*D] fp16.

    Replaces the math of modules/sd_hijack_optimizations.py:221-281 / hypernetwork.py:382-407 after to_q/to_k/to_v."""
    _lib.require_device()
    assert q.dtype == torch.float16 and k.dtype == torch.float16 and v.dtype == torch.float16
    b, n, c = q.shape
    m = k.shape[1]
    d = c // heads
    q, k, v = [t if t.stride(-1) == 1 and t.stride(0) == t.shape[1] * t.stride(1) else t.contiguous() for t in (q, k, v)]
    out = torch.empty((b, n, c), dtype=torch.float16, device=q.device)
    scale = d ** -0.5 if scale is None else scale
    if heads == 1 and d > 160 and d % 64 == 0:           # the VAE AttnBlock (d = 512): materialised scores, one image at a time
        ws_bytes = lib.sdmi_attention_wide_workspace_bytes(b, n, m, d)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        check(lib.sdmi_attention_wide(ptr(q), ptr(k), ptr(v), ptr(out), b, n, m, d, q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                      float(scale), ptr(ws), ws_bytes, stream_ptr()), "sdmi_attention_wide")
        return out
    ws_bytes = lib.sdmi_attention_workspace_bytes(b, heads, m, d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
    check(lib.sdmi_attention(ptr(q), ptr(k), ptr(v), ptr(out), b, heads, n, m, d, q.stride(1), k.stride(1), v.stride(1),
                             out.stride(1), float(scale), ptr(ws), ws_bytes, stream_ptr()), "sdmi_attention")
    return out


def attention_vt(q, k, vt, heads, m, scale=None, force_generic=False):
    """Same with V pre-transposed: vt [B, H*D, Mpad]."""
    b, n, c = q.shape
    d = c // heads
    out = torch.empty((b, n, c), dtype=torch.float16, device=q.device)
    scale = d ** -0.5 if scale is None else scale
    check(lib.sdmi_attention_vt(ptr(q), ptr(k), ptr(vt), ptr(out), b, heads, n, m, d, q.stride(1), k.stride(1), vt.shape[2],
                                out.stride(1), float(scale), 1 if force_generic else 0, stream_ptr()), "sdmi_attention_vt")
    return out


def pack_conv_weight(w: torch.Tensor, geglu: bool = False, pad: int = 64) -> torch.Tensor:
    """OIHW / [O,I] weight -> packed fp16 [O_pad][taps][I_pad] on the weight's device."""
    _lib.require_device()
    w = w.contiguous()
    if w.dim() == 2:
        w = w[:, :, None, None]
    o, i, kh, kw = w.shape
    o_pad, i_pad = _rup(o, pad), _rup(i, pad)
    out = torch.empty((o_pad, kh * kw, i_pad), dtype=torch.float16, device=w.device)
    check(lib.sdmi_pack_conv_weight(ptr(w), _lib.dtype_code(w), ptr(out), o, i, kh, kw, o_pad, i_pad, 1 if geglu else 0,
                                    stream_ptr()), "pack_conv_weight")
    return out


def pack_bias(b: Optional[torch.Tensor], n_pad: int, geglu: bool = False) -> Optional[torch.Tensor]:
    if b is None:
        return None
    bf = b.float()
    if geglu:
        o = bf.shape[0]
        half = o // 2
        idx = torch.arange(o, device=bf.device)
        g, r = idx // 64, idx % 64
        src = torch.where(r < 32, g * 32 + r, half + g * 32 + (r - 32))
        bf = bf[src]
    out = torch.zeros(n_pad, dtype=torch.float32, device=b.device)
    out[: bf.shape[0]] = bf
    return out


def conv_gemm(a0: torch.Tensor, w_packed: torch.Tensor, *, a1: Optional[torch.Tensor] = None, bias=None, rowbias=None,
              resid=None, taps: int = 9, stride: int = 1, pad: int = 1, up: bool = False, Ho=None, Wo=None,
              geglu: bool = False, out_f32: bool = False, nchw_real: int = 0, bias_row: bool = False, alpha: float = 1.0,
              impl: str = "mfma", transpose: bool = False, wrap: bool = False) -> torch.Tensor:
    """Implicit-GEMM conv / linear on NHWC fp16 tensors a0 [B,Hi,Wi,c0] (+ a1 [B,Hi,Wi,c1]).
    impl: "mfma" (LDS-direct loads), "mfma_reg" (register-staged variant), "generic" (simple HIP kernel)."""
    _lib.require_device()
    assert a0.dtype == torch.float16 and a0.is_contiguous()
    b, hi, wi, c0 = a0.shape
    c1 = a1.shape[3] if a1 is not None else 0
    n = w_packed.shape[0]
    if Ho is None:
        if up:
            Ho, Wo = 2 * hi, 2 * wi
        elif taps == 9:
            p2 = 2 if pad else 1
            Ho, Wo = (hi + p2 - 3) // stride + 1, (wi + p2 - 3) // stride + 1
        else:
            Ho, Wo = hi, wi
    d = ConvDesc()
    d.a0, d.a1, d.w = a0.data_ptr(), (a1.data_ptr() if a1 is not None else None), w_packed.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.rowbias = rowbias.data_ptr() if rowbias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.c0, d.c1, d.lda0, d.lda1 = c0, c1, c0, c1
    d.B, d.Hi, d.Wi, d.Ho, d.Wo = b, hi, wi, Ho, Wo
    d.taps, d.stride, d.pad, d.up = taps, stride, pad if taps == 9 else 0, 1 if up else 0
    d.N = n
    flags = 0
    n_out = n
    if geglu:
        flags |= EP_GEGLU
        n_out = n // 2
    if bias_row:
        flags |= EP_BIAS_ROW
    if wrap:
        flags |= _lib.EP_WRAP
    if nchw_real:
        flags |= EP_NCHW
        out = torch.empty((b, nchw_real, Ho, Wo), dtype=torch.float32, device=a0.device)
        d.n_real = nchw_real
    elif out_f32:
        flags |= EP_OUT_F32
        out = torch.empty((b, Ho, Wo, n_out), dtype=torch.float32, device=a0.device)
    elif transpose:
        flags |= _lib.EP_TRANSPOSE
        out = torch.empty((b, n_out, Ho * Wo), dtype=torch.float16, device=a0.device)       # out^T per image (V^T for attention)
    else:
        out = torch.empty((b, Ho, Wo, n_out), dtype=torch.float16, device=a0.device)
    d.out = out.data_ptr()
    d.ldo = Ho * Wo if transpose else n_out
    d.ldr = resid.shape[-1] if resid is not None else 0
    d.flags = flags
    d.alpha = alpha
    d.batch = 1
    d.force_generic = {"mfma": 0, "generic": 1, "mfma_reg": 2}[impl]
    wsb = lib.sdmi_conv_splitk_workspace_bytes(b * Ho * Wo, n, taps * (c0 + c1), 1)
    if wsb and not geglu and not nchw_real:
        ws = torch.empty(wsb, dtype=torch.uint8, device=a0.device)       # lets small-M / large-K shapes use split-K
        d.splitk_workspace, d.splitk_workspace_bytes = ws.data_ptr(), wsb
    check(lib.sdmi_conv_gemm(C.byref(d), stream_ptr()), "sdmi_conv_gemm")
    return out


def groupnorm(x0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, x1: Optional[torch.Tensor] = None,
              groups: int = 32, eps: float = 1e-5, silu: bool = True) -> torch.Tensor:
    """GroupNorm(+SiLU) over NHWC fp16 [B,H,W,c0] (optionally channel-concatenated with x1) -> [B,H,W,c0+c1] fp16."""
    _lib.require_device()
    b, h, w, c0 = x0.shape
    c1 = x1.shape[3] if x1 is not None else 0
    out = torch.empty((b, h, w, c0 + c1), dtype=torch.float16, device=x0.device)
    wsb = lib.sdmi_groupnorm_workspace_bytes(b, h * w, groups)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x0.device)
    check(lib.sdmi_groupnorm(ptr(x0), ptr(x1), c0, c1, ptr(gamma.float().contiguous()), ptr(beta.float().contiguous()),
                             ptr(out), b, h * w, groups, float(eps), 1 if silu else 0, ptr(ws), wsb, stream_ptr()), "sdmi_groupnorm")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _lib.require_device()
    c = x.shape[-1]
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib.sdmi_layernorm(ptr(x), ptr(gamma.float().contiguous()), ptr(beta.float().contiguous()), ptr(out),
                             x.numel() // c, c, float(eps), stream_ptr()), "sdmi_layernorm")
    return out


def rowchain_ff_pack(w1: torch.Tensor, b1, w2: torch.Tensor) -> torch.Tensor:
    """Packed operand stream of the fused feed-forward chain (csrc/rowchain.hip): w1 [2*hidden, C] = ff.net.0.proj.weight (value rows,
    then gate rows), b1 [2*hidden] or None, w2 [C, hidden] = ff.net.2.weight."""
    _lib.require_device()
    hidden, c = w2.shape[1], w2.shape[0]
    w1, w2 = w1.half().contiguous(), w2.half().contiguous()
    b1 = b1.float().contiguous() if b1 is not None else None
    packs = torch.empty(int(lib.sdmi_rowchain_ff_pack_bytes(c, hidden)), dtype=torch.uint8, device=w1.device)
    check(lib.sdmi_rowchain_ff_pack(ptr(w1), ptr(b1) if b1 is not None else None, ptr(w2), ptr(packs), c, hidden, stream_ptr()),
          "sdmi_rowchain_ff_pack")
    return packs


def rowchain_ff(x: torch.Tensor, gamma, beta, packs: torch.Tensor, b2, hidden: int, eps: float = 1e-5) -> torch.Tensor:
    """x + ff(LayerNorm(x)) of a BasicTransformerBlock (GEGLU feed-forward) as one launch; x [rows, C] fp16, rows % 128 == 0."""
    _lib.require_device()
    c = x.shape[-1]
    x = x.contiguous()
    out = torch.empty_like(x)
    b2 = b2.float().contiguous() if b2 is not None else None
    check(lib.sdmi_rowchain_ff(ptr(x), ptr(out), ptr(gamma.float().contiguous()), ptr(beta.float().contiguous()), ptr(packs),
                               ptr(b2) if b2 is not None else None, x.numel() // c, c, hidden, float(eps), stream_ptr()), "sdmi_rowchain_ff")
    return out


def philox_randn(shape, seed: int, offset: int, device) -> torch.Tensor:
    """One draw of rng_philox.Generator(seed) at ``offset`` (modules/rng_philox.py:84-102), generated on the GPU."""
    _lib.require_device()
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    check(lib.sdmi_philox_randn(ptr(out), out.numel(), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint32(offset), stream_ptr()),
          "sdmi_philox_randn")
    return out


def lincomb(out: torch.Tensor, terms, coefs) -> torch.Tensor:
    """out = sum_k coefs[k] * terms[k] on fp32 tensors of one shape (out may be one of the terms)."""
    _lib.require_device()
    n = len(terms)
    assert 1 <= n <= 6 and len(coefs) == n
    for t in terms:
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == out.numel()
    assert out.dtype == torch.float32 and out.is_contiguous()
    tarr = (C.c_void_p * n)(*[t.data_ptr() for t in terms])
    carr = (C.c_float * n)(*[float(c) for c in coefs])
    check(lib.sdmi_lincomb(ptr(out), tarr, carr, n, out.numel(), stream_ptr()), "sdmi_lincomb")
    return out


def mask_blend(x: torch.Tensor, init: torch.Tensor, mask: torch.Tensor, nmask: torch.Tensor) -> torch.Tensor:
    """x = init*mask + nmask*x in place (modules/sd_samplers_cfg_denoiser.py:206-209); mask/nmask are broadcast to x's shape."""
    _lib.require_device()
    m = mask.to(x.device, torch.float32).expand_as(x).contiguous()
    nm = nmask.to(x.device, torch.float32).expand_as(x).contiguous()
    init = init.to(x.device, torch.float32).expand_as(x).contiguous()
    check(lib.sdmi_mask_blend(ptr(x), ptr(init), ptr(m), ptr(nm), x.numel(), stream_ptr()), "sdmi_mask_blend")
    return x


def latent_resize(x: torch.Tensor, size, mode: str = "bilinear", antialias: bool = False) -> torch.Tensor:
    """F.interpolate(x, size=size, mode=mode, antialias=antialias) for fp32 NCHW latents (modules/processing.py:1392; antialias with
    the bilinear / bicubic modes only, as in torch)."""
    _lib.require_device()
    x = x.float().contiguous()
    b, c, hi, wi = x.shape
    ho, wo = int(size[0]), int(size[1])
    out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=x.device)
    code = {"nearest": 0, "nearest-exact": 1, "bilinear": 2, "bicubic": 3}[mode]
    if antialias:
        if mode not in ("bilinear", "bicubic"):
            raise ValueError("antialias is defined for the bilinear and bicubic modes")     # torch raises the same way
        code += 2
    check(lib.sdmi_latent_resize(ptr(x), ptr(out), b * c, hi, wi, ho, wo, code, stream_ptr()), "sdmi_latent_resize")
    return out


def image_to_u8(img: torch.Tensor) -> torch.Tensor:
    """fp32 NCHW in [-1,1] -> uint8 NHWC (modules/processing.py:1004-1005, 1034-1035)."""
    _lib.require_device()
    img = img.float().contiguous()
    b, c, h, w = img.shape
    out = torch.empty((b, h, w, c), dtype=torch.uint8, device=img.device)
    check(lib.sdmi_image_to_u8(ptr(img), ptr(out), b, c, h, w, stream_ptr()), "sdmi_image_to_u8")
    return out
