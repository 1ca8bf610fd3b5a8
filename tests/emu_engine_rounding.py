"""Prices the `residual_fp32` engine option on the CPU before a kernel is written (test infrastructure).

The fp32 oracle UNet is run with the ENGINE's rounding pattern emulated — fp32 arithmetic, and a binary16 rounding exactly where an
engine kernel stores an fp16 tensor (csrc/engine.cpp: GroupNorm+SiLU output, conv / linear outputs after their fused fp32 epilogue
(bias, time embedding, residual, GEGLU product), LayerNorm output, q / k / v, the attention probabilities before P V, the attention
output) — once with the CARRIED residual stream stored in fp16 like every other activation (today's engine), once with only that
stream kept in fp32 (what `residual_fp32` would do: ResBlock output skip(x) + h, conv_in / down / upsample outputs, proj_in output,
the three x += adds of a transformer block, proj_out + x_in; GEMMs that take the stream as their A operand read an fp16 copy).

    python tests/emu_engine_rounding.py [--rows 2] [--tiny]

prints rel-L2 against the unrounded fp32 oracle for both patterns.  tests/fp16_emu.py is the sibling that emulates the REFERENCE's
fp16-autocast pattern (torch rounds more often: after bias / emb / residual adds).
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import unet as ou  # noqa: E402


def r16(t):
    return t.half().float()


class Emu:
    """Walks an oracle UNetModel with the engine's store pattern.  ``stream_fp32``: the carried tensors keep fp32."""

    def __init__(self, net: ou.UNetModel, stream_fp32: bool, skip_fp32: bool = False, h1_fp32: bool = False, xin_fp32: bool = False):
        self.net, self.stream_fp32, self.skip_fp32, self.h1_fp32 = net, stream_fp32, skip_fp32, h1_fp32
        self.xin_fp32 = xin_fp32      # conv_in takes the latent as a (hi, lo) pair in its zero-padded input channels: x is not rounded

    def carried(self, t):
        return t if self.stream_fp32 else r16(t)

    def gn_silu(self, gn, x, silu=True):                      # gn_apply_kernel: statistics + normalise in fp32, one fp16 store
        y = F.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps)
        return r16(F.silu(y) if silu else y)

    def res(self, blk: ou.ResBlock, x, emb):
        a = self.gn_silu(blk.in_layers[0], x)
        e = blk.emb_layers[1](F.silu(emb))                    # small_linear_kernel: fp32 in, fp32 out
        h1 = blk.in_layers[2](a) + e[:, :, None, None]        # conv1 epilogue: bias + per-image embedding, one store
        h1 = h1 if self.h1_fp32 else r16(h1)                  # (read by the second GroupNorm only: never a matrix-core operand)
        b = self.gn_silu(blk.out_layers[0], h1)
        if isinstance(blk.skip_connection, torch.nn.Identity):
            skip = x
        else:                                                 # its own 1x1 GEMM over the fp16 copy of the stream
            skip = blk.skip_connection(r16(x))
            skip = skip if self.skip_fp32 else r16(skip)
        return self.carried(blk.out_layers[3](b) + skip)      # conv2 epilogue: bias + residual, one store

    def attn(self, at: ou.CrossAttention, xn, context):
        h = at.heads
        ctx = xn if context is None else r16(context)
        q, k, v = r16(at.to_q(xn)), r16(at.to_k(ctx)), r16(at.to_v(ctx))
        b, n, _ = q.shape
        split = lambda t: t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)
        q, k, v = split(q), split(k), split(v)
        p = torch.softmax(torch.einsum('bid,bjd->bij', q, k) * at.scale, dim=-1)          # fp32 scores and softmax
        # the flash kernel rounds the UNNORMALISED probabilities exp(s - m) to fp16 before P V and divides by the fp32 row sum after:
        # emulate with the row maximum as reference point (m = the running maximum ends at the row maximum)
        pmax = p.max(dim=-1, keepdim=True).values
        pu = r16(p / pmax)
        out = r16(torch.einsum('bij,bjd->bid', pu, v) / pu.sum(-1, keepdim=True))
        out = out.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
        return at.to_out[0](out)                              # the caller's epilogue adds the residual before the store

    def tblock(self, tb: ou.BasicTransformerBlock, x, context):
        ln = lambda m, t: r16(F.layer_norm(t, m.normalized_shape, m.weight, m.bias, m.eps))
        x = self.carried(self.attn(tb.attn1, ln(tb.norm1, x), None) + x)
        x = self.carried(self.attn(tb.attn2, ln(tb.norm2, x), context) + x)
        g = tb.ff.net[0]
        a, gate = g.proj(ln(tb.norm3, x)).chunk(2, dim=-1)
        x = self.carried(tb.ff.net[2](r16(a * F.gelu(gate))) + x)                    # GEGLU product in the ff1 epilogue, one store
        return x

    def st(self, s: ou.SpatialTransformer, x, context):
        b, c, h, w = x.shape
        x_in = x
        n = self.gn_silu(s.norm, x, silu=False)
        if not s.use_linear:
            t = s.proj_in(n).permute(0, 2, 3, 1).reshape(b, h * w, -1)
        else:
            t = s.proj_in(n.permute(0, 2, 3, 1).reshape(b, h * w, -1))
        t = self.carried(t)
        for blk in s.transformer_blocks:
            t = self.tblock(blk, t, context)
        t16 = r16(t)                                          # proj_out takes the stream as its A operand: the fp16 copy
        if s.use_linear:
            o = s.proj_out(t16).view(b, h, w, -1).permute(0, 3, 1, 2)
        else:
            o = s.proj_out(t16.view(b, h, w, -1).permute(0, 3, 1, 2))
        return self.carried(o + x_in)

    def seq(self, mods, h, emb, context):
        for layer in mods:
            if isinstance(layer, ou.ResBlock):
                h = self.res(layer, h, emb)
            elif isinstance(layer, ou.SpatialTransformer):
                h = self.st(layer, h, context)
            elif isinstance(layer, ou.Downsample):
                h = self.carried(layer.op(r16(h)))
            elif isinstance(layer, ou.Upsample):
                h = self.carried(layer.conv(F.interpolate(r16(h), scale_factor=2, mode="nearest")))
            else:                                             # conv_in: the input x arrives as fp32 and is stored fp16 by the layout kernel
                h = self.carried(layer(h if self.xin_fp32 else r16(h)))
        return h

    def __call__(self, x, timesteps, context):
        net = self.net
        emb = net.time_embed(ou.timestep_embedding(timesteps, net.cfg.model_channels))     # fp32 MLP in the engine
        hs, h = [], x
        for m in net.input_blocks:
            h = self.seq(m, h, emb, context)
            hs.append(h)
        h = self.seq(net.middle_block, h, emb, context)
        for m in net.output_blocks:
            h = self.seq(m, torch.cat([h, hs.pop()], dim=1), emb, context)
        return net.out[2](self.gn_silu(net.out[0], h))        # conv_out stores fp32


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--first", type=int, default=0, help="skip the first N store patterns")
    ap.add_argument("--save", default="", help="write the emulated outputs (fp16 stream / all-fp32 non-operand tensors) to this .npz: "
                                               "tests/golden/emu_engine_c1.npz is read by tests/test_gpu_c1_parity.py")
    args = ap.parse_args()
    schema = importlib.import_module("stable-diffusion-webui_amd.schema")
    from helpers import seeded, usable_cpus
    torch.set_num_threads(usable_cpus(32))
    if args.tiny:
        ucfg, ocfg, hw, cdim = schema.tiny_unet(), ou.tiny_config(), 16, 64
    else:
        ucfg, ocfg, hw, cdim = schema.sd15_unet(), ou.sd15_config(), 64, 768
    sd = schema.synthetic_state_dict(ucfg, None, dtype=torch.float16)
    net = ou.build_unet(ocfg, sd)
    B = 16
    x = seeded((B, 4, hw, hw), 101)[:args.rows]               # the inputs of tests/test_gpu_c1_parity.py::test_c1_unet_cfg_forward_16_rows_vs_oracle
    t = torch.linspace(999.0, 1.0, B)[:args.rows]
    ctx = seeded((B, 77, cdim), 102).half().float()[:args.rows]
    saved = []
    with torch.no_grad():
        t0 = time.time()
        ref = net(x, t, ctx)
        print(f"fp32 oracle: {time.time() - t0:.1f} s for {args.rows} rows", flush=True)
        for name, kw in (("engine pattern, fp16 residual stream (today)", dict(stream_fp32=False)),
                         ("engine pattern, fp32 residual stream (residual_fp32)", dict(stream_fp32=True)),
                         ("engine pattern, fp32 residual stream + fp32 skip_connection output", dict(stream_fp32=True, skip_fp32=True)),
                         ("... + fp32 conv1 output (every tensor that is not a matrix-core operand in fp32)", dict(stream_fp32=True, skip_fp32=True, h1_fp32=True)),
                         ("... + the latent into conv_in as a (hi, lo) pair", dict(stream_fp32=True, skip_fp32=True, h1_fp32=True, xin_fp32=True)))[args.first:]:
            got = Emu(net, **kw)(x, t, ctx)
            print(f"{name}: rel-L2 {rel_l2(got, ref):.3e}   per row {[f'{rel_l2(got[i], ref[i]):.2e}' for i in range(args.rows)]}", flush=True)
            saved.append((got, rel_l2(got, ref)))
    if args.save:
        import numpy as np
        assert args.first == 0, "--save needs every store pattern"
        np.savez_compressed(args.save, rows=args.rows, fp16_stream=saved[0][0].numpy(), fp32_stream=saved[1][0].numpy(),
                            fp32_stream_skip=saved[2][0].numpy(), fp32_all_non_operand=saved[3][0].numpy(),
                            fp32_all_non_operand_xin=saved[4][0].numpy(), fp32_oracle=ref.numpy(),
                            rel_l2_vs_fp32_oracle=np.array([e for _, e in saved]))
        print("wrote", args.save)


if __name__ == "__main__":
    main()
