#!/usr/bin/env python3
"""Numerics of a Winograd F(2x2, 3x3) convolution with fp16 operand storage against the direct fp16-operand convolution the engine runs
(price before building).  CPU only, seconds.

Both take the same fp16 activations and weights and accumulate in fp32 (as the MFMA does); the Winograd form additionally stores the
TRANSFORMED operands in fp16 — U = G g G^T per (cout, cin), V = B^T d B per 4x4 input tile — because those are what the matrix cores
would multiply.  Reference: the fp32 convolution of the same fp16 inputs.  Output rounded to fp16 in both forms.

    python tools/cpu/winograd_numerics.py
"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
r16 = lambda t: t.half().double()


def winograd(x, w, fp16_transforms=True):
    """x [B, C, H, W] (H, W even), w [O, C, 3, 3]; padding 1.  Returns [B, O, H, W] in float64."""
    B, C, H, W = x.shape
    O = w.shape[0]
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)                       # [O, C, 4, 4]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                          # [B, C, H/2, W/2, 4, 4]
    V = torch.einsum('ij,bcyxjk,lk->bcyxil', BT, tiles, BT)
    if fp16_transforms:
        U, V = r16(U), r16(V)
    M = torch.einsum('ocil,bcyxil->boyxil', U, V)                       # fp32-class accumulation over C (float64 here: no extra error)
    Y = torch.einsum('ij,boyxjk,lk->boyxil', AT, M, AT)                 # [B, O, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, O, H, W)


def main():
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"{'layer':34s} {'direct fp16 out':>16s} {'winograd, fp16 U/V':>20s} {'ratio':>7s} {'winograd, U/V exact':>20s}")
    for name, (B, C, O, H) in {"level 2  1280 -> 1280, 16x16": (1, 1280, 1280, 16), "level 1   640 ->  640, 32x32": (1, 640, 640, 32),
                               "level 1  1920 ->  640, 32x32": (1, 1920, 640, 32), "level 3  1280 -> 1280,  8x8": (2, 1280, 1280, 8)}.items():
        x = r16(torch.randn(B, C, H, H))                                # post GroupNorm + SiLU scale ~ 1
        w = r16(torch.randn(O, C, 3, 3) * (9 * C) ** -0.5)
        ref = F.conv2d(x, w, padding=1)                                 # float64 "fp32 reference" of the same fp16 operands
        direct = r16(ref)                                               # direct conv: exact products, fp32 accumulate, one fp16 store
        wg = r16(winograd(x, w, True))
        wg_exact = r16(winograd(x, w, False))
        e_d, e_w, e_x = rel(direct, ref), rel(wg, ref), rel(wg_exact, ref)
        print(f"{name:34s} {e_d:16.3e} {e_w:20.3e} {e_w / e_d:7.2f} {e_x:20.3e}")


if __name__ == "__main__":
    main()
