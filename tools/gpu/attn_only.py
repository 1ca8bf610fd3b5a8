#!/usr/bin/env python3
"""Only the level-0 self-attention launch of the C1 job (B 16, H 8, N = M = 4096, d 40), 20 times — the target of the PMC passes of
tools/gpu/r03_attn_pmc.sh.  (Does not import oracle/.)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
ops = importlib.import_module(f"{PKG}.ops")
importlib.import_module(f"{PKG}._lib").require_device()
g = torch.Generator().manual_seed(1)
b, h, n, d = 16, 8, 4096, 40
q, k = torch.randn(b, n, h * d, generator=g).half().cuda(), torch.randn(b, n, h * d, generator=g).half().cuda()
vt = torch.randn(b, h * d, n, generator=g).half().cuda()
for _ in range(20):
    ops.attention_vt(q, k, vt, h, n)
torch.cuda.synchronize()
print("done")
