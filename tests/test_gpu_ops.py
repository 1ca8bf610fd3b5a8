"""GPU parity tests, op level: every HIP kernel behind the C ABI against the fp32 CPU oracle / a plain torch fp32
reference of the same op, on seeded inputs.  Tolerances are stated per test (fp16 storage, fp32 accumulate)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2, seeded

pytestmark = pytest.mark.gpu


def sub(name):
    return importlib.import_module("stable-diffusion-webui_amd." + name)


@pytest.fixture(scope="module")
def dev():
    lib = sub("_lib")
    lib.require_device()          # fail loudly: no fallback
    return torch.device("cuda", 0)


ATTN_TAU_DEFAULT = 8          # csrc/attention.hip g_attn_tau
ATTN_FOLD_MIN_M_DEFAULT = 1024     # csrc/attention.hip g_attn_fold_min_m


def h(t):
    return t.half().float()       # the fp16-rounded value the kernel sees


# ------------------------------------------------------------------------------------------------------------
# Philox
# ------------------------------------------------------------------------------------------------------------
def test_philox_bit_exact_vs_reference_fixture(dev, golden_dir):
    ops = sub("ops")
    z = np.load(os.path.join(golden_dir, "philox.npz"))
    total = mism = 0
    for key in sorted(k for k in z.files if k.startswith("c")):
        _, seed, draw = key.split("_")
        got = ops.philox_randn(z[key].shape, int(seed[4:]), int(draw[4:]), dev).cpu().numpy()
        total += got.size
        mism += int((got != z[key]).sum())
        np.testing.assert_allclose(got, z[key], rtol=0, atol=2.4e-7 * 8)       # never more than a few ulp
    # device libm double log/sin vs the host's: allow a vanishing fraction of 1-ulp double-rounding differences
    assert mism <= max(1, total // 20000), f"{mism}/{total} values differ from the reference bit pattern"


def test_image_rng_matches_oracle(dev):
    from oracle import rng as orng
    r = sub("rng").ImageRNG((4, 8, 8), [1000, 1001, 2 ** 32 + 7], device=dev)
    o = orng.ImageRNG((4, 8, 8), [1000, 1001, 2 ** 32 + 7])
    for _ in range(3):
        np.testing.assert_allclose(r.next().cpu().numpy(), o.next().numpy(), rtol=0, atol=1e-6)


# ------------------------------------------------------------------------------------------------------------
# implicit-GEMM conv / linear
# ------------------------------------------------------------------------------------------------------------
def _conv_ref(x_nhwc, w, bias, stride=1, pad=1, up=False, asym=False):
    x = x_nhwc.permute(0, 3, 1, 2)
    if up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    if asym:
        x = F.pad(x, (0, 1, 0, 1))
        y = F.conv2d(x, w, bias, stride=stride, padding=0)
    else:
        y = F.conv2d(x, w, bias, stride=stride, padding=pad if w.shape[-1] == 3 else 0)
    return y.permute(0, 2, 3, 1).contiguous()


IMPLS = ["mfma", "mfma_reg", "generic"]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("case", [
    dict(B=2, H=12, W=10, cin=64, cout=128, k=3),                       # M = 240: partial 128-row tile
    dict(B=1, H=16, W=16, cin=128, cout=320, k=3),                      # N = 320 -> 256x64 tiles (5 column tiles)
    dict(B=2, H=9, W=7, cin=192, cout=64, k=3, stride=2),               # stride 2, odd sizes
    dict(B=1, H=8, W=8, cin=64, cout=64, k=3, up=True),                 # fused nearest x2 upsample
    dict(B=1, H=9, W=9, cin=64, cout=64, k=3, stride=2, asym=True),     # VAE-encoder downsample (pad right/bottom)
    dict(B=3, H=5, W=5, cin=320, cout=640, k=1),                        # 1x1
    dict(B=1, H=64, W=64, cin=64, cout=128, k=3),                       # M = 4096: full 128x128 tiles, XCD remap
    dict(B=2, H=32, W=32, cin=128, cout=64, k=3),                       # 256x64 config, many tiles
])
def test_conv_gemm_vs_torch(dev, impl, case):
    ops = sub("ops")
    B, H, W, cin, cout, k = case["B"], case["H"], case["W"], case["cin"], case["cout"], case["k"]
    stride, up, asym = case.get("stride", 1), case.get("up", False), case.get("asym", False)
    x = seeded((B, H, W, cin), 1)
    w = seeded((cout, cin, k, k), 2, scale=(cin * k * k) ** -0.5)
    b = seeded((cout,), 3, scale=0.1)
    ref = _conv_ref(h(x), h(w), b, stride=stride, up=up, asym=asym)
    wp = ops.pack_conv_weight(w.half().to(dev))
    bp = ops.pack_bias(b.to(dev), wp.shape[0])
    got = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=k * k, stride=stride, pad=0 if asym else 1, up=up, impl=impl)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    # fp32 accumulate, fp16 store: |err| <= 2^-11 |y| + accumulation noise
    assert rel_l2(got.float().cpu(), ref) < 6e-4, (impl, case)
    assert float((got.float().cpu() - ref).abs().max()) < 4e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("impl", IMPLS)
def test_conv_gemm_epilogues(dev, impl):
    """bias + per-image rowbias (ResBlock emb add) + residual; fp32 output; NCHW store of a 4-channel conv_out."""
    ops = sub("ops")
    B, H, W, cin, cout = 2, 8, 8, 64, 128
    x, w = seeded((B, H, W, cin), 1), seeded((cout, cin, 3, 3), 2, scale=(cin * 9) ** -0.5)
    b, rb, res = seeded((cout,), 3, 0.1), seeded((B, cout), 4), seeded((B, H, W, cout), 5)
    ref = _conv_ref(h(x), h(w), b) + rb[:, None, None, :] + h(res)
    wp = ops.pack_conv_weight(w.half().to(dev))
    got = ops.conv_gemm(x.half().to(dev), wp, bias=b.to(dev), rowbias=rb.to(dev).contiguous(), resid=res.half().to(dev), impl=impl)
    assert rel_l2(got.float().cpu(), ref) < 6e-4
    got32 = ops.conv_gemm(x.half().to(dev), wp, bias=b.to(dev), out_f32=True, impl=impl)
    assert rel_l2(got32.cpu(), _conv_ref(h(x), h(w), b)) < 2e-5          # fp32 store: only accumulation-order noise
    # conv_out-like: 4 real output channels, NCHW fp32
    w4, b4 = seeded((4, cin, 3, 3), 6, scale=(cin * 9) ** -0.5), seeded((4,), 7, 0.1)
    w4p = ops.pack_conv_weight(w4.half().to(dev))
    assert w4p.shape[0] == 64
    got4 = ops.conv_gemm(x.half().to(dev), w4p, bias=ops.pack_bias(b4.to(dev), 64), nchw_real=4, impl=impl)
    ref4 = _conv_ref(h(x), h(w4), b4).permute(0, 3, 1, 2)
    assert got4.shape == (B, 4, H, W) and rel_l2(got4.cpu(), ref4) < 2e-5


@pytest.mark.parametrize("impl", IMPLS)
def test_conv_gemm_two_sources_equals_concat(dev, impl):
    """skip-connection concat elided: reading (h, skip) as two sources == conv over torch.cat([h, skip], C)."""
    ops = sub("ops")
    B, H, W, c0, c1, cout = 1, 8, 8, 128, 64, 128
    x0, x1 = seeded((B, H, W, c0), 1), seeded((B, H, W, c1), 2)
    w = seeded((cout, c0 + c1, 3, 3), 3, scale=((c0 + c1) * 9) ** -0.5)
    ref = _conv_ref(h(torch.cat([x0, x1], dim=3)), h(w), None)
    wp = ops.pack_conv_weight(w.half().to(dev))
    got = ops.conv_gemm(x0.half().to(dev), wp, a1=x1.half().to(dev), impl=impl)
    assert rel_l2(got.float().cpu(), ref) < 6e-4


@pytest.mark.parametrize("impl", IMPLS)
def test_linear_geglu_epilogue(dev, impl):
    """FeedForward's GEGLU: out = value * gelu(gate) fused into the first GEMM (ldm GEGLU; erf GELU)."""
    ops = sub("ops")
    rows, c = 200, 64
    x = seeded((1, rows, 1, c), 1)
    w, b = seeded((8 * c, c), 2, scale=c ** -0.5), seeded((8 * c,), 3, 0.1)
    y = F.linear(h(x).reshape(rows, c), h(w), b)
    a, g = y.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    wp = ops.pack_conv_weight(w.half().to(dev), geglu=True)
    bp = ops.pack_bias(b.to(dev), 8 * c, geglu=True)
    got = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=1, geglu=True, impl=impl)
    assert got.shape[-1] == 4 * c
    assert rel_l2(got.float().cpu().reshape(rows, 4 * c), ref) < 8e-4


@pytest.mark.parametrize("cfg,split", [(-1, 0), (0, 3), (5, 4), (8, 2), (3, 6), (4, 2)])
def test_conv_gemm_split_k_is_exact_and_deterministic(dev, cfg, split):
    """Deep-level shape (small M, K = 11520): split-K slices summed in slice order by the reduce pass, with the full
    epilogue (bias + per-image emb add + residual).  Same result as torch, and bit-identical run to run."""
    ops, lib = sub("ops"), sub("_lib")
    B, H, W, cin, cout = 2, 8, 8, 1280, 1280
    x, w = seeded((B, H, W, cin), 1), seeded((cout, cin, 3, 3), 2, scale=(cin * 9) ** -0.5)
    b, rb, res = seeded((cout,), 3, 0.1), seeded((B, cout), 4), seeded((B, H, W, cout), 5)
    ref = _conv_ref(h(x), h(w), b) + rb[:, None, None, :] + h(res)
    wp = ops.pack_conv_weight(w.half().to(dev))
    lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", cfg)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", split))
    try:
        args = dict(bias=b.to(dev), rowbias=rb.to(dev).contiguous(), resid=res.half().to(dev))
        got = ops.conv_gemm(x.half().to(dev), wp, **args)
        again = ops.conv_gemm(x.half().to(dev), wp, **args)
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", 0))
    assert rel_l2(got.float().cpu(), ref) < 6e-4, (cfg, split)
    assert torch.equal(got, again)


PHASE_CASES = [
    # (cfg, B, H, W, c0, c1, cout, taps, stride, up, split)   cfg 5 = 256x320, 4 = 256x256, 8 = 128x320
    (5, 2, 16, 16, 64, 0, 320, 1, 1, False, 0),        # K = 64: a single K tile (prologue only)
    (5, 2, 16, 16, 128, 0, 320, 1, 1, False, 0),       # K = 128: two K tiles (no steady state)
    (5, 1, 24, 20, 192, 0, 640, 1, 1, False, 0),       # K = 192, M = 480: ragged last row tile, two column tiles
    (5, 2, 16, 16, 320, 0, 320, 9, 1, False, 0),       # 3x3 with zero padding, 45 K tiles
    (5, 2, 17, 15, 128, 0, 320, 9, 2, False, 0),       # stride 2, odd sizes
    (5, 1, 12, 12, 128, 0, 320, 9, 1, True, 0),        # fused nearest x2 upsample
    (5, 1, 16, 16, 128, 64, 320, 9, 1, False, 0),      # two sources (skip concat), source switch inside the K loop
    (4, 2, 16, 16, 128, 0, 256, 9, 1, False, 0),       # 256x256 tile
    (4, 1, 20, 20, 64, 0, 512, 1, 1, False, 0),        # 256x256, K = 64
    (8, 2, 16, 16, 192, 0, 320, 9, 1, False, 0),       # 128x320 tile, two phases
    (8, 1, 10, 10, 64, 0, 640, 1, 1, False, 0),        # 128x320, single K tile, ragged M
    (5, 2, 8, 8, 1280, 0, 1280, 9, 1, False, 4),       # split-K slices (each slice runs its own prologue / tail)
    (8, 2, 8, 8, 1280, 0, 1280, 9, 1, False, 3),
    (8, 2, 16, 16, 128, 0, 320, 1, 1, False, 0),       # 128x320: K = 128 (two K tiles)
    (8, 1, 12, 12, 128, 64, 640, 9, 1, False, 0),      # 128x320: two sources
    (8, 2, 17, 15, 128, 0, 320, 9, 2, False, 0),       # 128x320: stride 2
    (8, 1, 12, 12, 128, 0, 320, 9, 1, True, 0),        # 128x320: upsample
]


@pytest.mark.parametrize("case", PHASE_CASES)
def test_pingpong_gemm_is_bit_identical_to_two_stage_kernel(dev, case):
    """The ping-pong kernel (default for the 256-row tiles) refills LDS piecewise ~1.5 K tiles ahead with counted vmcnt
    waits and runs its two wave groups one barrier apart; a staging race would show up as wrong tiles.  Same MFMA order as the two-stage kernel => the outputs must be identical bit for bit (and equal to torch
    within the usual tolerance), on every path of the gather (padding, stride, upsample, concat) and for K loops too short
    to reach the steady state."""
    ops, lib = sub("ops"), sub("_lib")
    cfg, B, H, W, c0, c1, cout, taps, stride, up, split = case
    k = 3 if taps == 9 else 1
    x0 = seeded((B, H, W, c0), 1)
    x1 = seeded((B, H, W, c1), 2) if c1 else None
    w = seeded((cout, c0 + c1, k, k), 3, scale=((c0 + c1) * k * k) ** -0.5)
    b, res_seed = seeded((cout,), 4, 0.1), 5
    xin = torch.cat([x0, x1], dim=3) if c1 else x0
    ref = _conv_ref(h(xin), h(w), b, stride=stride, up=up)
    res = seeded(tuple(ref.shape), res_seed)
    ref = ref + h(res)
    wp = ops.pack_conv_weight(w.half().to(dev))
    args = dict(a1=None if x1 is None else x1.half().to(dev), bias=ops.pack_bias(b.to(dev), wp.shape[0]),
                resid=res.half().to(dev), taps=taps, stride=stride, up=up)
    outs = {}
    try:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", cfg)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", split))
        lib.check(lib.lib.sdmi_debug_set(b"conv_korder", 0))          # tap-major on both sides (the default row-shared walk sums in another order)
        pp = 4 if cfg == 8 else 3                          # 4 also routes the 128x320 tile to the ping-pong kernel
        for pipe in (0, pp, pp, pp):
            lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", pipe))
            outs.setdefault(3 if pipe else 0, []).append(ops.conv_gemm(x0.half().to(dev), wp, **args))
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", 0))
        lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", -1)); lib.check(lib.lib.sdmi_debug_set(b"conv_korder", -1))
    torch.cuda.synchronize()
    assert rel_l2(outs[0][0].float().cpu(), ref) < 6e-4, case
    for o in outs[3]:
        assert torch.equal(o, outs[0][0]), case


@pytest.mark.parametrize("case", [c for c in PHASE_CASES if c[7] == 9 and not c[9]])
def test_channel_block_major_k_order_pingpong_equals_two_stage(dev, case):
    """conv_korder = 1 (channel blocks outer, the 9 taps inner: 31 % less HBM traffic, selectable — tap-major is the default again,
    profiles/r02_conv_korder.md): the ping-pong kernel addresses a tap as ONE per-row pointer + a block-uniform tap offset under a
    9-bit validity mask carried in address bits 48..56 (computed in the M section); the two-stage kernel rebuilds the address per
    tile.  Same K-tile order => same bits, on padding / stride 2 / two sources / split-K; and equal to torch within the usual bound."""
    ops, lib = sub("ops"), sub("_lib")
    cfg, B, H, W, c0, c1, cout, taps, stride, up, split = case
    x0 = seeded((B, H, W, c0), 1)
    x1 = seeded((B, H, W, c1), 2) if c1 else None
    w = seeded((cout, c0 + c1, 3, 3), 3, scale=((c0 + c1) * 9) ** -0.5)
    b = seeded((cout,), 4, 0.1)
    xin = torch.cat([x0, x1], dim=3) if c1 else x0
    ref = _conv_ref(h(xin), h(w), b, stride=stride)
    wp = ops.pack_conv_weight(w.half().to(dev))
    args = dict(a1=None if x1 is None else x1.half().to(dev), bias=ops.pack_bias(b.to(dev), wp.shape[0]), taps=9, stride=stride)
    outs = []
    try:
        lib.check(lib.lib.sdmi_debug_set(b"conv_korder", 1))
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", cfg)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", split))
        for pipe in (0, 4 if cfg == 8 else 3):
            lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", pipe))
            outs.append(ops.conv_gemm(x0.half().to(dev), wp, **args))
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", 0))
        lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", -1)); lib.check(lib.lib.sdmi_debug_set(b"conv_korder", -1))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), case
    assert rel_l2(outs[1].float().cpu(), ref) < 6e-4, case


DX_CASES = [
    # (cfg, B, H, W, c0, c1, cout, split, resid)      row-shared walk (conv_korder 2, the default): stride-1 3x3, tile = whole image rows
    (5, 2, 64, 64, 128, 0, 320, 0, True),              # 256x320: 4 rows of 64 per tile (level 0), 6 K tiles = 2 groups
    (5, 2, 64, 64, 64, 0, 320, 0, False),              # one 64-channel block per kernel row: 3 groups, the shortest loop
    (5, 1, 32, 32, 192, 128, 320, 0, True),            # two sources, source switch between groups, 8 rows of 32 per tile
    (5, 1, 24, 16, 320, 0, 320, 0, True),              # W = 16: 16 image rows per tile (a guard row between every fragment), ragged M = 384
    (5, 2, 16, 16, 1280, 0, 1280, 4, True),            # split-K: slices of whole (dy, channel block) groups
    (4, 2, 64, 64, 128, 0, 256, 0, False),             # 256x256 (VAE widths)
    (4, 1, 16, 256, 128, 0, 256, 0, True),             # W = 256 = BM: one image row per tile, the halves of a row belong to different wave groups
    (8, 2, 32, 32, 192, 0, 320, 0, True),              # 128x320 (two phases): 4 rows of 32 per tile (level 1)
    (8, 1, 48, 64, 128, 64, 640, 0, False),            # 128x320: 2 rows of 64, two sources, two column tiles
    (8, 2, 16, 16, 640, 0, 1280, 2, True),             # 128x320 split-K
]


@pytest.mark.parametrize("case", DX_CASES)
def test_row_shared_3x3_walk_vs_torch_and_tap_major(dev, case):
    """conv_korder = 2 (default since round 4, gemm_mfma_pingpong_dx_kernel): the activation tile of a (dy, channel block) group is staged
    once and the dx = 0 / 2 K tiles read it one LDS row up / down, zero padding coming from guard rows between the image rows.  The fp32
    summation order differs from the tap-major walk, so the two agree to accumulation rounding (a few fp16 ulps on a few outputs), not
    bitwise; each equals torch within the conv tolerance; the walk is deterministic; and shapes it cannot take (checked in
    tools/micro/conv_check.cpp: W = 8, stride 2, upsample) keep the tap-major bits."""
    ops, lib = sub("ops"), sub("_lib")
    cfg, B, H, W, c0, c1, cout, split, with_res = case
    x0 = seeded((B, H, W, c0), 1)
    x1 = seeded((B, H, W, c1), 2) if c1 else None
    w = seeded((cout, c0 + c1, 3, 3), 3, scale=((c0 + c1) * 9) ** -0.5)
    b = seeded((cout,), 4, 0.1)
    xin = torch.cat([x0, x1], dim=3) if c1 else x0
    ref = _conv_ref(h(xin), h(w), b)
    res = seeded(tuple(ref.shape), 5) if with_res else None
    if with_res:
        ref = ref + h(res)
    wp = ops.pack_conv_weight(w.half().to(dev))
    args = dict(a1=None if x1 is None else x1.half().to(dev), bias=ops.pack_bias(b.to(dev), wp.shape[0]),
                resid=None if res is None else res.half().to(dev), taps=9)
    outs = {}
    try:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", cfg)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", split))
        for korder in (0, 2, 2):
            lib.check(lib.lib.sdmi_debug_set(b"conv_korder", korder))
            outs.setdefault(korder, []).append(ops.conv_gemm(x0.half().to(dev), wp, **args))
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", 0))
        lib.check(lib.lib.sdmi_debug_set(b"conv_korder", -1))
    torch.cuda.synchronize()
    tap, dx, dx2 = outs[0][0], outs[2][0], outs[2][1]
    assert torch.equal(dx, dx2), case
    assert rel_l2(tap.float().cpu(), ref) < 6e-4 and rel_l2(dx.float().cpu(), ref) < 6e-4, case
    assert rel_l2(dx.float().cpu(), tap.float().cpu()) < 1e-4, case               # measured ~2e-5 (tools/micro/conv_check.cpp)
    assert not torch.equal(dx, tap) or c0 + c1 == 64, case                        # the walk really differs (one block per row: same order)


@pytest.mark.parametrize("rows,cin,cout,cfg", [(4096, 320, 320, -1), (1024, 640, 640, -1), (256, 1280, 1280, -1), (1024, 320, 320, 9),
                                               (512, 64, 128, 0), (4096, 320, 320, 5)])
def test_transposed_output_gemm_is_the_same_bits_transposed(dev, rows, cin, cout, cfg):
    """EP_TRANSPOSE (the V projection written as V^T [C][tokens]): the MFMA operands swap roles; the products and the K-tile order
    do not change, so the result equals the ordinary projection transposed up to the matrix core's own (not transposition-symmetric)
    summation inside one instruction: at most an fp16 ulp apart on a small fraction of elements — measured and bounded here — and
    bit-identical between the LDS-direct and register-staged kernels of each form; every tile family (ping-pong, two-stage)."""
    ops, lib = sub("ops"), sub("_lib")
    b = 2
    x = seeded((b, rows, 1, cin), 1)
    w, bias = seeded((cout, cin), 2, scale=cin ** -0.5), seeded((cout,), 3, 0.1)
    wp = ops.pack_conv_weight(w.half().to(dev))
    bp = ops.pack_bias(bias.to(dev), wp.shape[0])
    trs = []
    try:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", cfg))
        for impl in ("mfma", "mfma_reg"):
            plain = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=1, impl=impl)                     # [b, rows, 1, cout]
            tr = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=1, impl=impl, transpose=True)       # [b, cout, rows]
            assert tr.shape == (b, cout, rows)
            want = plain[:, :, 0, :].permute(0, 2, 1)
            neq = (tr != want)
            frac, worst = float(neq.float().mean()), float((tr.float() - want.float()).abs().max())
            print(f"[transpose {rows}x{cin}x{cout} cfg {cfg} {impl}] unequal fraction {frac:.2e}, max abs diff {worst:.3e}")
            assert frac < 0.02 and worst <= 4e-3, (frac, worst)             # |values| < 4: one fp16 ulp = 2^-9..2^-8
            trs.append(tr)
        assert torch.equal(trs[0], trs[1])
        gen = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=1, impl="generic", transpose=True)
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1))
    ref = (x[:, :, 0, :].half().float() @ w.half().float().T + bias).permute(0, 2, 1)
    assert rel_l2(tr.float().cpu(), ref) < 6e-4 and rel_l2(gen.float().cpu(), ref) < 6e-4


@pytest.mark.parametrize("cfg,B,H,W,cin,cout,stride,up", [(-1, 2, 16, 16, 128, 320, 1, False), (5, 2, 24, 20, 64, 320, 1, False),
                                                          (4, 1, 16, 16, 128, 256, 2, False), (8, 2, 12, 12, 128, 320, 1, True),
                                                          (0, 1, 9, 7, 64, 128, 1, False), (2, 1, 8, 8, 64, 64, 2, False)])
def test_circular_padding_conv_vs_torch(dev, cfg, B, H, W, cin, cout, stride, up):
    """EP_WRAP = Conv2d(padding=1, padding_mode='circular') (p.tiling, modules/sd_hijack.py:311-318), on every tile family, with
    stride 2 and behind the fused nearest-x2 upsample; the generic kernel is the independent cross-check."""
    ops, lib = sub("ops"), sub("_lib")
    x = seeded((B, H, W, cin), 1)
    w, b = seeded((cout, cin, 3, 3), 2, scale=(cin * 9) ** -0.5), seeded((cout,), 3, 0.1)
    xin = h(x).permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="circular"), h(w), b, stride=stride).permute(0, 2, 3, 1)
    wp = ops.pack_conv_weight(w.half().to(dev))
    bp = ops.pack_bias(b.to(dev), wp.shape[0])
    try:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", cfg))
        got = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=9, stride=stride, up=up, wrap=True)
        reg = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=9, stride=stride, up=up, wrap=True, impl="mfma_reg")
        gen = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=9, stride=stride, up=up, wrap=True, impl="generic")
        zero = ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=9, stride=stride, up=up)
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1))
    assert got.shape == ref.shape
    assert rel_l2(got.float().cpu(), ref) < 6e-4 and rel_l2(gen.float().cpu(), ref) < 6e-4
    assert torch.equal(got, reg)
    assert rel_l2(zero.float().cpu(), ref) > 1e-2             # zero padding differs along the border


def test_pingpong_geglu_bit_identical(dev):
    ops, lib = sub("ops"), sub("_lib")
    rows, c = 700, 320
    x = seeded((1, rows, 1, c), 1)
    w, b = seeded((8 * c, c), 2, scale=c ** -0.5), seeded((8 * c,), 3, 0.1)
    wp = ops.pack_conv_weight(w.half().to(dev), geglu=True)
    bp = ops.pack_bias(b.to(dev), 8 * c, geglu=True)
    outs = []
    try:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", 4))
        for pipe in (0, 3):
            lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", pipe))
            outs.append(ops.conv_gemm(x.half().to(dev), wp, bias=bp, taps=1, geglu=True))
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", -1))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("cfg,pipe", [(-1, -1), (0, -1), (3, -1), (4, 3), (4, 0), (5, 3), (5, 0), (7, -1), (8, 4), (8, 0), (9, -1)])
def test_wide_epilogue_is_bit_identical_to_the_8_byte_epilogue(dev, cfg, pipe):
    """16-byte epilogue accesses (v_permlane16_swap_b32 pairs two row tiles; gemm.hip swap16) are pure data movement: bias + row
    bias + residual, ragged M, GEGLU and the transposed store must give the same bits as the 8-byte form on every tile family;
    a misaligned output (channel offset of 4) must fall back by itself."""
    ops, lib = sub("ops"), sub("_lib")
    B, H, W, cin, cout = 3, 15, 10, 128, 320                       # M = 450: ragged in every row-tile size
    x, w = seeded((B, H, W, cin), 1), seeded((cout, cin, 3, 3), 2, scale=(cin * 9) ** -0.5)
    b, rb, res = seeded((cout,), 3, 0.1), seeded((B, cout), 4), seeded((B, H, W, cout), 5)
    wp = ops.pack_conv_weight(w.half().to(dev))
    xl = seeded((2, 712, 1, 320), 6)                               # 1424 rows (712 per image: a multiple of 8 for the transposed form)
    wl, bl = seeded((2560, 320), 7, scale=320 ** -0.5), seeded((2560,), 8, 0.1)
    wg = ops.pack_conv_weight(wl.half().to(dev), geglu=True)
    bg = ops.pack_bias(bl.to(dev), 2560, geglu=True)
    wt = ops.pack_conv_weight(wl[:640].half().to(dev))
    bt = ops.pack_bias(bl[:640].to(dev), wt.shape[0])
    outs = {}
    try:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", cfg)); lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", pipe))
        for wide in (1, 0):
            lib.check(lib.lib.sdmi_debug_set(b"ep_wide", wide))
            o = [ops.conv_gemm(x.half().to(dev), wp, bias=b.to(dev), rowbias=rb.to(dev).contiguous(), resid=res.half().to(dev)),
                 ops.conv_gemm(x.half().to(dev), wp, bias=b.to(dev)),
                 ops.conv_gemm(xl.half().to(dev), wt, bias=bt, taps=1, transpose=True)]
            if cfg in (-1, 4):
                o.append(ops.conv_gemm(xl.half().to(dev), wg, bias=bg, taps=1, geglu=True))
            outs[wide] = o
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(lib.lib.sdmi_debug_set(b"gemm_pipe", -1))
        lib.check(lib.lib.sdmi_debug_set(b"ep_wide", 1))
    torch.cuda.synchronize()
    for a, c in zip(outs[1], outs[0]):
        assert torch.equal(a, c), (cfg, pipe, tuple(a.shape))
    ref = _conv_ref(h(x), h(w), b) + rb[:, None, None, :] + h(res)
    assert rel_l2(outs[1][0].float().cpu(), ref) < 6e-4


LEAN_WALK_CASES = [
    # cfg, B, H, W, c0, c1, cout, taps, split
    (7, 2, 32, 32, 320, 0, 640, 1, 0),          # 128x64 two-stage, linear walk
    (7, 3, 10, 10, 192, 64, 128, 1, 0),         # two sources, ragged M
    (12, 2, 16, 16, 1280, 0, 1280, 1, 0),       # 128x64 ring of 3 (the tuned choice of the 8x8-level linear layers): 20 K steps through the ring
    (13, 2, 16, 16, 2560, 0, 1280, 1, 4),       # 128x160 ring of 3, split-K 4
    (10, 1, 32, 32, 1280, 0, 1280, 1, 0),       # 128x160 ring of 4
    (11, 1, 32, 32, 640, 640, 640, 1, 0),       # 128x128 ring of 4, two sources
    (0, 2, 24, 24, 128, 0, 128, 9, 0),          # 128x128 two-stage, lean 3x3 walk (the VAE's 128-channel convs)
    (2, 1, 16, 16, 320, 0, 64, 9, 0),           # 64x64 (conv_out's tile), lean 3x3 walk
    (7, 2, 12, 20, 128, 64, 320, 9, 0),         # 3x3, two sources, non-square
]


@pytest.mark.parametrize("case", LEAN_WALK_CASES)
def test_lean_k_walks_and_ring_tiles_give_the_bits_of_the_general_gather(dev, case):
    """Round 5: 1x1 launches and plain 3x3 convs on the 4-wave tiles walk K with running pointers (gemm.hip LIN / LIN3), and the ring tiles
    synchronise with a bare s_barrier under counted vmcnt waits — real asynchrony, which the CPU emulation of the kernel source cannot
    order.  Same loads, same K order => the bits of the general gather (knob gemm_lin 0) and, for a ring tile, of its two-stage twin;
    repeated launches must agree (a staging race shows as a changing tile)."""
    ops, lib = sub("ops"), sub("_lib")
    cfg, B, H, W, c0, c1, cout, taps, split = case
    k = 3 if taps == 9 else 1
    x0 = seeded((B, H, W, c0), 11)
    x1 = seeded((B, H, W, c1), 12) if c1 else None
    w = seeded((cout, c0 + c1, k, k), 13, scale=((c0 + c1) * k * k) ** -0.5)
    b = seeded((cout,), 14, 0.1)
    xin = torch.cat([x0, x1], dim=3) if c1 else x0
    ref = _conv_ref(h(xin), h(w), b)
    res = seeded(tuple(ref.shape), 15)
    ref = ref + h(res)
    wp = ops.pack_conv_weight(w.half().to(dev))
    args = dict(a1=None if x1 is None else x1.half().to(dev), bias=ops.pack_bias(b.to(dev), wp.shape[0]), resid=res.half().to(dev), taps=taps)
    twin = {12: 7, 13: 9, 10: 9, 11: 0}.get(cfg)
    outs = []
    try:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_split", split)); lib.check(lib.lib.sdmi_debug_set(b"conv_korder", 0))
        for c, lin in ((cfg, 1), (cfg, 1), (cfg, 1), (cfg, 0)) + (((twin, 1),) if twin is not None else ()):
            lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", c)); lib.check(lib.lib.sdmi_debug_set(b"gemm_lin", lin))
            outs.append(ops.conv_gemm(x0.half().to(dev), wp, **args))
    finally:
        lib.check(lib.lib.sdmi_debug_set(b"gemm_cfg", -1)); lib.check(lib.lib.sdmi_debug_set(b"gemm_split", 0))
        lib.check(lib.lib.sdmi_debug_set(b"gemm_lin", 1)); lib.check(lib.lib.sdmi_debug_set(b"conv_korder", -1))
    torch.cuda.synchronize()
    assert rel_l2(outs[0].float().cpu(), ref) < 6e-4, case
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), case


def test_mfma_glds_and_register_staging_agree_bitwise(dev):
    """Same LDS image, same MFMA order => identical bits; catches any mismatch in the LDS-direct load path."""
    ops = sub("ops")
    x, w = seeded((2, 16, 16, 128), 1), seeded((128, 128, 3, 3), 2, scale=(128 * 9) ** -0.5)
    wp = ops.pack_conv_weight(w.half().to(dev))
    a = ops.conv_gemm(x.half().to(dev), wp, impl="mfma")
    b = ops.conv_gemm(x.half().to(dev), wp, impl="mfma_reg")
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, heads):
    b, n, c = q.shape
    d = c // heads
    sp = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    s = torch.einsum('bhid,bhjd->bhij', sp(q), sp(k)) * d ** -0.5
    o = torch.einsum('bhij,bhjd->bhid', s.softmax(-1), sp(v))
    return o.permute(0, 2, 1, 3).reshape(b, n, c)


@pytest.mark.parametrize("d,heads,n,m", [
    (40, 8, 256, 256), (40, 2, 200, 77), (64, 2, 128, 128), (64, 1, 70, 333), (80, 8, 256, 256), (80, 2, 64, 77),
    (128, 1, 130, 64), (160, 8, 64, 64), (160, 2, 256, 77), (160, 1, 300, 300),
    (32, 2, 50, 60),          # generic kernel (head size without an MFMA instance)
])
def test_attention_vs_oracle(dev, d, heads, n, m):
    ops = sub("ops")
    b = 2
    q, k, v = seeded((b, n, heads * d), 1), seeded((b, m, heads * d), 2), seeded((b, m, heads * d), 3)
    ref = _attn_ref(h(q), h(k), h(v), heads)
    got = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), heads)
    torch.cuda.synchronize()
    # fp32 scores/softmax, P rounded to fp16 before PV, fp16 output
    assert rel_l2(got.float().cpu(), ref) < 1.5e-3, (d, heads, n, m)


def test_attention_matches_reference_sub_quadratic_fixture(dev, golden_dir):
    """Against outputs of the reference's own modules/sub_quadratic_attention.py (tests/golden/make_golden.py)."""
    ops = sub("ops")
    z = np.load(os.path.join(golden_dir, "subquad_attention.npz"))
    for ci in range(3):
        bh, n, mk, d = [int(v) for v in z[f"c{ci}_shape"]]
        q, k, v = seeded((bh, n, d), 10 + ci), seeded((bh, mk, d), 20 + ci), seeded((bh, mk, d), 30 + ci)
        got = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), heads=1)
        # inputs are rounded to fp16 for the kernel, the fixture used fp32 inputs: tolerance covers that rounding
        assert rel_l2(got.float().cpu(), z[f"c{ci}_out"]) < 4e-3


def test_attention_softmax_is_shift_invariant_and_handles_spikes(dev):
    """Online-softmax rescale branch: one key with a huge score late in the sequence (forces a max jump at a later KV
    tile) and a constant shift of all scores must not change the result beyond rounding."""
    ops = sub("ops")
    d, heads, n, m = 64, 1, 64, 256
    q, k, v = seeded((1, n, d), 1), seeded((1, m, d), 2), seeded((1, m, d), 3)
    k[0, 200] = q[0, 5] * 6.0                     # spike for query 5 inside the 4th KV tile
    ref = _attn_ref(h(q), h(k), h(v), heads)
    got = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), heads)
    assert rel_l2(got.float().cpu(), ref) < 1.5e-3
    assert float((got[0, 5].float().cpu() - ref[0, 5]).abs().max()) < 2e-2


def test_attention_text_context_as_one_96_key_tile(dev):
    """Knob attn_kvt = 96: key sequences of 65..96 tokens (the 77-token text context) run as ONE tile of three 32-key blocks instead of a
    64-key tile plus a ragged one — every head size the UNets use, ragged query counts, 65 / 77 / 96 keys; same arithmetic per score, so the
    result agrees with the two-tile form to fp16 rounding of P and with fp32 to the attention tolerance."""
    ops, lib = sub("ops"), sub("_lib")
    for d, heads, n, m in ((40, 8, 512, 77), (40, 2, 200, 77), (40, 1, 130, 65), (40, 1, 128, 96), (80, 8, 256, 77), (160, 8, 64, 77), (64, 10, 300, 77), (64, 20, 128, 96)):
        q, k, v = seeded((2, n, heads * d), 91), seeded((2, m, heads * d), 92), seeded((2, m, heads * d), 93)
        base = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), heads)
        lib.check(lib.lib.sdmi_debug_set(b"attn_kvt", 96))
        try:
            got = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), heads)
            torch.cuda.synchronize()
        finally:
            lib.check(lib.lib.sdmi_debug_set(b"attn_kvt", 0))
        assert rel_l2(got.float().cpu(), _attn_ref(h(q), h(k), h(v), heads)) < 5e-4, (d, heads, n, m)
        assert rel_l2(got.float().cpu(), base.float().cpu()) < 5e-4, (d, heads, n, m)


@pytest.mark.parametrize("variant", [17, 20, 21, 30, 31])
def test_attention_role_offset_kernel(dev, variant):
    """(17 = the production 4-wave kernel with the softmax shift folded into the S^T MFMA like 21.)  The 8-wave role-offset kernel (attn_occ 20; 21 = with the softmax shift folded into the S^T MFMA: Q pre-multiplied by
    scale * log2 e, K's padding column at 1.0, Q's padding element at -shift): level-0 / hires shapes, ragged query and key counts, 1 to
    9 KV tiles, d = 64 (SDXL; no padding column, so 21 runs the unfolded arithmetic there); 30 / 31 = the same two forms with THREE wave
    groups (12 waves, 384 queries per workgroup, the VALU work of a tile split over two sections) — against fp32; the unfolded form runs
    variant 15's arithmetic in variant 15's order and must give its bits.  Then the shift logic of 21 on adversarial rows: a spike
    late in the sequence, scores that rise tile after tile (the shift is raised in every tile), all scores far below zero (first-tile
    initialisation with a negative shift) and far above (shift near 100: fp16 spacing 0.06)."""
    ops, lib = sub("ops"), sub("_lib")

    def run(q, k, v, heads, occ):
        lib.check(lib.lib.sdmi_debug_set(b"attn_occ", occ))
        lib.check(lib.lib.sdmi_debug_set(b"attn_tau", 0))       # (variant 15's bits as the yardstick: re-based whenever a running maximum moves)
        lib.check(lib.lib.sdmi_debug_set(b"attn_fold_min_m", 0))     # (15 means 15 here: not handed on to 17 for the long rows)
        try:
            out = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), heads)
            torch.cuda.synchronize()
        finally:
            lib.check(lib.lib.sdmi_debug_set(b"attn_occ", 15))
            lib.check(lib.lib.sdmi_debug_set(b"attn_tau", ATTN_TAU_DEFAULT))
            lib.check(lib.lib.sdmi_debug_set(b"attn_fold_min_m", ATTN_FOLD_MIN_M_DEFAULT))
        return out

    for d, heads, n, m in ((40, 8, 512, 512), (40, 2, 256, 256), (40, 1, 300, 333), (40, 2, 1024, 576), (40, 1, 256, 290), (40, 8, 4096, 4096),
                           (40, 2, 384, 256), (40, 1, 768, 320), (40, 1, 700, 449), (40, 2, 200, 77), (40, 1, 130, 64), (64, 2, 512, 512), (64, 1, 260, 400)):
        b = 1 if n >= 4096 else 2
        q, k, v = seeded((b, n, heads * d), 61), seeded((b, m, heads * d), 62), seeded((b, m, heads * d), 63)
        got = run(q, k, v, heads, variant)
        e = rel_l2(got.float().cpu(), _attn_ref(h(q), h(k), h(v), heads))
        assert e < 5e-4, (d, heads, n, m, e)
        if variant in (20, 30) or d != 40:
            base = run(q, k, v, heads, 15 if d == 40 else 0)
            assert torch.equal(got, base) or rel_l2(got.float().cpu(), base.float().cpu()) < 1e-4, (d, heads, n, m)
    # adversarial rows for the shift bookkeeping
    d, heads, n, m = 40, 1, 384, 576
    q, k, v = seeded((1, n, d), 71), seeded((1, m, d), 72), seeded((1, m, d), 73)
    k2 = k.clone()
    k2[0, 500] = q[0, 5] * 6.0                                # spike for query 5 in the 8th KV tile
    k3 = k.clone()
    ramp = torch.linspace(-4.0, 6.0, m)[:, None]              # q . k rises along the sequence for query 7: a new maximum in every tile
    k3[0] = k3[0] * 0.2 + ramp * q[0, 7][None, :] / q[0, 7].norm() * 2.0
    k4 = k.clone() - 8.0 * q[0, 9][None, None, :] / q[0, 9].norm()      # every score of query 9 far below zero
    k5 = k.clone() + 40.0 * q[0, 11][None, None, :] / q[0, 11].norm()   # ... of query 11 far above (shift ~ 90 in log2 units)
    for kk, row in ((k2, 5), (k3, 7), (k4, 9), (k5, 11)):
        ref = _attn_ref(h(q), h(kk), h(v), heads)
        got = run(q, kk, v, heads, variant).float().cpu()
        assert torch.isfinite(got).all()
        assert rel_l2(got, ref) < 1.5e-3, row
        assert float((got[0, row] - ref[0, row]).abs().max()) < 2e-2, row


# ------------------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c0,c1,hw", [(320, 0, (8, 8)), (64, 0, (16, 16)), (1280, 640, (4, 4)), (128, 64, (32, 32)),
                                      (2560, 0, (8, 8)), (128, 0, (64, 64)),
                                      # the single-launch register-resident kernel (cpg % 8 == 0, <= 24 vectors per thread): one and two
                                      # sources, 4 / 12 / 24 vectors per thread, a pixel count that is not a multiple of the stride
                                      (1280, 0, (8, 8)), (1280, 1280, (16, 16)), (1280, 0, (32, 32)), (512, 0, (16, 16)), (256, 0, (8, 8)),
                                      (1280, 1280, (5, 7)), (2560, 0, (16, 16))])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_silu_vs_torch(dev, c0, c1, hw, silu):
    ops = sub("ops")
    B, (H, W) = 2, hw
    x0 = seeded((B, H, W, c0), 1) * 1.7 + 0.3
    x1 = seeded((B, H, W, c1), 2) * 0.6 - 0.2 if c1 else None
    C = c0 + c1
    g, bt = 1 + 0.1 * seeded((C,), 3), 0.1 * seeded((C,), 4)
    xx = h(x0) if x1 is None else torch.cat([h(x0), h(x1)], dim=3)
    ref = F.group_norm(xx.permute(0, 3, 1, 2), 32, g, bt, eps=1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    got = ops.groupnorm(x0.half().to(dev), g.to(dev), bt.to(dev), x1=None if x1 is None else x1.half().to(dev), eps=1e-5, silu=silu)
    assert rel_l2(got.float().cpu(), ref) < 5e-4


@pytest.mark.parametrize("c", [64, 320, 640, 1280, 1920, 2560, 3072])     # 2560: hidden width of a [1, 2, 1] hypernetwork on the 1280-wide levels
def test_layernorm_vs_torch(dev, c):
    ops = sub("ops")
    x = seeded((3, 50, c), 1) * 2 + 0.5
    g, bt = 1 + 0.1 * seeded((c,), 2), 0.1 * seeded((c,), 3)
    ref = F.layer_norm(h(x), (c,), g, bt, eps=1e-5)
    got = ops.layernorm(x.half().to(dev), g.to(dev), bt.to(dev))
    assert rel_l2(got.float().cpu(), ref) < 5e-4


# ------------------------------------------------------------------------------------------------------------
# sampler arithmetic
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
@pytest.mark.parametrize("src,dst", [((64, 64), (128, 128)), ((16, 24), (25, 31)), ((40, 30), (24, 21)), ((96, 96), (20, 36))])
def test_latent_resize_antialiased_matches_torch_interpolate(dev, mode, src, dst):
    """"Latent (antialiased)" / "Latent (bicubic antialiased)": F.interpolate(..., antialias=True) (modules/shared.py:57, 59)."""
    ops = sub("ops")
    x = seeded((2, 4) + src, 61)
    want = F.interpolate(x, size=dst, mode=mode, antialias=True)
    got = ops.latent_resize(x.to(dev), dst, mode, antialias=True)
    assert got.shape == want.shape and float((got.cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("mode", ["nearest", "nearest-exact", "bilinear", "bicubic"])
@pytest.mark.parametrize("src,dst", [((64, 64), (128, 128)), ((16, 24), (25, 31)), ((40, 30), (24, 21))])
def test_latent_resize_matches_torch_interpolate(dev, mode, src, dst):
    """The hires-fix latent upscale (modules/processing.py:1392) against torch's own CPU interpolate, incl. non-integer and
    downscaling factors."""
    ops = sub("ops")
    x = seeded((2, 4) + src, 60)
    want = F.interpolate(x, size=dst, mode=mode, antialias=False) if mode in ("bilinear", "bicubic") else F.interpolate(x, size=dst, mode=mode)
    got = ops.latent_resize(x.to(dev), dst, mode)
    assert got.shape == want.shape
    if mode.startswith("nearest"):
        assert torch.equal(got.cpu(), want)
    else:
        assert float((got.cpu() - want).abs().max()) < 2e-5


def test_lincomb_and_mask_blend(dev):
    ops = sub("ops")
    ts = [seeded((2, 4, 8, 8), 40 + i) for i in range(6)]
    cs = [0.5, -1.25, 3.0, 0.125, -0.75, 2.0]
    for n in (1, 2, 3, 6):
        want = ts[0] * cs[0]
        for k in range(1, n):
            want = want + ts[k] * cs[k]
        got = ops.lincomb(torch.empty(2, 4, 8, 8, device=dev), [t.to(dev) for t in ts[:n]], cs[:n])
        assert torch.equal(got.cpu(), want), n                       # same left-to-right fp32 mul/add sequence
    x = ts[0].to(dev).clone()
    ops.lincomb(x, [x, ts[1].to(dev)], [1.0, 0.5])                   # in place on a term
    assert torch.equal(x.cpu(), ts[0] * 1.0 + ts[1] * 0.5)
    mask = (seeded((2, 1, 8, 8), 50) > 0).float()
    xb = ops.mask_blend(ts[2].to(dev).clone(), ts[3].to(dev), mask.to(dev), (1 - mask).to(dev))
    assert torch.equal(xb.cpu(), ts[2] * (1 - mask) + ts[3] * mask)


def test_sampler_kernels_match_oracle_formulas(dev):
    from oracle import kdiffusion as okd
    lib = sub("_lib")
    L, check, ptr, sp = lib.lib, lib.check, lib.ptr, lib.stream_ptr
    B, chw = 3, 4 * 8 * 8
    x = seeded((B, 4, 8, 8), 1) * 5
    eps = seeded((2 * B, 4, 8, 8), 2)
    noise = seeded((B, 4, 8, 8), 3)
    sigma = torch.tensor(3.3)
    # CompVisDenoiser + combine_denoised
    c_out = -sigma
    x_out = torch.cat([x, x]) + eps * c_out
    den_ref = okd.CFGDenoiser.combine_denoised(x_out, [[(i, 1.0)] for i in range(B)], B, 7.0)
    xd, ed = x.to(dev), eps.to(dev)
    den = torch.empty_like(xd)
    co = torch.full((B,), float(c_out), device=dev)
    check(L.sdmi_cfg_combine(ptr(xd), ptr(ed), ptr(co), 7.0, 0, None, None, None, ptr(den), B, chw, sp()))
    np.testing.assert_allclose(den.cpu().numpy(), den_ref.numpy(), rtol=1e-6, atol=1e-5)
    # Euler-ancestral update
    sd_, su = okd.get_ancestral_step(torch.tensor(3.3), torch.tensor(2.1))
    ref = x + okd.to_d(x, torch.tensor(3.3), den_ref) * (sd_ - 3.3)
    ref = ref + noise * 1.0 * su
    xs = xd.clone()
    check(L.sdmi_euler_step(ptr(xs), ptr(den), ptr(noise.to(dev)), 3.3, float(sd_), float(su), 1.0, xs.numel(), sp()))
    np.testing.assert_allclose(xs.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-5)
    # prepare input (fp16 and fp32)
    ci = torch.full((B,), 0.29, device=dev)
    xin = torch.empty((2 * B, 4, 8, 8), dtype=torch.float16, device=dev)
    check(L.sdmi_cfg_prepare_input(ptr(xd), ptr(ci), ptr(xin), 0, B, 2, chw, sp()))
    want = (torch.cat([x, x]) * torch.tensor(0.29)).half()
    assert torch.equal(xin.cpu(), want)
    # uint8 conversion incl. truncation and clamping
    img = torch.tensor([-1.2, -1.0, 0.0, 0.999, 1.0, 1.7, 0.5019, -0.25]).reshape(1, 1, 2, 4).repeat(1, 3, 1, 1)
    got = sub("ops").image_to_u8(img.to(dev)).cpu().numpy()
    from oracle.vae import to_uint8_hwc
    np.testing.assert_array_equal(got, to_uint8_hwc(img))


def test_image_rng_variation_seeds_and_seed_resize_vs_reference(dev, golden_dir):
    """Device ImageRNG (Philox kernel, sdmi_slerp, centred paste) against the reference-generated draws of
    tests/golden/image_rng.npz: plain / ENSD / seed-resize are bit-exact, the slerp cases agree to fp32 rounding of sin / acos."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    rng = sub("rng")
    z = np.load(os.path.join(golden_dir, "image_rng.npz"))
    for name, shape, seeds, kw, ensd in mg.IMAGE_RNG_CASES:
        r = rng.ImageRNG(shape, seeds, eta_noise_seed_delta=ensd, device=dev, **kw)
        for k in range(3):
            got = r.next().cpu().numpy()
            if "subseed" in name and k == 0:
                np.testing.assert_allclose(got, z[f"{name}_{k}"], rtol=0, atol=3e-6, err_msg=name)
            else:
                assert np.array_equal(got, z[f"{name}_{k}"]), (name, k)


def test_attention_experiment_variants_match_production_kernel(dev):
    """The forms of the d = 40 flash kernel (SDMI_ATTN_OCC / sdmi_debug_set("attn_occ", v): 0 = round-1 kernel, 5 = 128-VGPR budget
    + lazy O rescale, 15 = the default: lazy rescale + MFMA fragments prefetched per phase + v_permlane32_swap max exchange) run the
    same arithmetic in the same order: self- and cross-attention shapes incl. ragged tails and 1 / 2 / 3 / 6 KV tiles agree with 0."""
    ops, lib = sub("ops"), sub("_lib")
    for heads, n, m in ((8, 512, 512), (2, 200, 77), (1, 130, 333), (1, 128, 64), (1, 70, 128), (2, 256, 192)):
        q, k, v = seeded((2, n, heads * 40), 41).half().to(dev), seeded((2, m, heads * 40), 42).half().to(dev), seeded((2, m, heads * 40), 43).half().to(dev)
        lib.check(lib.lib.sdmi_debug_set(b"attn_occ", 0))
        lib.check(lib.lib.sdmi_debug_set(b"attn_tau", 0))           # (0 re-bases lazily too since round 6: every form under the same rule)
        try:
            base = ops.attention(q, k, v, heads)
            lib.check(lib.lib.sdmi_debug_set(b"attn_tau", -1))      # ... and the round-1 behaviour (O^T rescaled in every tile) has the same bits
            assert torch.equal(base, ops.attention(q, k, v, heads)), (heads, n, m)
        finally:
            lib.check(lib.lib.sdmi_debug_set(b"attn_occ", 15))
            lib.check(lib.lib.sdmi_debug_set(b"attn_tau", ATTN_TAU_DEFAULT))
        for variant in (5, 15):
            try:
                lib.check(lib.lib.sdmi_debug_set(b"attn_occ", variant))
                lib.check(lib.lib.sdmi_debug_set(b"attn_tau", 0))       # same re-basing rule as 0: whenever a running maximum moves
                got = ops.attention(q, k, v, heads)
                torch.cuda.synchronize()
            finally:
                lib.check(lib.lib.sdmi_debug_set(b"attn_occ", 15))
                lib.check(lib.lib.sdmi_debug_set(b"attn_tau", ATTN_TAU_DEFAULT))
            # same operation counts in the compiled loops (packed vs scalar forms of the same fp32 ops): the bits are expected equal
            assert torch.equal(got, base) or rel_l2(got.float().cpu(), base.float().cpu()) < 1e-4, (heads, n, m, variant)



def test_attention_lazy_rebase_slack(dev):
    """Round 6: the lazy forms of the d = 40 kernel re-base the exponent (and rescale O^T) only when some query's tile maximum exceeds the
    base in use by more than tau (log2 units; knob attn_tau, default 8) — P <= 2^tau instead of <= 1, the same softmax up to the rounding of
    P to fp16.  tau = 0 is the old rule and gives variant 0's bits; the default holds fp32 to the same bound and stays within two fp16
    realisations of tau = 0; rows built against the bookkeeping: scores that climb by less than tau per tile (never re-based after the first
    tile: P grows to 2^tau-ish), by more than tau per tile (re-based in every tile), one spike far above everything late in the row,
    and a first tile far below zero."""
    ops, lib = sub("ops"), sub("_lib")

    def run(q, k, v, heads, tau, occ=15):
        lib.check(lib.lib.sdmi_debug_set(b"attn_occ", occ))
        lib.check(lib.lib.sdmi_debug_set(b"attn_tau", tau))
        lib.check(lib.lib.sdmi_debug_set(b"attn_fold_min_m", 0))     # (15 means 15 here: not handed on to 17 for the long rows)
        try:
            out = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), heads)
            torch.cuda.synchronize()
        finally:
            lib.check(lib.lib.sdmi_debug_set(b"attn_occ", 15))
            lib.check(lib.lib.sdmi_debug_set(b"attn_tau", ATTN_TAU_DEFAULT))
            lib.check(lib.lib.sdmi_debug_set(b"attn_fold_min_m", ATTN_FOLD_MIN_M_DEFAULT))
        return out.float().cpu()

    for heads, n, m in ((8, 512, 512), (2, 200, 77), (1, 130, 333), (2, 1024, 4096), (1, 256, 290)):
        q, k, v = seeded((2, n, heads * 40), 141), seeded((2, m, heads * 40), 142), seeded((2, m, heads * 40), 143)
        ref = _attn_ref(h(q), h(k), h(v), heads)
        old, new = run(q, k, v, heads, 0), run(q, k, v, heads, ATTN_TAU_DEFAULT)
        v0 = run(q, k, v, heads, 0, occ=0)
        assert torch.equal(old, v0) or rel_l2(old, v0) < 1e-4, (heads, n, m)
        assert rel_l2(old, ref) < 5e-4 and rel_l2(new, ref) < 5e-4, (heads, n, m, rel_l2(old, ref), rel_l2(new, ref))
        assert rel_l2(new, old) < 6e-4, (heads, n, m, rel_l2(new, old))
        fold = run(q, k, v, heads, ATTN_TAU_DEFAULT, occ=17)       # the folded-shift form (hires default): shift raised with the same slack
        assert rel_l2(fold, ref) < 5e-4, (heads, n, m, rel_l2(fold, ref))
    # the production dispatch at the level-0 shape (form 17 from 1024 keys on, with the slack)
    q, k, v = seeded((1, 4096, 320), 161), seeded((1, 4096, 320), 162), seeded((1, 4096, 320), 163)
    got = ops.attention(q.half().to(dev), k.half().to(dev), v.half().to(dev), 8).float().cpu()
    assert rel_l2(got, _attn_ref(h(q), h(k), h(v), 8)) < 5e-4
    d, n, m = 40, 256, 640                               # 10 KV tiles of 64 keys
    q = torch.zeros(1, n, d)
    q[..., 0] = 1.0
    q[:, :, 1:] = 0.05 * seeded((1, n, d - 1), 151)
    scale = d ** -0.5
    tile = torch.arange(m) // 64
    for name, per_tile in (("climbs by 3 per tile", 3.0), ("climbs by 11 per tile", 11.0), ("falls by 5 per tile", -5.0)):
        k = 0.05 * seeded((1, m, d), 152)
        k[0, :, 0] = (per_tile * tile.float() - (60.0 if per_tile > 0 else 0.0)) / (scale * 1.4426950408889634)     # log2-unit steps
        v = seeded((1, m, d), 153)
        ref = _attn_ref(h(q), h(k), h(v), 1)
        for occ in (15, 17):
            for tau in (0, 4, ATTN_TAU_DEFAULT, 12):
                e = rel_l2(run(q, k, v, 1, tau, occ=occ), ref)
                assert e < 5e-4, (name, occ, tau, e)
    k = 0.05 * seeded((1, m, d), 154)
    k[0, 500, 0] = 40.0 / scale                          # one key 40 nats above the rest, in the 8th tile
    v = seeded((1, m, d), 155)
    ref = _attn_ref(h(q), h(k), h(v), 1)
    for occ in (15, 17):
        for tau in (0, ATTN_TAU_DEFAULT):
            assert rel_l2(run(q, k, v, 1, tau, occ=occ), ref) < 5e-4, (occ, tau)


# ------------------------------------------------------------------------------------------------------------
# fused row-local chains of the transformer block (csrc/rowchain.hip)
# ------------------------------------------------------------------------------------------------------------
def test_rowchain_feed_forward_vs_fp32(dev):
    """norm3 -> ff.net.0.proj (GEGLU) -> ff.net.2 -> + x as one launch, at the C1 level-0 width, against torch fp32 on the fp16-rounded
    operands (exact-erf GELU, as ldm's GEGLU)."""
    ops = sub("ops")
    g = torch.Generator(device="cpu").manual_seed(11)
    C_, hidden, rows = 320, 1280, 4096
    x = torch.randn(rows, C_, generator=g).half()
    gam, bet = 1 + 0.1 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    w1 = (torch.randn(2 * hidden, C_, generator=g) / C_ ** 0.5).half()
    b1 = 0.1 * torch.randn(2 * hidden, generator=g)
    w2 = (torch.randn(C_, hidden, generator=g) / hidden ** 0.5).half()
    b2 = 0.1 * torch.randn(C_, generator=g)
    packs = ops.rowchain_ff_pack(w1.to(dev), b1.to(dev), w2.to(dev))
    out = ops.rowchain_ff(x.to(dev), gam.to(dev), bet.to(dev), packs, b2.to(dev), hidden).float().cpu()
    xf = x.float()
    n = F.layer_norm(xf, (C_,), gam, bet, 1e-5)
    hcat = F.linear(n, w1.float(), b1)
    ref = xf + F.linear(hcat[:, :hidden] * F.gelu(hcat[:, hidden:]), w2.float(), b2)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 3.5e-4                          # measured 2.1e-4 (the fp16 rounding of the output)
    assert rel_l2(out - xf, ref - xf) < 8e-4                  # measured 4.9e-4 on the branch alone (fp16 LayerNorm output and hidden tensor)
