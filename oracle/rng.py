"""Philox4x32-10 + Box-Muller noise source and ImageRNG (oracle; tests only).

Restates modules/rng_philox.py:32-102 (randn_source "NV": the CPU emulation of CUDA torch.randn) and the
plain path of modules/rng.py:99-163 (ImageRNG: one generator per image seeded ``seed + i`` by the caller,
``first()`` draws the initial latent, every later ``next()`` draws one more tensor per generator;
eta_noise_seed_delta re-seeds after the first draw, :147-149), the variation-seed slerp (:85-96, 120-127) and seed-resize
(:131-143) — the whole class pinned by tests/golden/image_rng.npz.

Arithmetic notes that matter for bit-exactness (all visible in rng_philox.py):
  * counter = [offset, 0, index, 0]; key = (seed lo32, seed hi32); 10 rounds, key += (0x9E3779B9, 0xBB67AE85)
    between rounds (:44-63, :84-99)
  * Box-Muller uses outputs 0 and 1 only; constants are float32 (2.3283064e-10 and that times 6.2831855)
    but ``uint32_array * float32_array`` promotes to float64 in numpy, so log/sqrt/sin run in float64 and
    the result is rounded to float32 once at the end (:66-74).
Golden vector: rng_philox.py:12-14 (Generator(seed=0).randn((3,4))).
"""
from __future__ import annotations

import numpy as np
import torch

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
TWO_POW32_INV = np.array([2.3283064e-10], dtype=np.float32)
TWO_POW32_INV_2PI = np.array([2.3283064e-10 * 6.2831855], dtype=np.float32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments are uint32 numpy arrays of equal shape; returns the four output words."""
    mask = np.uint64(0xFFFFFFFF)
    for r in range(10):
        p0 = c0.astype(np.uint64) * M0
        p1 = c2.astype(np.uint64) * M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        if r != 9:
            with np.errstate(over="ignore"):
                k0 = (k0 + W0).astype(np.uint32)
                k1 = (k1 + W1).astype(np.uint32)
    return c0, c1, c2, c3


def box_muller_first(x, y):
    u = x * TWO_POW32_INV + TWO_POW32_INV / 2
    v = y * TWO_POW32_INV_2PI + TWO_POW32_INV_2PI / 2
    s = np.sqrt(-2.0 * np.log(u))
    return (s * np.sin(v)).astype(np.float32)


class Generator:
    def __init__(self, seed: int):
        self.seed = int(seed)
        self.offset = 0

    def randn(self, shape):
        n = int(np.prod(shape))
        idx = np.arange(n, dtype=np.uint32)
        zeros = np.zeros(n, dtype=np.uint32)
        off = np.full(n, self.offset, dtype=np.uint32)
        self.offset += 1
        k0 = np.full(n, self.seed & 0xFFFFFFFF, dtype=np.uint32)
        k1 = np.full(n, (self.seed >> 32) & 0xFFFFFFFF, dtype=np.uint32)
        g0, g1, _, _ = philox4x32_10(off, zeros.copy(), idx, zeros.copy(), k0, k1)
        return box_muller_first(g0, g1).reshape(shape)


def slerp(val, low, high):
    """modules/rng.py:85-96: spherical interpolation of two [C, H, W] noise tensors, direction measured along dim 1; nearly
    parallel inputs fall back to a (reversed-weight, as in the reference) linear blend."""
    dot = ((low / torch.norm(low, dim=1, keepdim=True)) * (high / torch.norm(high, dim=1, keepdim=True))).sum(1)
    if dot.mean() > 0.9995:
        return low * val + high * (1 - val)
    omega = torch.acos(dot)
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


def paste_centered(canvas, patch):
    """modules/rng.py:131-143: copy the centre-aligned overlap of ``patch`` [C, h, w] into ``canvas`` [C, H, W] (in place)."""
    (_, H, W), (_, h, w) = canvas.shape, patch.shape
    oy, ox = (H - h) // 2, (W - w) // 2                     # negative: the patch is larger and is cropped instead
    hh, ww = (h if oy >= 0 else h + 2 * oy), (w if ox >= 0 else w + 2 * ox)
    cy, cx, py, px = max(oy, 0), max(ox, 0), max(-oy, 0), max(-ox, 0)
    canvas[:, cy:cy + hh, cx:cx + ww] = patch[:, py:py + hh, px:px + ww]
    return canvas


class ImageRNG:
    """modules/rng.py:99-163 for randn_source "NV", pinned by tests/golden/image_rng.npz (the reference class executed over the
    real rng_philox.py): per-image generators; variation seeds (subseed slerp, :120-127), seed-resize (:114, 122-143: the noise of
    another image size is drawn from FRESH generators and pasted centred over this size's own first draw), eta_noise_seed_delta
    (:147-149)."""

    def __init__(self, shape, seeds, eta_noise_seed_delta: int = 0, subseeds=None, subseed_strength=0.0, seed_resize_from_h=0,
                 seed_resize_from_w=0):
        self.shape = tuple(map(int, shape))
        self.seeds = list(seeds)
        self.ensd = eta_noise_seed_delta
        self.subseeds, self.subseed_strength = subseeds, subseed_strength
        self.resize_from = (seed_resize_from_h, seed_resize_from_w)
        self.generators = [Generator(s) for s in self.seeds]
        self.is_first = True

    def first(self):
        rh, rw = self.resize_from
        noise_shape = self.shape if rh <= 0 or rw <= 0 else (self.shape[0], int(rh) // 8, int(rw // 8))
        draw = lambda seed: torch.from_numpy(Generator(seed).randn(noise_shape))
        xs = []
        for i, (seed, gen) in enumerate(zip(self.seeds, self.generators)):
            resized = noise_shape != self.shape
            noise = draw(seed) if resized else torch.from_numpy(gen.randn(self.shape))
            if self.subseeds is not None and self.subseed_strength != 0:
                noise = slerp(self.subseed_strength, noise, draw(self.subseeds[i] if i < len(self.subseeds) else 0))
            if resized:
                noise = paste_centered(torch.from_numpy(gen.randn(self.shape)), noise)
            xs.append(noise)
        if self.ensd:
            self.generators = [Generator(s + self.ensd) for s in self.seeds]
        return torch.stack(xs)

    def next(self):
        if self.is_first:
            self.is_first = False
            return self.first()
        return torch.stack([torch.from_numpy(g.randn(self.shape)) for g in self.generators])
