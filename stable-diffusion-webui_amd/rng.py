"""Per-image noise streams generated on the GPU (mirror of modules/rng.py + modules/rng_philox.py, "NV" source).

``ImageRNG`` keeps the reference's semantics (modules/rng.py:99-163): image i owns ``Generator(seed_i)``; the first
``next()`` returns the initial latent noise, every later ``next()`` one more draw per image, so an image's stream does
not depend on which batch (or which GPU) it is generated in.  Subseed slerp and seed-resize keep the reference's
formulas (modules/rng.py:85-96, 131-143) on device tensors.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import lib, check, ptr, stream_ptr


class Generator:
    """rng_philox.Generator: counter offset advances by one per randn() call (modules/rng_philox.py:77-102)."""

    def __init__(self, seed: int, device="cuda"):
        self.seed = int(seed)
        self.offset = 0
        self.device = device

    def randn(self, shape) -> torch.Tensor:
        out = ops.philox_randn(shape, self.seed, self.offset, self.device)
        self.offset += 1
        return out


def slerp(val, low, high):
    """modules/rng.py:85-96 on one image's [C, H, W] noise pair (sdmi_slerp)."""
    low, high = low.contiguous(), high.contiguous()
    c, h, w = low.shape
    out = torch.empty_like(low)
    scratch = torch.empty((c * w,), dtype=torch.float32, device=low.device)
    check(lib.sdmi_slerp(ptr(out), ptr(low), ptr(high), float(val), c, h, w, ptr(scratch), stream_ptr()), "sdmi_slerp")
    return out


def _paste_centered(canvas, patch):
    """modules/rng.py:131-143: the centre-aligned overlap of ``patch`` replaces that part of ``canvas`` (both [C, h, w])."""
    (_, big_h, big_w), (_, h, w) = canvas.shape, patch.shape
    oy, ox = (big_h - h) // 2, (big_w - w) // 2            # negative offset: the patch is the larger one and gets cropped
    hh, ww = (h if oy >= 0 else h + 2 * oy), (w if ox >= 0 else w + 2 * ox)
    cy, cx, py, px = max(oy, 0), max(ox, 0), max(-oy, 0), max(-ox, 0)
    canvas[:, cy:cy + hh, cx:cx + ww] = patch[:, py:py + hh, px:px + ww]
    return canvas


class ImageRNG:
    def __init__(self, shape, seeds, subseeds=None, subseed_strength=0.0, seed_resize_from_h=0, seed_resize_from_w=0,
                 eta_noise_seed_delta: int = 0, device="cuda"):
        self.shape = tuple(map(int, shape))
        self.seeds = list(seeds)
        self.subseeds = subseeds
        self.subseed_strength = subseed_strength
        self.seed_resize_from_h = seed_resize_from_h
        self.seed_resize_from_w = seed_resize_from_w
        self.eta_noise_seed_delta = eta_noise_seed_delta
        self.device = device
        self.generators = [Generator(seed, device) for seed in self.seeds]
        self.is_first = True

    def first(self):
        """modules/rng.py:113-151.  With seed-resize the noise of the OTHER image size comes from fresh generators (so it equals
        what a job of that size would start from) and is pasted over this size's own first draw; the variation seed is blended
        in before the paste."""
        resize = self.seed_resize_from_h > 0 and self.seed_resize_from_w > 0
        noise_shape = (self.shape[0], int(self.seed_resize_from_h) // 8, int(self.seed_resize_from_w // 8)) if resize else self.shape
        resized = noise_shape != self.shape
        vary = self.subseeds is not None and self.subseed_strength != 0
        xs = []
        for i, (seed, generator) in enumerate(zip(self.seeds, self.generators)):
            noise = Generator(seed, self.device).randn(noise_shape) if resized else generator.randn(self.shape)
            if vary:
                subseed = self.subseeds[i] if i < len(self.subseeds) else 0
                noise = slerp(self.subseed_strength, noise, Generator(subseed, self.device).randn(noise_shape))
            if resized:
                noise = _paste_centered(generator.randn(self.shape), noise)
            xs.append(noise)
        if self.eta_noise_seed_delta:
            self.generators = [Generator(seed + self.eta_noise_seed_delta, self.device) for seed in self.seeds]
        return torch.stack(xs)

    def next(self):
        if self.is_first:
            self.is_first = False
            return self.first()
        return torch.stack([g.randn(self.shape) for g in self.generators])
