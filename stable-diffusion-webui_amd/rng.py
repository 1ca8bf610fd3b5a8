"""Per-image noise streams generated on the GPU (mirror of modules/rng.py + modules/rng_philox.py, "NV" source).

``ImageRNG`` keeps the reference's semantics (modules/rng.py:99-163): image i owns ``Generator(seed_i)``; the first
``next()`` returns the initial latent noise, every later ``next()`` one more draw per image, so an image's stream does
not depend on which batch (or which GPU) it is generated in.  Subseed slerp and seed-resize keep the reference's
formulas (modules/rng.py:85-96, 131-143) on device tensors.
"""
from __future__ import annotations

import torch

from . import ops


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
    """modules/rng.py:85-96"""
    low_norm = low / torch.norm(low, dim=1, keepdim=True)
    high_norm = high / torch.norm(high, dim=1, keepdim=True)
    dot = (low_norm * high_norm).sum(1)
    if dot.mean() > 0.9995:
        return low * val + high * (1 - val)
    omega = torch.acos(dot)
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


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
        noise_shape = self.shape if self.seed_resize_from_h <= 0 or self.seed_resize_from_w <= 0 else \
            (self.shape[0], int(self.seed_resize_from_h) // 8, int(self.seed_resize_from_w // 8))
        xs = []
        for i, (seed, generator) in enumerate(zip(self.seeds, self.generators)):
            subnoise = None
            if self.subseeds is not None and self.subseed_strength != 0:
                subseed = 0 if i >= len(self.subseeds) else self.subseeds[i]
                subnoise = Generator(subseed, self.device).randn(noise_shape)
            if noise_shape != self.shape:
                noise = Generator(seed, self.device).randn(noise_shape)
            else:
                noise = generator.randn(self.shape)
            if subnoise is not None:
                noise = slerp(self.subseed_strength, noise, subnoise)
            if noise_shape != self.shape:
                x = generator.randn(self.shape)
                dx = (self.shape[2] - noise_shape[2]) // 2
                dy = (self.shape[1] - noise_shape[1]) // 2
                w = noise_shape[2] if dx >= 0 else noise_shape[2] + 2 * dx
                h = noise_shape[1] if dy >= 0 else noise_shape[1] + 2 * dy
                tx = 0 if dx < 0 else dx
                ty = 0 if dy < 0 else dy
                dx = max(-dx, 0)
                dy = max(-dy, 0)
                x[:, ty:ty + h, tx:tx + w] = noise[:, dy:dy + h, dx:dx + w]
                noise = x
            xs.append(noise)
        if self.eta_noise_seed_delta:
            self.generators = [Generator(seed + self.eta_noise_seed_delta, self.device) for seed in self.seeds]
        return torch.stack(xs)

    def next(self):
        if self.is_first:
            self.is_first = False
            return self.first()
        return torch.stack([g.randn(self.shape) for g in self.generators])
