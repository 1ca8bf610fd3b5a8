"""Host-side Brownian-tree noise source of the SDE samplers (DPM++ SDE / 2M SDE / 2M SDE Heun / 3M SDE).

Replaces ``k_diffusion.sampling.BrownianTreeNoiseSampler`` as the reference builds it in
modules/sd_samplers_common.py:334-342 — one ``torchsde.BrownianTree`` per image seed, so that an image's noise depends on its own
seed and the queried (sigma, sigma_next) interval only: independent of the batch it sits in (the point of the reference's comment
"deterministic results across different batch sizes") and of the rank it lands on (parallel.process_images_sharded).

torchsde / k-diffusion are third-party packages that are neither vendored in the reference nor installed here, so the tree is built
from the published construction (Levy's Brownian-bridge refinement over the dyadic tree of [t0, t1], every node seeded from numpy's
``SeedSequence(entropy=seed).spawn`` hierarchy — what torchsde's BrownianInterval uses for the same purpose), not bit-for-bit from
torchsde's code.  Draws come from ``torch.randn`` on a CPU generator, as in the reference's CPU configuration; the [B, C, h, w] fp32
result is copied to the engine's device (a few KB per sampler step — the update that consumes it is an sdmi_lincomb launch).

Iterative implementation with a per-tree cache of visited nodes (the sampler queries adjacent intervals, which share most of their
root paths).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

_TOL = 1e-6
_POOL = 24


def _draw(shape, entropy, key: Tuple[int, ...]) -> torch.Tensor:
    ss = np.random.SeedSequence(entropy=entropy, spawn_key=key, pool_size=_POOL)
    g = torch.Generator(device="cpu").manual_seed(int(ss.generate_state(1, dtype=np.uint64)[0] & 0x7FFFFFFFFFFFFFFF))
    return torch.randn(shape, generator=g, dtype=torch.float32)


class _Tree:
    """W on [t0, t1] for one seed.  A node is named by its spawn key: () is the root, key + (0,) / (1,) its left / right half,
    key + (2,) the generator of its bridge variate."""

    def __init__(self, t0: float, t1: float, shape, entropy: int):
        self.t0, self.t1, self.shape, self.entropy = float(t0), float(t1), tuple(shape), int(entropy)
        self.cache: Dict[Tuple[int, ...], torch.Tensor] = {(): _draw(self.shape, self.entropy, ()) * math.sqrt(self.t1 - self.t0)}

    def _left_increment(self, key, a, b, m, cache=True) -> torch.Tensor:
        """W(m) - W(a) of node `key` = [a, b] (its own increment is cached under `key`)."""
        ck = key + (0,)
        if cache and ck in self.cache:
            return self.cache[ck]
        w_ab = self.cache[key]
        xi = _draw(self.shape, self.entropy, key + (2,))
        w_am = w_ab * ((m - a) / (b - a)) + math.sqrt((m - a) * (b - m) / (b - a)) * xi
        if cache:
            self.cache[ck] = w_am
            self.cache[key + (1,)] = w_ab - w_am
        return w_am

    def w_to(self, t: float) -> torch.Tensor:
        t = min(max(float(t), self.t0), self.t1)
        a, b, key = self.t0, self.t1, ()
        acc = torch.zeros(self.shape)
        while True:
            if t <= a:
                return acc
            if t >= b:
                return acc + self.cache[key]
            if (b - a) < _TOL:                               # leaf: place t itself by one more bridge step (not cached: t is arbitrary)
                return acc + self._left_increment(key, a, b, t, cache=False)
            m = 0.5 * (a + b)
            w_am = self._left_increment(key, a, b, m)
            if t < m:
                key, b = key + (0,), m
            else:
                acc = acc + w_am
                key, a = key + (1,), m

    def increment(self, ta: float, tb: float) -> torch.Tensor:
        return self.w_to(tb) - self.w_to(ta)


class BrownianTreeNoiseSampler:
    """``noise_sampler(sigma, sigma_next)`` -> unit-variance noise [B, C, h, w] on ``x``'s device."""

    def __init__(self, x: torch.Tensor, sigma_min, sigma_max, seed: List[int]):
        t0, t1 = float(sigma_min), float(sigma_max)
        if t0 > t1:
            t0, t1 = t1, t0
        seeds = list(seed) if isinstance(seed, (list, tuple)) else [seed]
        if len(seeds) != x.shape[0]:
            raise ValueError(f"{len(seeds)} seeds for a batch of {x.shape[0]}")
        self.device = x.device
        self.trees = [_Tree(t0, t1, tuple(x.shape[1:]), s) for s in seeds]

    def __call__(self, sigma, sigma_next) -> torch.Tensor:
        ta, tb = float(sigma), float(sigma_next)
        sign = 1.0
        if ta > tb:
            ta, tb, sign = tb, ta, -1.0
        w = torch.stack([tree.increment(ta, tb) for tree in self.trees]) * (sign / math.sqrt(abs(tb - ta)))
        return w.to(self.device).contiguous()
