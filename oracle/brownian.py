"""Brownian-tree noise source of the SDE samplers (oracle; tests only).

The reference builds ``k_diffusion.sampling.BrownianTreeNoiseSampler(x, sigma_min, sigma_max, seed=<this batch's seeds>)``
(/root/reference/modules/sd_samplers_common.py:334-342), one ``torchsde.BrownianTree`` per image seed, so that the noise of an image
depends only on its own seed and on the (sigma, sigma_next) pair asked for — not on the batch it is in, nor on how the interval was
subdivided by earlier queries.  ``torchsde`` (pinned 0.2.6 in the reference's requirements_versions.txt) and ``k_diffusion`` are not
installed here and not vendored: PARITY UNPINNED.  What is restated is the published construction:

  * W is a Brownian motion on [t0, t1] (t = the sampler's transformed sigma), W(t0) = 0, W(t1) ~ N(0, t1 - t0);
  * values inside are filled in by Levy's Brownian-bridge construction on the dyadic tree of the interval: for a node [a, b] with
    increment W_ab and midpoint m:  W_am = W_ab (m - a)/(b - a) + sqrt((m - a)(b - m)/(b - a)) xi,  W_mb = W_ab - W_am;
  * every node draws its xi from its own generator, seeded from numpy's ``SeedSequence(entropy=seed, pool_size=24)`` spawn tree
    (children = ``spawn(2)``), so a value is a pure function of (seed, position in the tree);
  * the tree is descended until the node is narrower than ``tol`` = 1e-6; the query point inside that leaf is placed by one more
    bridge step (from the leaf's own third spawn), which keeps every query a pure function of (seed, t);
  * ``BrownianTreeNoiseSampler(sigma, sigma_next) = (W(t_next) - W(t)) / sqrt(|t_next - t|)`` — unit-variance noise.
Draws are ``torch.randn`` on a CPU ``torch.Generator`` (the reference's CPU configuration draws on the CPU generator as well).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _randn(shape, seedseq) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(int(seedseq.generate_state(1, dtype=np.uint64)[0] & 0x7FFFFFFFFFFFFFFF))
    return torch.randn(shape, generator=g, dtype=torch.float32)


class BrownianTree:
    def __init__(self, t0: float, t1: float, shape, entropy: int, tol: float = 1e-6, pool_size: int = 24):
        assert t1 > t0
        self.t0, self.t1, self.shape, self.tol = float(t0), float(t1), tuple(shape), tol
        root = np.random.SeedSequence(entropy=int(entropy), pool_size=pool_size)
        self._root = (root, _randn(self.shape, root) * math.sqrt(self.t1 - self.t0))

    def _w_to(self, t: float) -> torch.Tensor:
        """W(t) - W(t0) for t in [t0, t1]."""
        t = min(max(float(t), self.t0), self.t1)
        a, b = self.t0, self.t1
        seq, w_ab = self._root
        acc = torch.zeros(self.shape)
        while True:
            if t <= a:
                return acc
            if t >= b:
                return acc + w_ab
            leaf = (b - a) < self.tol
            # children of a node are a pure function of its (entropy, spawn_key): spawn from a fresh copy, never from a shared object
            kids = np.random.SeedSequence(entropy=seq.entropy, spawn_key=seq.spawn_key, pool_size=seq.pool_size).spawn(3)
            m = t if leaf else 0.5 * (a + b)
            xi = _randn(self.shape, kids[2])
            w_am = w_ab * ((m - a) / (b - a)) + math.sqrt((m - a) * (b - m) / (b - a)) * xi
            if leaf:
                return acc + w_am
            if t < m:
                seq, w_ab, b = kids[0], w_am, m
            else:
                acc = acc + w_am
                seq, w_ab, a = kids[1], w_ab - w_am, m

    def __call__(self, ta: float, tb: float) -> torch.Tensor:
        return self._w_to(tb) - self._w_to(ta)


class BrownianTreeNoiseSampler:
    """k-diffusion's class of the same name with ``transform = identity`` and one tree per seed (BatchedBrownianTree)."""

    def __init__(self, x: torch.Tensor, sigma_min, sigma_max, seed):
        t0, t1 = float(sigma_min), float(sigma_max)
        self.sign = 1.0
        if t0 > t1:
            t0, t1, self.sign = t1, t0, -1.0
        seeds = list(seed) if isinstance(seed, (list, tuple)) else [seed]
        assert len(seeds) == x.shape[0]
        self.trees = [BrownianTree(t0, t1, x.shape[1:], s) for s in seeds]

    def __call__(self, sigma, sigma_next) -> torch.Tensor:
        ta, tb = float(sigma), float(sigma_next)
        sign = 1.0
        if ta > tb:
            ta, tb, sign = tb, ta, -1.0
        w = torch.stack([tree(ta, tb) for tree in self.trees]) * (self.sign * sign)
        return w / math.sqrt(abs(tb - ta))
