"""Shared test helpers (seeded inputs identical to tests/golden/make_golden.py)."""
import numpy as np
import torch


def seeded(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def seeded_module_weights(module, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in module.state_dict().items():
            if p.ndim >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_(torch.randn(p.shape, generator=g) * fan_in ** -0.5)
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.02 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def usable_cpus(cap=32):
    """Hardware threads this process may use (affinity mask; os.cpu_count() reports the host's even on a restricted box, and
    oversubscribing the oracle's fp32 GEMMs is pathological), capped where they stop scaling."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(cap, n))
