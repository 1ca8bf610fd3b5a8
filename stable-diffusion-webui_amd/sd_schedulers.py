"""Noise schedulers — host-side mirror of /root/reference/modules/sd_schedulers.py (same names, table, argument meaning).

Schedules are a few dozen floats computed once per job on the host, exactly as in the reference; the sigma table then
drives the device-side sampler steps.  The k-diffusion functions the reference table points at (get_sigmas_karras /
_exponential / _polyexponential, third-party) are restated here; every in-repo scheduler follows its reference function
line by line in arithmetic (cited) and is checked against tests/golden/schedulers.npz, produced by executing the
reference file."""
from __future__ import annotations

import dataclasses
import math

import numpy as np
import torch

from . import shared


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7., device='cpu'):
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return append_zero(sigmas).to(device)


def get_sigmas_exponential(n, sigma_min, sigma_max, device='cpu'):
    sigmas = torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp()
    return append_zero(sigmas)


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1., device='cpu'):
    ramp = torch.linspace(1, 0, n, device=device) ** rho
    sigmas = torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))
    return append_zero(sigmas)


@dataclasses.dataclass
class Scheduler:
    """One row of the scheduler table (modules/sd_schedulers.py:18-26, 130-143)."""
    name: str
    label: str
    function: any
    default_rho: float = -1
    need_inner_model: bool = False
    aliases: list = None


def _with_final_zero(values, device):
    """fp32 sigma table ending in the terminal 0 every sampler loop expects."""
    return torch.tensor([float(v) for v in values] + [0.0], dtype=torch.float32).to(device)


def _model_time_range(inner_model, sigma_min, sigma_max):
    return inner_model.sigma_to_t(torch.tensor(sigma_max)), inner_model.sigma_to_t(torch.tensor(sigma_min))


def uniform(n, sigma_min, sigma_max, inner_model, device):
    """:27-28 — the wrapped model's own table (uniform in its timestep index)."""
    return inner_model.get_sigmas(n).to(device)


def sgm_uniform(n, sigma_min, sigma_max, inner_model, device):
    """:31-39 — n + 1 uniformly spaced model times, the last one dropped."""
    t_hi, t_lo = _model_time_range(inner_model, sigma_min, sigma_max)
    return _with_final_zero([inner_model.t_to_sigma(t) for t in torch.linspace(t_hi, t_lo, n + 1)[:-1]], device)


AYS_SIGMAS_SD15 = (14.615, 6.475, 3.861, 2.697, 1.886, 1.396, 0.963, 0.652, 0.399, 0.152, 0.029)      # :60-63, arXiv 2404.14507
AYS_SIGMAS_SDXL = (14.615, 6.315, 3.771, 2.181, 1.342, 0.862, 0.555, 0.380, 0.234, 0.113, 0.029)


def get_align_your_steps_sigmas(n, sigma_min, sigma_max, device):
    """:42-68 — the published 11-point tables, log-linearly resampled when another step count is asked for."""
    table = AYS_SIGMAS_SDXL if getattr(shared.sd_model, "is_sdxl", False) else AYS_SIGMAS_SD15
    if n == len(table):
        return _with_final_zero(table, device)
    ascending_log = np.log(np.asarray(table, dtype=np.float64)[::-1])
    resampled = np.interp(np.linspace(0, 1, n), np.linspace(0, 1, len(table)), ascending_log)
    return torch.FloatTensor(np.append(np.exp(resampled)[::-1].copy(), [0.0])).to(device)


def kl_optimal(n, sigma_min, sigma_max, device):
    """:71-76 — tan of a linear blend of the two end angles, n + 1 points (no appended zero: the last point IS sigma_min)."""
    lo = torch.arctan(torch.tensor(sigma_min, device=device))
    hi = torch.arctan(torch.tensor(sigma_max, device=device))
    frac = torch.arange(n + 1, device=device) / n
    return torch.tan(frac * lo + (1.0 - frac) * hi)


def simple_scheduler(n, sigma_min, sigma_max, inner_model, device):
    """:79-85 — every (N / n)-th entry of the model's table, counted from the noisy end."""
    stride = len(inner_model.sigmas) / n
    return _with_final_zero([inner_model.sigmas[-(1 + int(k * stride))] for k in range(n)], device)


def normal_scheduler(n, sigma_min, sigma_max, inner_model, device, sgm=False, floor=False):
    """:88-103 — uniformly spaced model times (sgm: one more point, last dropped)."""
    t_hi, t_lo = _model_time_range(inner_model, sigma_min, sigma_max)
    times = torch.linspace(t_hi, t_lo, n + 1)[:-1] if sgm else torch.linspace(t_hi, t_lo, n)
    return _with_final_zero([inner_model.t_to_sigma(t) for t in times], device)


def ddim_scheduler(n, sigma_min, sigma_max, inner_model, device):
    """:106-115 — table entries 1, 1 + s, 1 + 2s, ... with s = max(N // n, 1), noisy end first."""
    stride = max(len(inner_model.sigmas) // n, 1)
    picked = [inner_model.sigmas[k] for k in range(1, len(inner_model.sigmas), stride)]
    return _with_final_zero(picked[::-1], device)


def beta_scheduler(n, sigma_min, sigma_max, inner_model, device):
    """:118-127 — quantiles of Beta(opts.beta_dist_alpha, opts.beta_dist_beta) mapped linearly onto [sigma_min, sigma_max]
    (arXiv 2407.12173)."""
    from scipy import stats
    a, b = shared.opts.beta_dist_alpha, shared.opts.beta_dist_beta
    quantiles = [stats.beta.ppf(q, a, b) for q in 1 - np.linspace(0, 1, n)]
    return _with_final_zero([sigma_min + (q * (sigma_max - sigma_min)) for q in quantiles], device)


schedulers = [                                                           # the table of :130-143, same order
    Scheduler('automatic', 'Automatic', None),
    Scheduler('uniform', 'Uniform', uniform, need_inner_model=True),
    Scheduler('karras', 'Karras', get_sigmas_karras, default_rho=7.0),
    Scheduler('exponential', 'Exponential', get_sigmas_exponential),
    Scheduler('polyexponential', 'Polyexponential', get_sigmas_polyexponential, default_rho=1.0),
    Scheduler('sgm_uniform', 'SGM Uniform', sgm_uniform, need_inner_model=True, aliases=["SGMUniform"]),
    Scheduler('kl_optimal', 'KL Optimal', kl_optimal),
    Scheduler('align_your_steps', 'Align Your Steps', get_align_your_steps_sigmas),
    Scheduler('simple', 'Simple', simple_scheduler, need_inner_model=True),
    Scheduler('normal', 'Normal', normal_scheduler, need_inner_model=True),
    Scheduler('ddim', 'DDIM', ddim_scheduler, need_inner_model=True),
    Scheduler('beta', 'Beta', beta_scheduler, need_inner_model=True),
]

schedulers_map = {**{x.name: x for x in schedulers}, **{x.label: x for x in schedulers}}
