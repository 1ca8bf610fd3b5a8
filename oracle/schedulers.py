"""fp32 CPU restatement of the noise schedulers (oracle; tests only).

In-repo functions restated from /root/reference/modules/sd_schedulers.py (file:line on each function) and pinned by
tests/golden/schedulers.npz, which tests/golden/make_golden.py produces by EXECUTING that reference file (with this
package's CompVisDenoiser standing in for k-diffusion's, the only third-party object the functions touch).
Third-party (crowsonkb/k-diffusion @ ab527a9, sampling.py — not on disk; restated from the published code, anchored on
the scheduler table at modules/sd_schedulers.py:130-143): get_sigmas_karras (oracle/kdiffusion.py),
get_sigmas_exponential, get_sigmas_polyexponential.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .kdiffusion import append_zero, get_sigmas_karras


def get_sigmas_exponential(n, sigma_min, sigma_max):
    sigmas = torch.linspace(math.log(sigma_max), math.log(sigma_min), n).exp()
    return append_zero(sigmas)


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0):
    ramp = torch.linspace(1, 0, n) ** rho
    sigmas = torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))
    return append_zero(sigmas)


def uniform(n, sigma_min, sigma_max, inner_model):                       # sd_schedulers.py:27-28
    return inner_model.get_sigmas(n)


def sgm_uniform(n, sigma_min, sigma_max, inner_model):                   # sd_schedulers.py:31-39
    start = inner_model.sigma_to_t(torch.tensor(sigma_max))
    end = inner_model.sigma_to_t(torch.tensor(sigma_min))
    sigs = [inner_model.t_to_sigma(ts) for ts in torch.linspace(start, end, n + 1)[:-1]]
    sigs += [0.0]
    return torch.FloatTensor(sigs)


AYS_SD15 = [14.615, 6.475, 3.861, 2.697, 1.886, 1.396, 0.963, 0.652, 0.399, 0.152, 0.029]
AYS_SDXL = [14.615, 6.315, 3.771, 2.181, 1.342, 0.862, 0.555, 0.380, 0.234, 0.113, 0.029]


def align_your_steps(n, sigma_min, sigma_max, is_sdxl=False):            # sd_schedulers.py:42-68
    def loglinear_interp(t_steps, num_steps):
        xs = np.linspace(0, 1, len(t_steps))
        ys = np.log(t_steps[::-1])
        new_xs = np.linspace(0, 1, num_steps)
        new_ys = np.interp(new_xs, xs, ys)
        return np.exp(new_ys)[::-1].copy()
    sigmas = list(AYS_SDXL if is_sdxl else AYS_SD15)
    if n != len(sigmas):
        sigmas = np.append(loglinear_interp(sigmas, n), [0.0])
    else:
        sigmas.append(0.0)
    return torch.FloatTensor(sigmas)


def kl_optimal(n, sigma_min, sigma_max):                                 # sd_schedulers.py:71-76
    alpha_min = torch.arctan(torch.tensor(sigma_min))
    alpha_max = torch.arctan(torch.tensor(sigma_max))
    step_indices = torch.arange(n + 1)
    return torch.tan(step_indices / n * alpha_min + (1.0 - step_indices / n) * alpha_max)


def simple_scheduler(n, sigma_min, sigma_max, inner_model):              # sd_schedulers.py:79-85
    sigs = []
    ss = len(inner_model.sigmas) / n
    for x in range(n):
        sigs += [float(inner_model.sigmas[-(1 + int(x * ss))])]
    sigs += [0.0]
    return torch.FloatTensor(sigs)


def normal_scheduler(n, sigma_min, sigma_max, inner_model, sgm=False):   # sd_schedulers.py:88-103
    start = inner_model.sigma_to_t(torch.tensor(sigma_max))
    end = inner_model.sigma_to_t(torch.tensor(sigma_min))
    timesteps = torch.linspace(start, end, n + 1)[:-1] if sgm else torch.linspace(start, end, n)
    sigs = [inner_model.t_to_sigma(ts) for ts in timesteps]
    sigs += [0.0]
    return torch.FloatTensor(sigs)


def ddim_scheduler(n, sigma_min, sigma_max, inner_model):                # sd_schedulers.py:106-115
    sigs = []
    ss = max(len(inner_model.sigmas) // n, 1)
    x = 1
    while x < len(inner_model.sigmas):
        sigs += [float(inner_model.sigmas[x])]
        x += ss
    sigs = sigs[::-1]
    sigs += [0.0]
    return torch.FloatTensor(sigs)


def beta_scheduler(n, sigma_min, sigma_max, inner_model, alpha=0.6, beta=0.6):   # sd_schedulers.py:118-127; defaults shared_options.py
    from scipy import stats
    timesteps = 1 - np.linspace(0, 1, n)
    timesteps = [stats.beta.ppf(x, alpha, beta) for x in timesteps]
    sigmas = [sigma_min + (x * (sigma_max - sigma_min)) for x in timesteps]
    sigmas += [0.0]
    return torch.FloatTensor(sigmas)


SCHEDULERS = {
    'uniform': (uniform, True), 'karras': (get_sigmas_karras, False), 'exponential': (get_sigmas_exponential, False),
    'polyexponential': (get_sigmas_polyexponential, False), 'sgm_uniform': (sgm_uniform, True), 'kl_optimal': (kl_optimal, False),
    'align_your_steps': (align_your_steps, False), 'simple': (simple_scheduler, True), 'normal': (normal_scheduler, True),
    'ddim': (ddim_scheduler, True), 'beta': (beta_scheduler, True),
}
