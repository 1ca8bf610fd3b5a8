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


def _table(values):
    """fp32 sigmas followed by the terminal zero."""
    return torch.tensor([float(v) for v in values] + [0.0], dtype=torch.float32)


def _t_range(inner_model, sigma_min, sigma_max):
    return inner_model.sigma_to_t(torch.tensor(sigma_max)), inner_model.sigma_to_t(torch.tensor(sigma_min))


def uniform(n, sigma_min, sigma_max, inner_model):
    """sd_schedulers.py:27-28"""
    return inner_model.get_sigmas(n)


def sgm_uniform(n, sigma_min, sigma_max, inner_model):
    """sd_schedulers.py:31-39: n + 1 evenly spaced timesteps from t(sigma_max) to t(sigma_min), the last dropped."""
    hi, lo = _t_range(inner_model, sigma_min, sigma_max)
    return _table(inner_model.t_to_sigma(t) for t in torch.linspace(hi, lo, n + 1)[:-1])


AYS_SD15 = [14.615, 6.475, 3.861, 2.697, 1.886, 1.396, 0.963, 0.652, 0.399, 0.152, 0.029]
AYS_SDXL = [14.615, 6.315, 3.771, 2.181, 1.342, 0.862, 0.555, 0.380, 0.234, 0.113, 0.029]


def align_your_steps(n, sigma_min, sigma_max, is_sdxl=False):
    """sd_schedulers.py:42-68: the 11 published sigmas; other step counts by linear interpolation of log sigma over [0, 1]."""
    base = AYS_SDXL if is_sdxl else AYS_SD15
    if n == len(base):
        return _table(base)
    log_up = np.log(np.array(base)[::-1])
    picked = np.exp(np.interp(np.linspace(0, 1, n), np.linspace(0, 1, len(base)), log_up))[::-1]
    return torch.FloatTensor(np.append(picked.copy(), [0.0]))


def kl_optimal(n, sigma_min, sigma_max):
    """sd_schedulers.py:71-76: sigma_i = tan((i/n) atan(sigma_min) + (1 - i/n) atan(sigma_max)), i = 0..n."""
    w = torch.arange(n + 1) / n
    return torch.tan(w * torch.arctan(torch.tensor(sigma_min)) + (1.0 - w) * torch.arctan(torch.tensor(sigma_max)))


def simple_scheduler(n, sigma_min, sigma_max, inner_model):
    """sd_schedulers.py:79-85"""
    step = len(inner_model.sigmas) / n
    return _table(inner_model.sigmas[-(1 + int(i * step))] for i in range(n))


def normal_scheduler(n, sigma_min, sigma_max, inner_model, sgm=False):
    """sd_schedulers.py:88-103"""
    hi, lo = _t_range(inner_model, sigma_min, sigma_max)
    ts = torch.linspace(hi, lo, n + 1)[:-1] if sgm else torch.linspace(hi, lo, n)
    return _table(inner_model.t_to_sigma(t) for t in ts)


def ddim_scheduler(n, sigma_min, sigma_max, inner_model):
    """sd_schedulers.py:106-115"""
    step = max(len(inner_model.sigmas) // n, 1)
    return _table(reversed([inner_model.sigmas[i] for i in range(1, len(inner_model.sigmas), step)]))


def beta_scheduler(n, sigma_min, sigma_max, inner_model, alpha=0.6, beta=0.6):
    """sd_schedulers.py:118-127 with the option defaults of modules/shared_options.py:408-409."""
    from scipy import stats
    return _table(sigma_min + (stats.beta.ppf(q, alpha, beta) * (sigma_max - sigma_min)) for q in 1 - np.linspace(0, 1, n))


SCHEDULERS = {
    'uniform': (uniform, True), 'karras': (get_sigmas_karras, False), 'exponential': (get_sigmas_exponential, False),
    'polyexponential': (get_sigmas_polyexponential, False), 'sgm_uniform': (sgm_uniform, True), 'kl_optimal': (kl_optimal, False),
    'align_your_steps': (align_your_steps, False), 'simple': (simple_scheduler, True), 'normal': (normal_scheduler, True),
    'ddim': (ddim_scheduler, True), 'beta': (beta_scheduler, True),
}
