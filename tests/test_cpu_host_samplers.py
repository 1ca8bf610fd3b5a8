"""The product's sampler functions (stable-diffusion-webui_amd/sd_samplers.py) on the CPU, against the oracle's op-by-op restatements.

The samplers keep every tensor on the device and launch fused HIP kernels for the updates; what runs on the HOST — and what this file
checks without a GPU — is everything else: the step plans, the ancestral / SDE step sizes, the DPM-Solver and multistep coefficients that
are folded into one `sdmi_lincomb` per update, the order in which noise is drawn, the callbacks.  For these tests only, the device launches
are replaced by the same elementwise arithmetic in torch, taken from the C ABI's own contract (include/sdmi.h: sdmi_euler_step,
sdmi_dpmpp2m_step, sdmi_ddim_step, sdmi_axpby, sdmi_lincomb, sdmi_dpm_error_partials); the kernels themselves are compared with the oracle
on the GPU (tests/test_gpu_ops.py, tests/test_gpu_models.py).  Nothing here touches libsdmi.so's compute entry points.
"""
import importlib
import math

import pytest
import torch

from oracle import kdiffusion as okd
from tests.helpers import seeded

PKG = "stable-diffusion-webui_amd"


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


class _TorchStepKernels:
    """The elementwise contracts of include/sdmi.h, in place on CPU tensors (`ptr` is the identity under the fixture below)."""

    @staticmethod
    def sdmi_euler_step(x, den, noise, sigma, sigma_down, sigma_up, s_noise, n, stream):
        d = (x - den) / sigma
        x.add_(d * (sigma_down - sigma))
        if noise is not None:
            x.add_(noise * (s_noise * sigma_up))
        return 0

    @staticmethod
    def sdmi_dpmpp2m_step(x, den, old, ratio, em1, c1, c2, n, stream):
        dd = c1 * den - (c2 * old if old is not None else 0.0)
        x.copy_(ratio * x - em1 * dd)
        return 0

    @staticmethod
    def sdmi_ddim_step(x, e_t, noise, pred_x0, a_t, a_prev, sigma_t, sqrt_one_minus_at, n, stream):
        p0 = (x - sqrt_one_minus_at * e_t) / math.sqrt(a_t)
        out = math.sqrt(a_prev) * p0 + math.sqrt(max(1.0 - a_prev - sigma_t ** 2, 0.0)) * e_t
        if noise is not None:
            out = out + sigma_t * noise
        if pred_x0 is not None:
            pred_x0.copy_(p0)
        x.copy_(out)
        return 0

    @staticmethod
    def sdmi_axpby(y, x, a, z, b, n, stream):
        y.copy_(a * x + (b * z if z is not None else 0.0))
        return 0

    @staticmethod
    def sdmi_dpm_error_partials(lo, hi, prev, atol, rtol, partial, n, stream):
        delta = torch.maximum(torch.full_like(lo, atol), rtol * torch.maximum(lo.abs(), prev.abs()))
        partial.zero_()
        partial[0] = (((lo - hi) / delta) ** 2).double().sum().float()
        return 0


@pytest.fixture()
def ss(monkeypatch):
    mod = sub("sd_samplers")
    monkeypatch.setattr(mod, "lib", _TorchStepKernels())
    monkeypatch.setattr(mod, "ptr", lambda t: t)
    monkeypatch.setattr(mod, "stream_ptr", lambda: None)
    monkeypatch.setattr(mod, "_lc", lambda out, terms, coefs: out.copy_(sum(float(c) * t for c, t in zip(coefs, terms))))
    return mod


def model(x, sigma, **kw):
    """An analytic denoiser with a non-linear part (so that a wrong evaluation point or coefficient shows), smooth in x and sigma."""
    s = sigma[:, None, None, None]
    return x / (1 + s * s) + torch.tanh(0.5 * x) * (s * s / (1 + s * s)) * 0.3


SHAPE = (2, 4, 8, 8)


def karras(n, smin=0.03, smax=14.6, zero=True):
    s = okd.get_sigmas_karras(n, smin, smax)
    return s if zero else s[:-1]


def noise_pair(seed0, count=64):
    a = iter([seeded(SHAPE, seed0 + i) for i in range(count)])
    b = iter([seeded(SHAPE, seed0 + i) for i in range(count)])
    return a, b


def close(got, want, tol=2e-5):
    return float((got - want).norm() / want.norm()) < tol


# name, oracle call (model, x, sigmas, noise_fn / noise_sampler), product call (ss, model, x, sigmas, noise_sampler)
KD = [
    ("euler", lambda m, x, s, n: okd.sample_euler(m, x, s, {}), lambda ss, m, x, s, n: ss.sample_euler(m, x, s)),
    ("euler s_churn", lambda m, x, s, n: okd.sample_euler(m, x, s, {}, lambda: next(n), s_churn=6.0, s_tmin=0.2, s_tmax=8.0, s_noise=1.003),
     lambda ss, m, x, s, n: ss.sample_euler(m, x, s, s_churn=6.0, s_tmin=0.2, s_tmax=8.0, s_noise=1.003, noise_sampler=lambda *a: next(n))),
    ("euler_ancestral", lambda m, x, s, n: okd.sample_euler_ancestral(m, x, s, {}, lambda: next(n), eta=0.8, s_noise=0.95),
     lambda ss, m, x, s, n: ss.sample_euler_ancestral(m, x, s, eta=0.8, s_noise=0.95, noise_sampler=lambda *a: next(n))),
    ("heun s_churn", lambda m, x, s, n: okd.sample_heun(m, x, s, {}, lambda: next(n), s_churn=4.0, s_noise=1.002),
     lambda ss, m, x, s, n: ss.sample_heun(m, x, s, s_churn=4.0, s_noise=1.002, noise_sampler=lambda *a: next(n))),
    ("dpm_2", lambda m, x, s, n: okd.sample_dpm_2(m, x, s, {}), lambda ss, m, x, s, n: ss.sample_dpm_2(m, x, s)),
    ("dpm_2_ancestral", lambda m, x, s, n: okd.sample_dpm_2_ancestral(m, x, s, {}, lambda: next(n), eta=0.9, s_noise=1.01),
     lambda ss, m, x, s, n: ss.sample_dpm_2_ancestral(m, x, s, eta=0.9, s_noise=1.01, noise_sampler=lambda *a: next(n))),
    ("lms", lambda m, x, s, n: okd.sample_lms(m, x, s, {}), lambda ss, m, x, s, n: ss.sample_lms(m, x, s)),
    ("dpmpp_2s_ancestral", lambda m, x, s, n: okd.sample_dpmpp_2s_ancestral(m, x, s, {}, lambda: next(n), eta=0.7, s_noise=0.98),
     lambda ss, m, x, s, n: ss.sample_dpmpp_2s_ancestral(m, x, s, eta=0.7, s_noise=0.98, noise_sampler=lambda *a: next(n))),
    ("dpmpp_2m", lambda m, x, s, n: okd.sample_dpmpp_2m(m, x, s, {}), lambda ss, m, x, s, n: ss.sample_dpmpp_2m(m, x, s)),
    ("dpmpp_sde", lambda m, x, s, n: okd.sample_dpmpp_sde(m, x, s, {}, lambda *a: next(n), eta=0.9, s_noise=1.01),
     lambda ss, m, x, s, n: ss.sample_dpmpp_sde(m, x, s, eta=0.9, s_noise=1.01, noise_sampler=lambda *a: next(n))),
    ("dpmpp_2m_sde midpoint", lambda m, x, s, n: okd.sample_dpmpp_2m_sde(m, x, s, {}, lambda *a: next(n), eta=0.8, s_noise=0.97),
     lambda ss, m, x, s, n: ss.sample_dpmpp_2m_sde(m, x, s, eta=0.8, s_noise=0.97, noise_sampler=lambda *a: next(n))),
    ("dpmpp_2m_sde heun", lambda m, x, s, n: okd.sample_dpmpp_2m_sde(m, x, s, {}, lambda *a: next(n), eta=1.0, solver_type="heun"),
     lambda ss, m, x, s, n: ss.sample_dpmpp_2m_sde(m, x, s, eta=1.0, solver_type="heun", noise_sampler=lambda *a: next(n))),
    ("dpmpp_3m_sde", lambda m, x, s, n: okd.sample_dpmpp_3m_sde(m, x, s, {}, lambda *a: next(n), eta=0.9, s_noise=1.02),
     lambda ss, m, x, s, n: ss.sample_dpmpp_3m_sde(m, x, s, eta=0.9, s_noise=1.02, noise_sampler=lambda *a: next(n))),
    ("lcm", lambda m, x, s, n: okd.sample_lcm(m, x, s, {}, lambda: next(n)),
     lambda ss, m, x, s, n: ss.sample_lcm(m, x, s, noise_sampler=lambda *a: next(n))),
    ("restart", lambda m, x, s, n: okd.restart_sampler(m, x, s, {}, lambda: next(n), s_noise=1.01),
     lambda ss, m, x, s, n: ss.restart_sampler(m, x, s, s_noise=1.01, noise_sampler=lambda *a: next(n))),
]


@pytest.mark.parametrize("name,oracle_call,host_call", KD, ids=[k[0] for k in KD])
@pytest.mark.parametrize("steps", [1, 2, 9])
def test_host_kdiffusion_samplers_match_the_oracle_restatements(ss, name, oracle_call, host_call, steps):
    """Same schedule, same denoiser, same noise sequence: the product's folded coefficients reproduce the oracle's op-by-op loop, the
    callbacks fire once per step with the oracle's (i, sigma, sigma_hat), and the number of noise draws is the same (a sampler that
    drew one tensor more or less would shift every later image of a batch: the webui's seeds are consumed in order)."""
    sig = karras(steps)
    x0 = seeded(SHAPE, 4100 + steps) * sig[0]
    n1, n2 = noise_pair(5000)
    want = oracle_call(model, x0.clone(), sig, n1)
    got = host_call(ss, model, x0.clone(), sig, n2)
    assert close(got, want), (name, steps, float((got - want).norm() / want.norm()))
    # both noise iterators advanced equally far
    assert torch.equal(next(n1), next(n2)), f"{name}: the product drew a different number of noise tensors than the oracle"


@pytest.mark.parametrize("name", ["euler_ancestral", "dpmpp_2m", "heun s_churn", "dpm_2_ancestral", "dpmpp_sde", "dpmpp_3m_sde", "restart"])
def test_host_sampler_callbacks_match_the_oracle(ss, name):
    entry = next(k for k in KD if k[0] == name)
    sig = karras(6)
    x0 = seeded(SHAPE, 4200) * sig[0]
    rec = {"o": [], "h": []}

    def hook(key):
        return lambda d: rec[key].append((int(d["i"]), float(d["sigma"]), float(d["sigma_hat"]), float(d["denoised"].double().sum())))
    n1, n2 = noise_pair(5100)
    # the table's lambdas take no callback: call the functions directly with one
    fn_o = {"euler_ancestral": lambda: okd.sample_euler_ancestral(model, x0.clone(), sig, {}, lambda: next(n1), callback=hook("o")),
            "dpmpp_2m": lambda: okd.sample_dpmpp_2m(model, x0.clone(), sig, {}, callback=hook("o")),
            "heun s_churn": lambda: okd.sample_heun(model, x0.clone(), sig, {}, lambda: next(n1), callback=hook("o"), s_churn=4.0),
            "dpm_2_ancestral": lambda: okd.sample_dpm_2_ancestral(model, x0.clone(), sig, {}, lambda: next(n1), callback=hook("o")),
            "dpmpp_sde": lambda: okd.sample_dpmpp_sde(model, x0.clone(), sig, {}, lambda *a: next(n1), callback=hook("o")),
            "dpmpp_3m_sde": lambda: okd.sample_dpmpp_3m_sde(model, x0.clone(), sig, {}, lambda *a: next(n1), callback=hook("o")),
            "restart": lambda: okd.restart_sampler(model, x0.clone(), sig, {}, lambda: next(n1), callback=hook("o"))}[name]
    ns = lambda *a: next(n2)   # noqa: E731
    fn_h = {"euler_ancestral": lambda: ss.sample_euler_ancestral(model, x0.clone(), sig, callback=hook("h"), noise_sampler=ns),
            "dpmpp_2m": lambda: ss.sample_dpmpp_2m(model, x0.clone(), sig, callback=hook("h")),
            "heun s_churn": lambda: ss.sample_heun(model, x0.clone(), sig, callback=hook("h"), s_churn=4.0, noise_sampler=ns),
            "dpm_2_ancestral": lambda: ss.sample_dpm_2_ancestral(model, x0.clone(), sig, callback=hook("h"), noise_sampler=ns),
            "dpmpp_sde": lambda: ss.sample_dpmpp_sde(model, x0.clone(), sig, callback=hook("h"), noise_sampler=ns),
            "dpmpp_3m_sde": lambda: ss.sample_dpmpp_3m_sde(model, x0.clone(), sig, callback=hook("h"), noise_sampler=ns),
            "restart": lambda: ss.restart_sampler(model, x0.clone(), sig, callback=hook("h"), noise_sampler=ns)}[name]
    fn_o(); fn_h()
    assert len(rec["o"]) == len(rec["h"]) > 0 and entry is not None
    for (io, so, sho, do), (ih, sh, shh, dh) in zip(rec["o"], rec["h"]):
        assert io == ih and abs(so - sh) <= 1e-6 * abs(so) and abs(sho - shh) <= 1e-6 * abs(sho), (name, io, so, sh, sho, shh)
        assert abs(do - dh) <= 2e-4 * max(1.0, abs(do)), (name, io, do, dh)


def test_host_dpm_adaptive_matches_the_oracle_step_for_step(ss):
    """The adaptive solver's accept / reject sequence is data dependent: the product's error measure (256 partial sums added on the
    host) and PID controller must take the oracle's decisions, or the two runs evaluate the model at different points."""
    smin, smax = 0.03, 14.6
    for order, rtol, eta in ((3, 0.05, 0.0), (2, 0.05, 0.0), (3, 0.02, 0.6)):
        x0 = seeded(SHAPE, 4300 + order) * smax
        n1, n2 = noise_pair(5200)
        want, wi = okd.sample_dpm_adaptive(model, x0.clone(), smin, smax, {}, lambda *a: next(n1), eta=eta, order=order, rtol=rtol, return_info=True)
        got, gi = ss.sample_dpm_adaptive(model, x0.clone(), smin, smax, order=order, rtol=rtol, eta=eta, noise_sampler=lambda *a: next(n2),
                                         return_info=True)
        assert {k: gi[k] for k in ("steps", "nfe", "n_accept", "n_reject")} == {k: wi[k] for k in ("steps", "nfe", "n_accept", "n_reject")}, (order, gi, wi)
        assert close(got, want, 5e-5), (order, rtol, eta)


def test_host_ddim_and_plms_match_the_pinned_oracle_loops(ss):
    """DDIM / PLMS walk the alphas table of the model wrapper; the oracle's loops are pinned to the reference's own
    (tests/golden/ddim.npz, plms.npz).  eps-model stand-in: e_t = f(x, t)."""
    acp = okd.make_alphas_cumprod()

    class Inner:                                             # what ddim() / plms() read from model.inner_model.inner_model
        alphas_cumprod = acp

    class Wrap:
        def __init__(self):
            self.inner_model = type("IM", (), {"inner_model": Inner})()

        def __call__(self, x, t, **kw):
            tt = t.float()[:, None, None, None] / 1000.0
            return 0.8 * x * tt + torch.tanh(x) * (1 - tt) * 0.2

    m = Wrap()
    for steps, eta in ((5, 0.0), (7, 0.5)):
        ts = okd.ddim_timesteps(steps)
        x0 = seeded(SHAPE, 4400 + steps)
        n1, n2 = noise_pair(5300)
        want = okd.sample_ddim(m, x0.clone(), ts, acp, {}, lambda: next(n1), eta=eta)
        got = ss.ddim(m, x0.clone(), ts, eta=eta, noise_sampler=lambda *a: next(n2))
        assert close(got, want), ("ddim", steps, eta, float((got - want).norm() / want.norm()))
    ts = okd.ddim_timesteps(8)
    x0 = seeded(SHAPE, 4500)
    want = okd.sample_plms(m, x0.clone(), ts, acp, {})
    got = ss.plms(m, x0.clone(), ts)
    assert close(got, want), ("plms", float((got - want).norm() / want.norm()))
