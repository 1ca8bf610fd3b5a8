"""Known-answer PROPERTY tests of the third-party arithmetic the oracle restates without a pin (k-diffusion samplers, torchsde's
Brownian tree: neither package is vendored in /root/reference nor installed here — oracle/kdiffusion.py, oracle/brownian.py headers).

What can be checked without the packages is the mathematics they implement.  For Gaussian data x0 ~ N(0, S^2) the ideal denoiser is
D(x, sigma) = x S^2 / (S^2 + sigma^2), the probability-flow ODE dx/dsigma = (x - D) / sigma has the closed-form solution
x(sigma) = x(sigma0) sqrt((S^2 + sigma^2) / (S^2 + sigma0^2)), and the reverse SDE keeps the marginal N(0, S^2 + sigma^2) at every
noise level.  So:
  * every deterministic sampler must converge to the closed form, at the order its paper states (Euler 1; Heun, DPM2, DPM++ 2M / 2S,
    DPM++ SDE family at eta = 0: 2; LMS(4) > 3): a wrong coefficient in a higher-order correction drops the measured order to 1;
  * every stochastic sampler must reproduce the marginal variance S^2 + sigma_end^2, with an error that shrinks with the step count;
    DPM++ SDE does so at second order only when its two noise queries of a step come from ONE Brownian path (the tree), which checks
    the tree's consistency over nested intervals at the same time;
  * the tree itself must be a Brownian motion (unit-variance normalised increments, additive over adjacent intervals, uncorrelated
    over disjoint ones), a pure function of (seed, interval), and the product's host implementation (brownian.py, iterative with a
    node cache) must return the oracle's values (recursive-descent restatement) to fp32 rounding.
These tests pin nothing to the packages' outputs — the rows stay "unpinned" in DESIGN.md section 2 — but a transcription error in a
restated sampler fails them.
"""
import importlib
import math

import pytest
import torch

from oracle import kdiffusion as K
from oracle.brownian import BrownianTreeNoiseSampler as OracleTreeSampler

PKG = "stable-diffusion-webui_amd"
S = 0.7                                   # data standard deviation
SMAX, SMIN = 10.0, 0.1


def model(x, sigma, **kw):
    s = sigma.view(-1, *([1] * (x.ndim - 1))).to(x.dtype)
    return x * (S * S / (S * S + s * s))


def flow(x0, s0, s1):
    return x0 * math.sqrt((S * S + s1 * s1) / (S * S + s0 * s0))


def sigmas(n):
    return torch.exp(torch.linspace(math.log(SMAX), math.log(SMIN), n + 1, dtype=torch.float64))


X0 = (torch.linspace(-3, 3, 16, dtype=torch.float64).view(4, 4) * 10)
ZERO = lambda *a: torch.zeros_like(X0)   # noqa: E731

DETERMINISTIC = [
    # name, call(sigmas), minimum measured order between n = 32 -> 64 -> 128
    ("euler", lambda s: K.sample_euler(model, X0, s, {}), 0.9),
    ("euler_ancestral eta=0", lambda s: K.sample_euler_ancestral(model, X0, s, {}, ZERO, eta=0.0), 0.9),
    ("heun", lambda s: K.sample_heun(model, X0, s, {}), 1.9),
    ("dpm_2", lambda s: K.sample_dpm_2(model, X0, s, {}), 1.9),
    ("dpm_2_ancestral eta=0", lambda s: K.sample_dpm_2_ancestral(model, X0, s, {}, ZERO, eta=0.0), 1.9),
    ("lms order 4", lambda s: K.sample_lms(model, X0, s, {}), 3.0),
    ("dpmpp_2m", lambda s: K.sample_dpmpp_2m(model, X0, s, {}), 1.9),
    ("dpmpp_2s_ancestral eta=0", lambda s: K.sample_dpmpp_2s_ancestral(model, X0, s, {}, ZERO, eta=0.0), 1.9),
    ("dpmpp_sde eta=0", lambda s: K.sample_dpmpp_sde(model, X0, s, {}, ZERO, eta=0.0), 1.9),
    ("dpmpp_2m_sde midpoint eta=0", lambda s: K.sample_dpmpp_2m_sde(model, X0, s, {}, ZERO, eta=0.0), 1.9),
    ("dpmpp_2m_sde heun eta=0", lambda s: K.sample_dpmpp_2m_sde(model, X0, s, {}, ZERO, eta=0.0, solver_type="heun"), 1.9),
    # (3M: the first step of the multistep start-up is first order, local error O(h^2), so the GLOBAL order is 2 — as published)
    ("dpmpp_3m_sde eta=0", lambda s: K.sample_dpmpp_3m_sde(model, X0, s, {}, ZERO, eta=0.0), 1.9),
]


def rel_err(out):
    ref = flow(X0, SMAX, SMIN)
    return float((out - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("name,call,min_order", DETERMINISTIC, ids=[d[0] for d in DETERMINISTIC])
def test_deterministic_sampler_restatements_converge_to_the_exact_gaussian_flow_at_their_published_order(name, call, min_order):
    errs = [rel_err(call(sigmas(n))) for n in (32, 64, 128)]
    orders = [math.log2(errs[i] / errs[i + 1]) for i in range(2)]
    assert errs[-1] < 1e-2, (name, errs)
    assert min(orders) >= min_order, (name, errs, orders)


def test_dpm_solver_fast_and_adaptive_converge_to_the_exact_gaussian_flow():
    e24 = rel_err(K.sample_dpm_fast(model, X0, SMIN, SMAX, 24, {}, ZERO))
    e96 = rel_err(K.sample_dpm_fast(model, X0, SMIN, SMAX, 96, {}, ZERO))
    assert e96 < 3e-4 and e96 < e24 / 8, (e24, e96)         # third-order steps: 4 x the evaluations buys > 8 x
    prev = None
    for rtol in (0.05, 0.005, 0.0005):
        out, info = K.sample_dpm_adaptive(model, X0, SMIN, SMAX, {}, ZERO, rtol=rtol, atol=rtol * 0.1, return_info=True)
        e = rel_err(out)
        assert e < 3 * rtol, (rtol, e, info)                   # rtol bounds the LOCAL error of a step: the global error stays within a small multiple
        assert prev is None or (e < prev and info["nfe"] > prev_nfe), (rtol, e, prev, info)
        prev, prev_nfe = e, info["nfe"]


N = 200_000
TARGET = S * S + SMIN * SMIN


def _x_start(gen):
    return torch.randn(N, 1, generator=gen, dtype=torch.float64) * math.sqrt(S * S + SMAX * SMAX)


STOCHASTIC = [
    # name, call(x, sigmas, noise_fn), |relative variance error| bound at n = 16 and at n = 64 (statistical floor sqrt(2 / N) = 0.3 %)
    ("euler_ancestral", lambda x, s, nf: K.sample_euler_ancestral(model, x, s, {}, nf), 0.30, 0.08),          # first order
    ("dpm_2_ancestral", lambda x, s, nf: K.sample_dpm_2_ancestral(model, x, s, {}, nf), 0.07, 0.015),
    ("dpmpp_2s_ancestral", lambda x, s, nf: K.sample_dpmpp_2s_ancestral(model, x, s, {}, nf), 0.06, 0.015),
    ("dpmpp_2m_sde", lambda x, s, nf: K.sample_dpmpp_2m_sde(model, x, s, {}, nf), 0.05, 0.015),
    ("dpmpp_2m_sde heun", lambda x, s, nf: K.sample_dpmpp_2m_sde(model, x, s, {}, nf, solver_type="heun"), 0.08, 0.015),
    ("dpmpp_3m_sde", lambda x, s, nf: K.sample_dpmpp_3m_sde(model, x, s, {}, nf), 0.05, 0.015),
    ("heun s_churn", lambda x, s, nf: K.sample_heun(model, x, s, {}, nf, s_churn=20.0), 0.10, 0.03),
]


@pytest.mark.parametrize("name,call,tol16,tol64", STOCHASTIC, ids=[d[0] for d in STOCHASTIC])
def test_stochastic_sampler_restatements_keep_the_gaussian_marginal(name, call, tol16, tol64):
    gen = torch.Generator().manual_seed(7)
    x = _x_start(gen)
    nf = lambda *a: torch.randn(N, 1, generator=gen, dtype=torch.float64)   # noqa: E731
    e16 = float(call(x, sigmas(16), nf).var()) / TARGET - 1
    e64 = float(call(x, sigmas(64), nf).var()) / TARGET - 1
    assert abs(e16) < tol16 and abs(e64) < tol64, (name, e16, e64)
    assert abs(e64) < abs(e16), (name, e16, e64)


def test_dpmpp_sde_is_second_order_in_the_marginal_only_on_one_brownian_path():
    """DPM++ SDE draws twice per step, over [sigma, sigma_mid'] and [sigma, sigma_next]: with the two draws taken from ONE path (the
    tree: nested intervals share their increments) the marginal variance converges at second order; with independent draws the same
    code falls back to the Euler-ancestral error.  Checks the sampler restatement and the tree's nesting consistency together."""
    gen = torch.Generator().manual_seed(7)
    x = _x_start(gen).view(1, N)
    errs = {}
    for n in (16, 64):
        tree = OracleTreeSampler(x, SMIN, SMAX, seed=[123])
        out = K.sample_dpmpp_sde(model, x, sigmas(n), {}, lambda a, b: tree(a, b).double())
        errs[n] = float(out.var()) / TARGET - 1
    assert abs(errs[16]) < 0.03 and abs(errs[64]) < 0.006, errs
    indep = lambda *a: torch.randn(1, N, generator=gen, dtype=torch.float64)   # noqa: E731
    e_ind = float(K.sample_dpmpp_sde(model, x, sigmas(64), {}, indep).var()) / TARGET - 1
    assert abs(e_ind) > 5 * abs(errs[64]) and abs(e_ind) > 0.03, (e_ind, errs)


def test_host_brownian_tree_equals_the_oracle_restatement_and_is_a_brownian_motion():
    host = importlib.import_module(f"{PKG}.brownian")
    shape = (3, 4, 32, 32)
    x = torch.zeros(shape)
    seeds = [11, 22, 33]
    h, o = host.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=seeds), OracleTreeSampler(x, 0.03, 14.6, seed=seeds)
    # same seeds, same bridge construction, same draws: equal up to the rounding of the final 1 / sqrt(dt) normalisation (one multiply
    # by sign / sqrt(dt) in brownian.py, a multiply and a divide in the oracle)
    close = lambda p, q: torch.allclose(p, q, rtol=2e-6, atol=1e-7)     # noqa: E731
    queries = [(14.6, 9.1), (9.1, 5.0), (14.6, 7.3), (5.0, 0.03), (2.0, 1.999999), (0.5, 0.7), (14.6, 0.03)]
    for a, b in queries:                                     # arbitrary order, reversed intervals, a sub-tolerance interval
        assert close(h(a, b), o(a, b)), (a, b)
    # a pure function of (seed, interval): independent of the batch an image sits in and of the queries made before
    h2 = host.BrownianTreeNoiseSampler(x[:1], 0.03, 14.6, seed=[22])
    assert torch.equal(h2(9.1, 5.0)[0], h(9.1, 5.0)[1])
    assert close(h(9.1, 5.0), o(9.1, 5.0))
    # a Brownian motion: normalised increments have unit variance, add up over adjacent intervals, do not correlate over disjoint ones
    big = torch.zeros(1, 1 << 16)
    t = host.BrownianTreeNoiseSampler(big, 0.03, 14.6, seed=[5])
    w = lambda a, b: t(a, b).double() * math.sqrt(abs(b - a))     # noqa: E731  (un-normalised increment, sign as queried)
    for a, b in ((14.6, 9.0), (9.0, 8.9), (1.0, 0.03)):
        v = float(t(a, b).double().var())
        assert abs(v - 1.0) < 0.03, (a, b, v)
    assert torch.allclose(w(14.6, 5.0), w(14.6, 9.0) + w(9.0, 5.0), atol=1e-5)
    c = float((t(14.6, 9.0).double() * t(9.0, 5.0).double()).mean())
    assert abs(c) < 0.02, c
    # nested intervals share their path: corr(W[a, c], W[a, b]) = sqrt((b - a) / (c - a))
    c_nested = float((t(14.6, 5.0).double() * t(14.6, 9.0).double()).mean())
    assert abs(c_nested - math.sqrt((14.6 - 9.0) / (14.6 - 5.0))) < 0.02, c_nested
