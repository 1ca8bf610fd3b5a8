"""CPU ORACLE (test infrastructure, never the product path): UniPC multistep sampler as the reference runs it.

Restates, in fp32 torch on the CPU,
  * modules/models/diffusion/uni_pc/uni_pc.py:6-175   NoiseScheduleVP('discrete') — log-alpha table, piecewise-linear
                                                       interpolation (interpolate_fn, :811-850), lambda and its inverse
  * modules/models/diffusion/uni_pc/uni_pc.py:435-448 data prediction  x0 = (x - sigma_t * eps) / alpha_t
  * modules/models/diffusion/uni_pc/uni_pc.py:459-474 time_uniform / time_quadratic / logSNR step placement
  * modules/models/diffusion/uni_pc/uni_pc.py:625-743 the B(h) predictor / corrector update (predict_x0 branch)
  * modules/models/diffusion/uni_pc/uni_pc.py:746-805 the multistep driver (warm-up orders, lower_order_final, no corrector
                                                       on the last step)
  * modules/sd_samplers_timesteps_impl.py:144-179     UniPCCFG / unipc(): model time (t - 1/N) * 1000, callback per update,
                                                       img2img start t = timesteps[-1]/1000 + 1/1000

Pinned by tests/golden/unipc.npz, which tests/golden/make_golden.py::gen_unipc produces by executing those reference files.
The 'vary_coeff' variant (uni_pc.py:522-623) is restated by _vary_update, including the reference's use of row K-2 of the inverted
coefficient matrix for the corrector's newest-difference term (its loop variable after the loop).
"""
import torch


def _interp(x, xp, yp):
    """uni_pc.py:811-850 for one channel: linear between the two keypoints around x, and the outermost segment extended
    beyond either end.  x: [N], xp/yp: [K] with xp ascending."""
    k = xp.shape[0]
    pos = torch.searchsorted(xp, x.contiguous())            # number of keypoints strictly below x
    seg = torch.where(pos == 0, torch.zeros_like(pos), torch.where(pos == k, torch.full_like(pos, k - 2), pos - 1))
    x0, x1, y0, y1 = xp[seg], xp[seg + 1], yp[seg], yp[seg + 1]
    return y0 + (x - x0) * (y1 - y0) / (x1 - x0)


class DiscreteVPSchedule:
    """NoiseScheduleVP('discrete', alphas_cumprod=...) — uni_pc.py:96-108, 125-175."""

    def __init__(self, alphas_cumprod):
        self.log_alpha = 0.5 * torch.log(alphas_cumprod.float().cpu())
        self.total_N = self.log_alpha.shape[0]
        self.T = 1.0
        self.t_grid = torch.linspace(0., 1., self.total_N + 1)[1:]

    def log_mean_coeff(self, t):
        return _interp(t.reshape(-1), self.t_grid, self.log_alpha)

    def alpha(self, t):
        return torch.exp(self.log_mean_coeff(t))

    def std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.log_mean_coeff(t)))

    def lam(self, t):
        lmc = self.log_mean_coeff(t)
        return lmc - 0.5 * torch.log(1. - torch.exp(2. * lmc))

    def inverse_lam(self, lamb):
        la = -0.5 * torch.logaddexp(torch.zeros((1,)), -2. * lamb)
        return _interp(la.reshape(-1), torch.flip(self.log_alpha, [0]), torch.flip(self.t_grid, [0]))


def time_steps(ns, skip_type, t_T, t_0, n):
    """uni_pc.py:459-474"""
    if skip_type == 'logSNR':
        l_T, l_0 = ns.lam(torch.tensor(t_T)), ns.lam(torch.tensor(t_0))
        return ns.inverse_lam(torch.linspace(l_T.item(), l_0.item(), n + 1))
    if skip_type == 'time_uniform':
        return torch.linspace(t_T, t_0, n + 1)
    if skip_type == 'time_quadratic':
        return torch.linspace(t_T ** 0.5, t_0 ** 0.5, n + 1).pow(2)
    raise ValueError(skip_type)


def _bh_update(ns, model_fn, x, m_list, t_list, t, order, variant, use_corrector):
    """uni_pc.py:625-743, predict_x0 branch.  m_list/t_list: histories, newest last.  Returns (x_t, model_t or None)."""
    assert order <= len(m_list)
    t_p0, m0 = t_list[-1], m_list[-1]
    lam_p0, lam_t = ns.lam(t_p0), ns.lam(t)
    sig_p0, sig_t = ns.std(t_p0), ns.std(t)
    alpha_t = torch.exp(ns.log_mean_coeff(t))
    h = lam_t - lam_p0
    rks, d1s = [], []
    for i in range(1, order):
        rk = ((ns.lam(t_list[-(i + 1)]) - lam_p0) / h)[0]
        rks.append(rk)
        d1s.append((m_list[-(i + 1)] - m0) / rk)
    rks.append(1.)
    rks = torch.tensor(rks)
    hh = -h[0]
    h_phi_1 = torch.expm1(hh)
    h_phi_k = h_phi_1 / hh - 1
    fact = 1
    if variant == 'bh1':
        b_h = hh
    elif variant == 'bh2':
        b_h = torch.expm1(hh)
    else:
        raise NotImplementedError(variant)
    rows, b = [], []
    for i in range(1, order + 1):
        rows.append(torch.pow(rks, i - 1))
        b.append(h_phi_k * fact / b_h)
        fact *= (i + 1)
        h_phi_k = h_phi_k / hh - 1 / fact
    r_mat = torch.stack(rows)
    b = torch.tensor(b)

    e4 = lambda v: v.reshape(-1, 1, 1, 1)
    rhos_p = None
    if d1s:
        d1s = torch.stack(d1s, dim=1)                                       # [B, K, C, H, W]
        rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(r_mat[:-1, :-1], b[:-1])
    else:
        d1s = None
    if use_corrector:
        rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(r_mat, b)

    x_base = e4(sig_t / sig_p0) * x - e4(alpha_t * h_phi_1) * m0
    pred = torch.einsum('k,bkchw->bchw', rhos_p, d1s) if d1s is not None else 0
    x_t = x_base - e4(alpha_t * b_h) * pred
    model_t = None
    if use_corrector:
        model_t = model_fn(x_t, t)
        corr = torch.einsum('k,bkchw->bchw', rhos_c[:-1], d1s) if d1s is not None else 0
        x_t = x_base - e4(alpha_t * b_h) * (corr + rhos_c[-1] * (model_t - m0))
    return x_t, model_t


def _vary_update(ns, model_fn, x, m_list, t_list, t, order, use_corrector):
    """uni_pc.py:522-623 (multistep_uni_pc_vary_update), predict_x0 branch."""
    assert order <= len(m_list)
    t_p0, m0 = t_list[-1], m_list[-1]
    lam_p0, lam_t = ns.lam(t_p0), ns.lam(t)
    sig_p0, sig_t = ns.std(t_p0), ns.std(t)
    alpha_t = torch.exp(ns.log_mean_coeff(t))
    h = lam_t - lam_p0
    rks, d1s = [], []
    for i in range(1, order):
        rk = ((ns.lam(t_list[-(i + 1)]) - lam_p0) / h)[0]
        rks.append(rk)
        d1s.append((m_list[-(i + 1)] - m0) / rk)
    rks.append(1.)
    rks = torch.tensor(rks)
    K = len(rks)
    cols, col = [], torch.ones_like(rks)
    for k in range(1, K + 1):                                            # C[i][k-1] = rks_i^(k-1) / k!
        cols.append(col)
        col = col * rks / (k + 1)
    C = torch.stack(cols, dim=1)
    a_p = torch.linalg.inv(C[:-1, :-1]) if d1s else None
    a_c = torch.linalg.inv(C) if use_corrector else None
    hh = -h[0]
    h_phi_1 = torch.expm1(hh)
    h_phi_ks, fact, h_phi_k = [], 1, h_phi_1
    for k in range(1, K + 2):
        h_phi_ks.append(h_phi_k)
        h_phi_k = h_phi_k / hh - 1 / fact
        fact *= (k + 1)
    e4 = lambda v: v.reshape(-1, 1, 1, 1)
    d1s = torch.stack(d1s, dim=1) if d1s else None                       # [B, K-1, C, H, W]
    x_base = e4(sig_t / sig_p0) * x - e4(alpha_t * h_phi_1) * m0
    x_t = x_base
    if d1s is not None:
        for k in range(K - 1):
            x_t = x_t - e4(alpha_t * h_phi_ks[k + 1]) * torch.einsum('bkchw,k->bchw', d1s, a_p[k])
    model_t = None
    if use_corrector:
        model_t = model_fn(x_t, t)
        d1_t = model_t - m0
        x_t = x_base
        k = 0
        for k in range(K - 1):
            x_t = x_t - e4(alpha_t * h_phi_ks[k + 1]) * torch.einsum('bkchw,k->bchw', d1s, a_c[k][:-1])
        x_t = x_t - e4(alpha_t * h_phi_ks[K]) * (d1_t * a_c[k][-1])     # (k = K-2 after the loop, 0 without one: as the reference)
    return x_t, model_t


def _update(ns, model_fn, x, m_list, t_list, t, order, variant, use_corrector):
    if variant == 'vary_coeff':
        return _vary_update(ns, model_fn, x, m_list, t_list, t, order, use_corrector)
    return _bh_update(ns, model_fn, x, m_list, t_list, t, order, variant, use_corrector)


def sample_unipc(model, x, timesteps, alphas_cumprod, extra_args, callback=None, is_img2img=False, variant='bh1',
                 skip_type='time_uniform', order=3, lower_order_final=True):
    """unipc() of modules/sd_samplers_timesteps_impl.py:170-179 + UniPC.sample(method='multistep') of uni_pc.py:746-805.
    ``model(x, t_model * s_in, **extra_args)`` returns eps (CFGDenoiserTimesteps)."""
    ns = DiscreteVPSchedule(alphas_cumprod)
    steps = len(timesteps)
    t_T = float(timesteps[-1] / 1000 + 1 / 1000) if is_img2img else ns.T
    t_0 = 1. / ns.total_N
    assert steps >= order, "UniPC order must be < sampling steps"
    ts = time_steps(ns, skip_type, t_T, t_0, steps)
    bsz = x.shape[0]
    state = {'i': 0}

    def model_fn(xx, t):                                                 # data prediction, uni_pc.py:435-448
        eps = model(xx, (t - 1. / ns.total_N) * 1000., **extra_args)
        return (xx - ns.std(t).reshape(-1, 1, 1, 1) * eps) / ns.alpha(t).reshape(-1, 1, 1, 1)

    def after_update(xx, model_x):
        if callback is not None:
            callback({'x': xx, 'i': state['i'], 'sigma': 0, 'sigma_hat': 0, 'denoised': model_x})
        state['i'] += 1

    vec_t = ts[0].expand(bsz)
    m_list, t_list = [model_fn(x, vec_t)], [vec_t]
    for init_order in range(1, order):                                   # warm-up with increasing order
        vec_t = ts[init_order].expand(bsz)
        x, model_x = _update(ns, model_fn, x, m_list, t_list, vec_t, init_order, variant, True)
        after_update(x, model_x)
        m_list.append(model_x)
        t_list.append(vec_t)
    for step in range(order, steps + 1):
        vec_t = ts[step].expand(bsz)
        step_order = min(order, steps + 1 - step) if lower_order_final else order
        x, model_x = _update(ns, model_fn, x, m_list, t_list, vec_t, step_order, variant, step != steps)
        after_update(x, model_x)
        for i in range(order - 1):
            t_list[i], m_list[i] = t_list[i + 1], m_list[i + 1]
        t_list[-1] = vec_t
        if step < steps:
            m_list[-1] = model_x
    return x
