"""The product's CFG denoiser (sd_samplers.CFGDenoiser.forward) on the CPU, over the 22 scenarios the oracle's CFGDenoiser is pinned on
by executing the reference class (tests/golden/make_golden.py CFG_SCENARIOS -> tests/golden/cfg_denoiser.npz).

What runs on the host in the product is the whole case analysis of modules/sd_samplers_cfg_denoiser.py:156-311 — which rows the UNet
batch holds (AND composition, the edit model's third group), when uncond is skipped (NGMS, skip-early) and what stands in for it,
how cond / uncond of different token counts are padded or split into two calls, which context rows are cached, how the general combine
is reduced to the fused kernel's [cond | uncond] form, the masks, the CFG++ bookkeeping, the infotext keys, the x0 preview.  Here the
engine is a stub whose "UNet" is an analytic function of (scaled input, timestep, context rows, image / vector conditioning) and the
fused launches are replaced — for these tests only — by the elementwise contracts written in include/sdmi.h; the oracle side wraps the
SAME function in its CompVisDenoiser.  On the GPU the same scenarios run through the real engine (tests/test_gpu_models.py).
"""
import importlib
import importlib.util
import os
import types

import pytest
import torch

from oracle import kdiffusion as okd

PKG = "stable-diffusion-webui_amd"
C = 4


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


def golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def unet(x_scaled, t, ctx, extra=None):
    """eps = U(x * c_in, t, context rows[, c_concat rows | c_adm rows]) — depends on every input and on the context's token COUNT."""
    c = ctx.sum(dim=(1, 2))[:, None, None, None]
    out = torch.tanh(0.5 * x_scaled + 0.2 * c + 0.002 * t.float()[:, None, None, None])
    if extra is None:
        return out
    if extra.dim() == 2:                                     # unCLIP: a vector per row
        return out + 0.05 * extra.sum(1)[:, None, None, None] * x_scaled
    return out + 0.05 * extra[:, :C] * x_scaled              # inpainting / edit: c_concat concatenated along the channels


class StubEngine:
    """What CFGDenoiser.forward asks of sd_model.engine."""

    def __init__(self, in_channels):
        self.unet_cfg = types.SimpleNamespace(in_channels=in_channels)
        self.weights_version = 0
        self.ctx = None
        self.calls = []

    def set_context(self, ctx):
        self.ctx = ctx.clone()

    def unet_forward(self, x, timesteps, context=None, y=None, out=None, uniform_t=False, cfg_pairs=False):
        ctx = self.ctx if context is None else context
        assert ctx.shape[0] == x.shape[0], "context rows do not match the UNet batch"
        assert not uniform_t or bool((timesteps == timesteps[0]).all())
        if cfg_pairs:                                        # the promise behind the shared CFG prefix: rows [B, 2B) repeat rows [0, B)
            h = x.shape[0] // 2
            assert torch.equal(x[:h], x[h:]) and torch.equal(timesteps[:h], timesteps[h:])
        extra = y if y is not None else (x[:, C:] if x.shape[1] > C else None)
        self.calls.append((x.shape[0], ctx.shape[1], bool(cfg_pairs)))
        out.copy_(unet(x[:, :C], timesteps, ctx, extra))
        return out


class TorchCfgKernels:
    """include/sdmi.h: sdmi_cfg_prepare_input / _concat / sdmi_cfg_combine / _affine, on CPU tensors (`ptr` is the identity here)."""

    @staticmethod
    def sdmi_cfg_prepare_input(x, c_in, dst, dtype, nb, reps, chw, stream):
        xs = x.reshape(nb, -1) if x.dim() < 4 or x.shape[0] != nb else x[:nb].reshape(nb, -1)
        for r in range(reps):
            for b in range(nb):
                dst[r * nb + b].copy_((xs[b] * (1.0 if c_in is None else c_in[b])).reshape(dst[0].shape))
        return 0

    @staticmethod
    def sdmi_cfg_prepare_concat(x, c_in, ic, dst, dtype, nb, reps, c, cc, hw, zero_reps, stream):
        xs = x.reshape(nb, c, -1)
        ics = ic.reshape(nb, cc, -1)
        for r in range(reps):
            for b in range(nb):
                row = dst[r * nb + b].reshape(c + cc, -1)
                row[:c].copy_(xs[b] * (1.0 if c_in is None else c_in[b]))
                row[c:].copy_(torch.zeros_like(ics[b]) if (zero_reps >> r) & 1 else ics[b])
        return 0

    @staticmethod
    def _blend(out, mask, nmask, init):
        return out if mask is None else out * nmask + init * mask

    @classmethod
    def sdmi_cfg_combine(cls, x, eps, c_out, scale, mode, mask, nmask, init, den, B, chw, stream):
        ec, eu = eps[:B], eps[B:2 * B]
        if mode == 0:
            co = c_out.view(B, 1, 1, 1)
            cd, ud = x + ec * co, x + eu * co
        else:
            cd, ud = ec, eu
        den.copy_(cls._blend(ud + (cd - ud) * scale, mask, nmask, init))
        return 0

    @classmethod
    def sdmi_cfg_combine_affine(cls, x, out, c_out, c_skip, scale, mask, nmask, init, den, B, chw, stream):
        co, cs = c_out.view(B, 1, 1, 1), c_skip.view(B, 1, 1, 1)
        cd, ud = out[:B] * co + x * cs, out[B:2 * B] * co + x * cs
        den.copy_(cls._blend(ud + (cd - ud) * scale, mask, nmask, init))
        return 0


@pytest.fixture()
def ss(monkeypatch):
    mod = sub("sd_samplers")
    monkeypatch.setattr(mod, "lib", TorchCfgKernels())
    monkeypatch.setattr(mod, "ptr", lambda t: t)
    monkeypatch.setattr(mod, "stream_ptr", lambda: None)
    monkeypatch.setattr(mod, "_lc", lambda out, terms, coefs: out.copy_(sum(float(c) * t for c, t in zip(coefs, terms))))
    monkeypatch.setattr(mod.ops, "mask_blend", lambda x, init, mask, nmask: x.copy_(x * nmask + init * mask))
    return mod


OPT_NAMES = ("skip_early_cond", "s_min_uncond_all", "pad_cond_uncond", "pad_cond_uncond_v0", "batch_cond_uncond", "live_preview_content")


def run_product(ss, monkeypatch, sc, inp, parameterization="eps"):
    base = dict(skip_early_cond=0.0, s_min_uncond_all=False, pad_cond_uncond=False, pad_cond_uncond_v0=False, batch_cond_uncond=True,
                live_preview_content="Prompt")
    base.update(sc.get("opts", {}))
    for k, v in base.items():
        monkeypatch.setattr(ss.shared.opts, k, v, raising=False)
    adm, edit = bool(sc.get("adm")), bool(sc.get("edit"))
    eng = StubEngine(C if adm else C + inp["image_cond"].shape[1])
    sd_model = types.SimpleNamespace(engine=eng, alphas_cumprod=okd.make_alphas_cumprod(), parameterization=parameterization,
                                     cond_stage_key="edit" if edit else "txt",
                                     model=types.SimpleNamespace(conditioning_key="crossattn-adm" if adm else "hybrid"),
                                     cond_stage_model_empty_prompt=inp["empty"], device=torch.device("cpu"))
    sampler = types.SimpleNamespace(sd_model=sd_model, last_latent=None, sampler_extra_args={})
    d = ss.CFGDenoiser(sampler, mode=0)
    d.p = types.SimpleNamespace(extra_generation_params={}, scripts=None)
    d.step, d.total_steps = sc.get("step", 0), sc.get("total_steps", 20)
    d.image_cfg_scale = sc.get("image_cfg_scale")
    d.cond_scale_miltiplier = sc.get("cond_scale_miltiplier", 1.0)
    d.need_last_noise_uncond = sc.get("need_last_noise_uncond", False)
    d.mask_before_denoising = sc.get("mask_before", False)
    if sc.get("mask") or edit:
        d.init_latent = inp["init_latent"]
    if sc.get("mask"):
        d.mask, d.nmask = inp["mask"].expand_as(inp["x"]).contiguous(), (1 - inp["mask"]).expand_as(inp["x"]).contiguous()
    # (the context cache is keyed on the identity of the caller's cond / uncond tensors, which a sampler loop keeps alive across its
    # steps: keep these alive as long as the denoiser, or a later allocation could take their address)
    d._keepalive = (inp["uncond"].clone(), inp["cond"].clone())
    out = d.forward(inp["x"].clone(), inp["sigma"], d._keepalive[0], (inp["conds_list"], d._keepalive[1]), 7.0,
                    sc.get("s_min_uncond", 0.0), inp["image_cond"])
    return d, out, eng


def run_oracle(sc, inp, wrapper=okd.CompVisDenoiser):
    den = wrapper(lambda xs, t, cond, ic=None: unet(xs, t, cond, ic), okd.make_alphas_cumprod())
    d = okd.CFGDenoiser(den)
    for key, val in sc.get("opts", {}).items():
        if key != "batch_cond_uncond":
            setattr(d, key, val)
    d.empty_prompt, d.step, d.total_steps = inp["empty"], sc.get("step", 0), sc.get("total_steps", 20)
    d.image_cfg_scale, d.is_edit_cond_stage = sc.get("image_cfg_scale"), bool(sc.get("edit"))
    d.adm = bool(sc.get("adm"))
    d.cond_scale_miltiplier = sc.get("cond_scale_miltiplier", 1.0)
    d.need_last_noise_uncond = sc.get("need_last_noise_uncond", False)
    d.mask_before_denoising = sc.get("mask_before", False)
    if sc.get("mask") or sc.get("edit"):
        d.init_latent = inp["init_latent"]
    if sc.get("mask"):
        d.mask, d.nmask = inp["mask"], 1 - inp["mask"]
    out = d(inp["x"].clone(), inp["sigma"], inp["uncond"], (inp["conds_list"], inp["cond"]), 7.0, sc.get("s_min_uncond", 0.0), inp["image_cond"])
    return d, out


MG = golden()


@pytest.mark.parametrize("k", range(len(MG.CFG_SCENARIOS)), ids=[n for n, _ in MG.CFG_SCENARIOS])
def test_host_cfg_denoiser_matches_the_pinned_oracle_class(ss, monkeypatch, k):
    name, sc = MG.CFG_SCENARIOS[k]
    inp = MG.cfg_scenario_inputs(k, sc)
    od, want = run_oracle(sc, inp)
    pd, got, eng = run_product(ss, monkeypatch, sc, inp)
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    assert rel(got, want) < 2e-6, (name, rel(got, want))
    # what an interrupted job returns / the "Prompt" live preview: the x0 prediction of each image's first prompt (:295-304)
    assert rel(pd.sampler.last_latent, od.last_latent) < 2e-6, name
    assert torch.equal(ss.shared.state.current_latent, pd.sampler.last_latent)
    assert (pd.padded_cond_uncond, pd.padded_cond_uncond_v0, pd.step) == (od.padded_cond_uncond, od.padded_cond_uncond_v0, od.step), name
    # infotext keys of :222-227
    info = pd.p.extra_generation_params
    skipped = od.skipped_uncond
    early = sc.get("opts", {}).get("skip_early_cond", 0.0)
    assert ("Skip Early CFG" in info) == bool(skipped and early), (name, info)
    assert ("NGMS" in info) == bool(skipped and not early), (name, info)
    assert ("NGMS all steps" in info) == bool(skipped and not early and sc.get("opts", {}).get("s_min_uncond_all")), (name, info)
    if od.need_last_noise_uncond:
        # the reference keeps the wrapped model's uncond rows (denoised in sigma space); the product keeps the UNet's (eps) and the sampler
        # converts: compare through the wrapper's affine map
        c_out = -inp["sigma"].view(-1, 1, 1, 1)
        assert rel(inp["x"] + pd.last_noise_uncond * c_out, od.last_noise_uncond) < 2e-6, name
    # the UNet batch: one call unless cond and uncond differ in token count (then one per length), the shared-prefix promise only for
    # the plain [cond | uncond] batch
    lengths_differ = inp["cond"].shape[1] != inp["uncond"].shape[1] and not (pd.padded_cond_uncond or pd.padded_cond_uncond_v0)
    assert len(eng.calls) == (2 if lengths_differ and not skipped else 1), (name, eng.calls)
    plain = sc.get("conds_list") is None and not skipped and not sc.get("edit") and not lengths_differ
    assert eng.calls[0][2] == plain, (name, eng.calls)


def test_host_cfg_denoiser_v_prediction_uses_the_affine_combine(ss, monkeypatch):
    """SD 2.x 768-v: denoised = v * c_out + x * c_skip per half, then CFG (sdmi_cfg_combine_affine)."""
    for k in (0, 1, 8):                                      # plain, AND, mask after
        name, sc = MG.CFG_SCENARIOS[k]
        inp = MG.cfg_scenario_inputs(k, sc)
        od, want = run_oracle(sc, inp, okd.CompVisVDenoiser)
        pd, got, _ = run_product(ss, monkeypatch, sc, inp, parameterization="v")
        assert float((got - want).norm() / want.norm()) < 2e-6, name
        assert float((pd.sampler.last_latent - od.last_latent).norm() / od.last_latent.norm()) < 2e-6, name


def test_host_cfg_denoiser_context_cache_follows_the_selection(ss, monkeypatch):
    """The cross-attention K / V cache is keyed on the tensors the context rows were selected from: the same tensors on the next step
    keep it (no second set_context), a different selection of the same shape replaces it."""
    name, sc = MG.CFG_SCENARIOS[0]
    inp = MG.cfg_scenario_inputs(0, sc)
    pd, _, eng = run_product(ss, monkeypatch, sc, inp)
    n_set = []
    orig = eng.set_context
    eng.set_context = lambda ctx: (n_set.append(ctx.shape), orig(ctx))
    cond, uncond = inp["cond"].clone(), inp["uncond"].clone()
    args = (inp["sigma"], uncond, (inp["conds_list"], cond), 7.0, 0.0, inp["image_cond"])
    pd.forward(inp["x"].clone(), *args)
    pd.forward(inp["x"].clone(), *args)
    assert len(n_set) == 1                                    # first call with these tensors sets it, the second reuses it
    other = cond + 1.0
    out_a = pd.forward(inp["x"].clone(), inp["sigma"], uncond, (inp["conds_list"], other), 7.0, 0.0, inp["image_cond"])
    assert len(n_set) == 2
    od, want = run_oracle(sc, dict(inp, cond=other))
    assert float((out_a - want).norm() / want.norm()) < 2e-6


def test_host_cfg_denoiser_obeys_interrupt(ss, monkeypatch):
    name, sc = MG.CFG_SCENARIOS[0]
    inp = MG.cfg_scenario_inputs(0, sc)
    pd, _, _ = run_product(ss, monkeypatch, sc, inp)
    monkeypatch.setattr(ss.shared.state, "interrupted", True, raising=False)
    with pytest.raises(ss.InterruptedException):
        pd.forward(inp["x"].clone(), inp["sigma"], inp["uncond"], (inp["conds_list"], inp["cond"]), 7.0, 0.0, inp["image_cond"])
