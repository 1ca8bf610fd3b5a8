"""Pin the CPU oracle to every vector the reference offers for this path.

Fixtures in tests/golden/*.npz were produced by executing the reference's own files
(tests/golden/make_golden.py); nothing here reads /root/reference.
"""
import math
import os

import numpy as np
import pytest
import torch

from helpers import rel_l2, seeded, seeded_module_weights
from oracle import kdiffusion as kd
from oracle import rng as orng
from oracle import unet as ounet
from oracle import vae as ovae


def test_philox_docstring_vector():
    """modules/rng_philox.py:12-14"""
    want = np.array([[-0.92466259, -0.42534415, -2.6438457, 0.14518388],
                     [-0.12086647, -0.57972564, -0.62285122, -0.32838709],
                     [-1.07454231, -0.36314407, -1.67105067, 2.26550497]], dtype=np.float32)
    got = orng.Generator(0).randn((3, 4))
    assert got.dtype == np.float32
    # The docstring prints CUDA's fp32 result; the reference's own numpy code (float64 Box-Muller rounded once)
    # lands within 1 ulp of it (measured 2.4e-7 here).  Bit-exactness vs the reference code is the next test.
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-7)


def test_philox_matches_reference_bitwise(golden_dir):
    z = np.load(os.path.join(golden_dir, "philox.npz"))
    gens = {}
    for key in sorted(k for k in z.files if k.startswith("c")):
        ci, seed, draw = key.split("_")
        seed, draw = int(seed[4:]), int(draw[4:])
        g = gens.setdefault((ci, seed), orng.Generator(seed))
        assert g.offset == draw
        got = g.randn(z[key].shape)
        np.testing.assert_array_equal(got, z[key], err_msg=key)


def test_image_rng_stream_semantics():
    """ImageRNG: image i uses Generator(seed_i); first() then next() advance offset by one (modules/rng.py:99-163)."""
    r = orng.ImageRNG((4, 8, 8), [1000, 1001, 1002])
    a, b = r.next(), r.next()
    g = orng.Generator(1001)
    np.testing.assert_array_equal(a[1].numpy(), g.randn((4, 8, 8)))
    np.testing.assert_array_equal(b[1].numpy(), g.randn((4, 8, 8)))
    # batch invariance: image 1 of a batch == that image generated alone
    solo = orng.ImageRNG((4, 8, 8), [1001])
    np.testing.assert_array_equal(solo.next()[0].numpy(), a[1].numpy())


def test_attention_matches_sub_quadratic_reference(golden_dir):
    """oracle CrossAttention core == modules/sub_quadratic_attention.py:141-215 outputs."""
    z = np.load(os.path.join(golden_dir, "subquad_attention.npz"))
    for ci in range(3):
        bh, n, mk, d = z[f"c{ci}_shape"]
        q, k, v = seeded((bh, n, d), 10 + ci), seeded((bh, mk, d), 20 + ci), seeded((bh, mk, d), 30 + ci)
        sim = torch.einsum('bid,bjd->bij', q, k) * (d ** -0.5)
        out = torch.einsum('bij,bjd->bid', sim.softmax(dim=-1), v)
        assert rel_l2(out, z[f"c{ci}_out"]) < 2e-6


def _copy_into(oracle_module, values_from_seed):
    seeded_module_weights(oracle_module, values_from_seed)


def test_vae_decoder_matches_reference_full_size(golden_dir):
    """oracle Decoder == modules/models/sd3/sd3_impls.py:305-355 VAEDecoder(z_channels=4), SD1.5 size."""
    z = np.load(os.path.join(golden_dir, "vae_decoder.npz"))
    assert int(z["full_nparams"]) == 49_490_179
    dec = ovae.Decoder(ovae.sd15_vae_config())
    assert sum(p.numel() for p in dec.parameters()) == 49_490_179
    seeded_module_weights(dec, 777)       # same state_dict order as the reference class => same values
    with torch.no_grad():
        out = dec(seeded((1, 4, 8, 8), 778))
    assert rel_l2(out, z["full_out"]) < 1e-5


def test_vae_decoder_matches_reference_at_bench_shape_512(golden_dir):
    """oracle Decoder == the reference's VAEDecoder on a 64x64 latent (the 512x512 decode bench.py times); every 4th pixel kept."""
    z = np.load(os.path.join(golden_dir, "vae_decoder_512.npz"))
    dec = ovae.Decoder(ovae.sd15_vae_config())
    seeded_module_weights(dec, 777)
    with torch.no_grad():
        out = dec(seeded((1, 4, 64, 64), 778))
    assert rel_l2(out[:, :, ::4, ::4], z["full_out_512_sub4"]) < 1e-5
    assert abs(float(out.mean()) - float(z["full_out_512_mean"])) < 1e-5 and abs(float(out.std()) - float(z["full_out_512_std"])) < 1e-5


def test_vae_decoder_and_encoder_match_reference_small(golden_dir):
    z = np.load(os.path.join(golden_dir, "vae_decoder.npz"))
    cfg = ovae.tiny_vae_config()
    dec = ovae.Decoder(cfg)
    seeded_module_weights(dec, 779)
    with torch.no_grad():
        assert rel_l2(dec(seeded((2, 4, 16, 16), 780)), z["small_out"]) < 1e-5
    ze = np.load(os.path.join(golden_dir, "vae_encoder.npz"))
    enc = ovae.Encoder(cfg)
    seeded_module_weights(enc, 781)
    with torch.no_grad():
        assert rel_l2(enc(seeded((2, 3, 32, 32), 782)), ze["small_out"]) < 1e-5


def test_ddim_matches_reference(golden_dir):
    """oracle sample_ddim == modules/sd_samplers_timesteps_impl.py:12-40 on an analytic eps model."""
    z = np.load(os.path.join(golden_dir, "ddim.npz"))
    ac = kd.make_alphas_cumprod()

    def model(x, t, **kw):
        return torch.tanh(0.7 * x + (t / 1000.0)[:, None, None, None]) * 0.9 + 0.05 * x

    for ci in range(2):
        steps, eta = z[f"c{ci}_steps_eta"]
        steps = int(steps)
        draws = iter([seeded((2, 4, 8, 8), 900 + i) for i in range(steps + 2)])
        out = kd.sample_ddim(model, seeded((2, 4, 8, 8), 890 + ci), kd.ddim_timesteps(steps), ac, {},
                             lambda: next(draws), eta=float(eta))
        np.testing.assert_allclose(out.numpy(), z[f"c{ci}_out"], rtol=0, atol=1e-6)


def test_ddim_cfgpp_matches_reference(golden_dir):
    """oracle sample_ddim_cfgpp == modules/sd_samplers_timesteps_impl.py:43-82 (uncond-eps direction term, multiplier 1/12.5)."""
    z = np.load(os.path.join(golden_dir, "ddim.npz"))
    ac = kd.make_alphas_cumprod()

    class Model:
        def __call__(self, x, t, **kw):
            self.last_noise_uncond = torch.sin(0.3 * x) * 0.8 - 0.1 * (t / 1000.0)[:, None, None, None]
            return torch.tanh(0.7 * x + (t / 1000.0)[:, None, None, None]) * 0.9 + 0.05 * x

    for ci in range(2):
        steps, eta = z[f"cfgpp{ci}_steps_eta"]
        steps = int(steps)
        draws = iter([seeded((2, 4, 8, 8), 950 + i) for i in range(steps + 2)])
        m = Model()
        out = kd.sample_ddim_cfgpp(m, seeded((2, 4, 8, 8), 940 + ci), kd.ddim_timesteps(steps), ac, {}, lambda: next(draws), eta=float(eta))
        assert m.cond_scale_miltiplier == 1 / 12.5
        np.testing.assert_allclose(out.numpy(), z[f"cfgpp{ci}_out"], rtol=0, atol=1e-6)


def test_restart_sampler_matches_reference(golden_dir):
    """oracle restart_sampler == modules/sd_samplers_extra.py:6-74 for 8 steps (no restart), 22 (one restart of 9 steps) and
    40 (two restarts of 10)."""
    z = np.load(os.path.join(golden_dir, "restart.npz"))
    den = kd.CompVisDenoiser(None, kd.make_alphas_cumprod())
    smin, smax = den.sigmas[0].item(), den.sigmas[-1].item()

    def model(x, sigma, **kw):
        s = sigma[:, None, None, None]
        return x / (1 + s * s) + torch.tanh(0.5 * x) * (s * s / (1 + s * s)) * 0.3

    for ci in range(3):
        steps = int(z[f"c{ci}_steps"][0])
        draws = iter([seeded((2, 4, 8, 8), 3000 + 10 * ci + i) for i in range(8)])
        sigmas = kd.get_sigmas_karras(steps, smin, smax)
        out = kd.restart_sampler(model, seeded((2, 4, 8, 8), 2990 + ci) * sigmas[0], sigmas, {}, lambda: next(draws))
        np.testing.assert_allclose(out.numpy(), z[f"c{ci}_out"], rtol=0, atol=2e-6)
    assert len(kd.restart_step_list(kd.get_sigmas_karras(22, smin, smax))) == 13 + 9


def test_plms_matches_reference(golden_dir):
    """oracle sample_plms == modules/sd_samplers_timesteps_impl.py:85-137 on an analytic eps model."""
    z = np.load(os.path.join(golden_dir, "plms.npz"))
    ac = kd.make_alphas_cumprod()

    def model(x, t, **kw):
        return torch.tanh(0.7 * x + (t / 1000.0)[:, None, None, None]) * 0.9 + 0.05 * x

    for ci in range(2):
        steps = int(z[f"c{ci}_steps"][0])
        out = kd.sample_plms(model, seeded((2, 4, 8, 8), 870 + ci), kd.ddim_timesteps(steps), ac, {})
        np.testing.assert_allclose(out.numpy(), z[f"c{ci}_out"], rtol=0, atol=1e-6)


def _unipc_cases(z):
    for ci in range(sum(1 for k in z.files if k.endswith("_cfg"))):
        steps, order, lof, t_enc = (int(v) for v in z[f"c{ci}_cfg"])
        variant, skip = (str(v) for v in z[f"c{ci}_variant_skip"])
        ts = kd.ddim_timesteps(steps)
        yield ci, (ts[:t_enc] if t_enc else ts), dict(is_img2img=bool(t_enc), variant=variant, skip_type=skip, order=order,
                                                       lower_order_final=bool(lof))


def test_unipc_matches_reference(golden_dir):
    """oracle/unipc.py == unipc() (modules/sd_samplers_timesteps_impl.py:144-179) over the real uni_pc.py: final latents, the
    model times of every evaluation, the callback count and the last data prediction, for the three skip types, bh1 / bh2 / vary_coeff,
    orders 1-4, lower_order_final off and an img2img start."""
    from oracle import unipc
    z = np.load(os.path.join(golden_dir, "unipc.npz"))
    ac = kd.make_alphas_cumprod()
    for ci, ts, kw in _unipc_cases(z):
        times, dens = [], []

        def model(x, t, **_):
            times.append(float(t[0]))
            return torch.tanh(0.7 * x + (t / 1000.0)[:, None, None, None]) * 0.9 + 0.05 * x

        batch = int(z[f"c{ci}_batch"][0])
        out = unipc.sample_unipc(model, seeded((batch, 4, 8, 8), 990 + ci), ts, ac, {}, callback=lambda d: dens.append(d['denoised']), **kw)
        np.testing.assert_allclose(out.numpy(), z[f"c{ci}_out"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.array(times), z[f"c{ci}_model_t"], rtol=0, atol=1e-4)
        assert len(dens) == int(z[f"c{ci}_n_callbacks"][0]) and dens[-1] is None
        np.testing.assert_allclose(dens[-2].numpy(), z[f"c{ci}_last_denoised"], rtol=0, atol=1e-6)


def test_lcm_matches_reference(golden_dir):
    """oracle LCMCompVisDenoiser / sample_lcm == modules/sd_samplers_lcm.py executed over the oracle's k-diffusion base classes:
    the 50-entry sigma table, get_sigmas (all / n steps), sigma_to_t, t_to_sigma, the scaled forward and 4- and 8-step runs."""
    z = np.load(os.path.join(golden_dir, "lcm.npz"))
    am = lambda x, t, c=None: torch.tanh(0.6 * x + (t.float() / 1000.0)[:, None, None, None]) * 0.8 + 0.1 * x
    den = kd.LCMCompVisDenoiser(am, kd.make_alphas_cumprod())
    assert np.array_equal(den.sigmas.numpy(), z["sigmas"]) and np.array_equal(den.get_sigmas().numpy(), z["get_sigmas_all"])
    assert np.array_equal(den.sigma_to_t(torch.tensor(z["probe_sigma"])).numpy(), z["probe_t"])
    np.testing.assert_allclose(den.t_to_sigma(torch.tensor([0., 19., 59., 333., 500.5, 999., 1200.])).numpy(), z["probe_t_to_sigma"], rtol=1e-6)
    x = seeded((2, 4, 8, 8), 4100)
    for k, sg in enumerate([14.6, 2.2, 0.4]):
        np.testing.assert_allclose(den(x * sg, torch.full((2,), sg), None).numpy(), z[f"forward{k}"], rtol=0, atol=1e-6)
    for ci, steps in enumerate([4, 8]):
        sig = den.get_sigmas(steps)
        np.testing.assert_allclose(sig.numpy(), z[f"c{ci}_sigmas"], rtol=1e-6)
        draws = iter([seeded((2, 4, 8, 8), 4200 + 10 * ci + i) for i in range(steps)])
        out = kd.sample_lcm(lambda xx, s, **kw: den(xx, s, None), seeded((2, 4, 8, 8), 4190 + ci) * sig[0], sig, {}, lambda: next(draws))
        np.testing.assert_allclose(out.numpy(), z[f"c{ci}_out"], rtol=0, atol=1e-6)


def _golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cfg_denoiser_matches_reference(golden_dir):
    """oracle CFGDenoiser == the reference class (modules/sd_samplers_cfg_denoiser.py:35-311) executed by make_golden over twenty-two
    scenarios: plain CFG, AND composition, NGMS / skip-early skip-uncond (odd, even, all steps, sigma above the threshold, with
    AND), mask before / after, cond and uncond of different token counts (two calls, pad_cond_uncond, pad_cond_uncond_v0, both
    directions), CFG++ bookkeeping, unbatched cond / uncond, InstructPix2Pix three-way CFG and its scale-1 fallback, unCLIP's c_adm rows
    (zeros for uncond; with AND + NGMS)."""
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "cfg_denoiser.npz"))
    assert len(mg.CFG_SCENARIOS) == 22
    for k, (name, sc) in enumerate(mg.CFG_SCENARIOS):
        inp = mg.cfg_scenario_inputs(k, sc)
        d = kd.CFGDenoiser(lambda xi, si, c, ic: mg.cfg_inner_model(xi, si, c, ic))
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
        out = d(inp["x"].clone(), inp["sigma"], inp["uncond"], (inp["conds_list"], inp["cond"]), 7.0, sc.get("s_min_uncond", 0.0),
                inp["image_cond"])
        fl = z[name + "_flags"]
        np.testing.assert_allclose(out.numpy(), z[name + "_denoised"], rtol=0, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(d.last_latent.numpy(), z[name + "_last_latent"], rtol=0, atol=1e-6, err_msg=name)
        assert (int(d.padded_cond_uncond), int(d.padded_cond_uncond_v0), d.step) == tuple(fl[:3]), name
        assert d.skipped_uncond == bool(fl[3] or fl[4]), name
        if d.need_last_noise_uncond:
            np.testing.assert_allclose(d.last_noise_uncond.numpy(), z[name + "_last_noise_uncond"], rtol=0, atol=1e-6)


def test_image_conditioning_builders_match_reference(golden_dir):
    """oracle/pipeline.py txt2img_image_conditioning / inpainting_image_conditioning / edit_image_conditioning == the reference
    functions (modules/processing.py:100-133, 321-324, 332-374) exec'd over a stand-in first stage: rounded / soft / absent mask
    and a mask weight below 1."""
    from oracle import pipeline as opipe
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "image_conditioning.npz"))
    fs = mg.FakeFirstStage()
    np.testing.assert_allclose(opipe.txt2img_image_conditioning(fs, 3, 12, 16).numpy(), z["txt2img"], rtol=0, atol=1e-6)
    img = seeded((2, 3, 12, 16), 6101).clamp(-1, 1)
    mask = torch.from_numpy(z["mask_u8"].astype(np.float32) / 255.0)[None, None]
    f = lambda **kw: opipe.inpainting_image_conditioning(fs, img, (6, 8), **kw).numpy()
    np.testing.assert_allclose(f(image_mask=mask), z["inpaint_round"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(f(image_mask=mask, round_image_mask=False), z["inpaint_soft"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(f(), z["inpaint_nomask"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(f(image_mask=mask, mask_weight=0.35), z["inpaint_weight"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(opipe.edit_image_conditioning(fs, img).numpy(), z["edit"], rtol=0, atol=1e-6)
    assert np.abs(z["inpaint_round"] - z["inpaint_soft"]).max() > 0.1


def test_resize_image_matches_reference(golden_dir):
    """oracle resize_image_mode0 == images.resize_image(0, ...) + the Upscaler loop + Lanczos / Nearest / None scalers
    (modules/images.py:252-291, modules/upscaler.py:54-154) exec'd by make_golden: bit-identical uint8 images."""
    from PIL import Image
    from oracle import pipeline as opipe
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "resize_image.npz"))
    for k, (w, h, name) in enumerate(mg.RESIZE_CASES):
        assert np.array_equal(np.array(opipe.resize_image_mode0(Image.fromarray(z["base"]), w, h, name)), z[f"r{k}"]), (w, h, name)
    assert opipe.hires_target_resolution(512, 768, 2.0) == (1024, 1536, 0, 0)
    assert opipe.hires_target_resolution(512, 768, hr_resize_x=1000) == (1000, 1500, 0, 0)
    assert opipe.hires_target_resolution(512, 768, hr_resize_y=900) == (600, 900, 0, 0)
    assert opipe.hires_target_resolution(512, 768, hr_resize_x=1024, hr_resize_y=1024) == (1024, 1536, 0, 64)
    assert opipe.hires_target_resolution(768, 512, hr_resize_x=1024, hr_resize_y=1024) == (1536, 1024, 64, 0)


def test_refiner_switch_decision_matches_reference(golden_dir):
    """oracle refiner_due == apply_refiner (modules/sd_samplers_common.py:158-190) exec'd by make_golden over 50 cases: sigma- and
    timestep-space progress, switch by sampling steps, missing / already active refiner, the three hires-pass options."""
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "refiner.npz"))["decisions"]
    table = kd.CompVisDenoiser(None, kd.make_alphas_cumprod()).sigmas
    cases = mg.refiner_cases()
    assert len(cases) == len(z) == 50 and 10 < z.sum() < 40
    for want, (step, total, sigma, sigma_space, switch_at, by_steps, has_ref, already, enable_hr, is_hr, hopt) in zip(z, cases):
        got = kd.refiner_due(step, total, None if sigma is None else torch.full((2,), float(sigma)), table if sigma_space else None,
                             switch_at, has_ref, already, by_steps, enable_hr, is_hr, hopt)
        assert bool(got) == bool(want), (step, sigma, switch_at, by_steps, enable_hr, is_hr, hopt)


def test_lycoris_calc_updown_matches_reference(golden_dir):
    """oracle.lora.calc_updown == NetworkModule*.calc_updown of extensions-builtin/Lora/network_*.py (loaded by make_golden) for 29
    cases: LoRA / LoCon incl. cp-decomposition, DoRA, dyn_dim; LoHa; LoKr; GLoRA; IA3; full; norm; OFT (kohya, constrained, conv,
    old LyCORIS rotation blocks) and BOFT (with rescale), modules with the dense "bias" entry (network.py:154, 196-199) — and the type
    dispatch."""
    from oracle import lora as olora
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "lyco.npz"))
    cases = {**mg.lyco_cases(), **mg.lyco_oft_cases(), **mg.lyco_bias_cases()}
    assert len(cases) == 29
    for k, (name, (kind, spec, build)) in enumerate(cases.items()):
        orig, w = mg.lyco_orig_weight(spec, k), build(9000 + 10 * k)
        assert olora.module_kind(w) == kind
        updown, ex_bias = olora.calc_updown(w, orig, 0.8, dyn_dim=3 if name == "lora_dyn" else None, with_bias=True)
        np.testing.assert_allclose(updown.numpy(), z[name + "_updown"], rtol=0, atol=1e-6, err_msg=name)
        if ex_bias is not None:
            np.testing.assert_allclose(ex_bias.numpy(), z[name + "_ex_bias"], rtol=0, atol=1e-6, err_msg=name)


def test_prompt_conditioning_containers_match_reference(golden_dir):
    """oracle/prompt_cond.py == modules/prompt_parser.py (get_multicond_prompt_list, reconstruct_cond_batch, stack_conds,
    reconstruct_multicond_batch; exec'd from the file's text by make_golden): AND splitting / weights of 7 prompts, and the
    per-step selection over prompt-editing schedules for tensor and dict (SDXL) conds, incl. padding of shorter conds."""
    import json
    from oracle import prompt_cond as opc
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "prompt_cond.npz"))
    meta = json.load(open(os.path.join(golden_dir, "prompt_cond.json")))
    per_prompt, flat, _ = opc.multicond_prompt_list(mg.MULTICOND_PROMPTS)
    assert [[list(x) for x in e] for e in per_prompt] == meta["res_indexes"] and flat == meta["flat"]
    for dict_conds in (False, True):
        tag = "dict_" if dict_conds else ""
        multi, uncond = mg.prompt_cond_schedules(dict_conds)
        batch = [[opc.Composable([opc.Scheduled(e, t) for e, t in sch], w) for sch, w in img] for img in multi]
        unc = [[opc.Scheduled(e, t) for e, t in sch] for sch in uncond]
        for step in (0, 2, 3, 4, 5, 9, 10, 25):
            conds_list, stacked = opc.reconstruct_multicond_batch(batch, step)
            u = opc.reconstruct_cond_batch(unc, step)
            assert [[list(x) for x in e] for e in conds_list] == meta[f"{tag}conds_list_{step}"]
            if dict_conds:
                for k in ("crossattn", "vector"):
                    assert np.array_equal(stacked[k].numpy(), z[f"{tag}c_{k}_{step}"]) and np.array_equal(u[k].numpy(), z[f"{tag}uc_{k}_{step}"])
            else:
                assert np.array_equal(stacked.numpy(), z[f"c_{step}"]) and np.array_equal(u.numpy(), z[f"uc_{step}"])


def test_image_rng_matches_reference(golden_dir):
    """oracle ImageRNG == modules/rng.py ImageRNG over the real rng_philox.py (randn_source "NV"), three draws each: plain,
    eta_noise_seed_delta, variation seeds (slerp branch, lerp branch, short subseed list), seed-resize (smaller, larger, mixed)
    and all of them together — bit-exact."""
    from oracle.rng import ImageRNG
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "image_rng.npz"))
    for name, shape, seeds, kw, ensd in mg.IMAGE_RNG_CASES:
        r = ImageRNG(shape, seeds, eta_noise_seed_delta=ensd, **kw)
        for k in range(3):
            assert np.array_equal(r.next().numpy(), z[f"{name}_{k}"]), (name, k)


def test_unet_pieces_match_in_tree_twins(golden_dir):
    """The pieces of ldm's UNet the webui re-implements in-tree, executed by make_golden on the oracle's own module instances:
    timestep_embedding and spatial_transformer_forward (modules/sd_hijack_unet.py:56-102) and the baseline
    attention_CrossAttention_forward (modules/hypernetworks/hypernetwork.py:382-407) — the oracle's forwards of the same modules
    give the same bits (conv and linear proj_in / proj_out, self- and cross-attention, even and odd embedding widths)."""
    from oracle import unet as ou
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "unet_twins.npz"))
    attn_self, attn_cross, st_conv, st_lin = mg.unet_twin_modules()
    t = torch.tensor([999.0, 500.25, 37.5, 0.0])
    assert np.array_equal(ou.timestep_embedding(t, 320).numpy(), z["temb_320"])
    assert np.array_equal(ou.timestep_embedding(t, 65).numpy(), z["temb_65"])
    x, ctx, img = seeded((2, 40, 64), 9600), seeded((2, 77, 48), 9601), seeded((2, 64, 6, 5), 9602)
    with torch.no_grad():
        np.testing.assert_allclose(attn_self(x).numpy(), z["attn_self"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(attn_cross(x, context=ctx).numpy(), z["attn_cross"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(st_conv(img, context=ctx).numpy(), z["st_conv"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(st_lin(img, context=ctx).numpy(), z["st_lin"], rtol=0, atol=1e-6)


def test_euler_sampler_matches_in_tree_twin(golden_dir):
    """oracle sample_euler / to_d == the copy of k-diffusion's Euler sampler the reference carries in-tree
    (modules/models/sd3/sd3_impls.py:145-163), on a Karras and a model-uniform schedule — same bits."""
    z = np.load(os.path.join(golden_dir, "euler_twin.npz"))

    def model(x, sigma, **kw):
        s = sigma[:, None, None, None]
        return x / (1 + s * s) + torch.tanh(0.5 * x) * (s * s / (1 + s * s)) * 0.3

    for ci in range(2):
        sigmas = torch.from_numpy(z[f"c{ci}_sigmas"])
        out = kd.sample_euler(model, seeded((2, 4, 8, 8), 9700 + ci) * sigmas[0], sigmas, {})
        assert np.array_equal(out.numpy(), z[f"c{ci}_out"]), ci


def test_zero_terminal_snr_schedule_matches_reference(golden_dir):
    """oracle rescale_zero_terminal_snr_abar == modules/sd_models.py:628-644 exec'd by make_golden (fp32 and fp16-downcast input);
    the rescaled schedule ends at the reference's constant and keeps its first entry."""
    z = np.load(os.path.join(golden_dir, "zsnr.npz"))
    ac = kd.make_alphas_cumprod()
    np.testing.assert_array_equal(kd.rescale_zero_terminal_snr_abar(ac.clone()).numpy(), z["fp32"])
    np.testing.assert_array_equal(kd.rescale_zero_terminal_snr_abar(ac.clone().half()).float().numpy(), z["downcast"])
    assert z["fp32"][-1] == np.float32(4.8973451890853435e-08) and abs(z["fp32"][0] - float(ac[0])) < 1e-7


def test_schedulers_match_reference_functions(golden_dir):
    """oracle/schedulers.py == the functions of modules/sd_schedulers.py executed by tests/golden/make_golden.py (sgm_uniform,
    kl_optimal, align_your_steps incl. the SDXL table, simple, normal, ddim, beta, uniform), same table of names / labels /
    default_rho / need_inner_model as :130-143."""
    from oracle import schedulers as osch
    z = np.load(os.path.join(golden_dir, "schedulers.npz"))
    inner = kd.CompVisDenoiser(None, kd.make_alphas_cumprod())
    smin, smax = inner.sigmas[0].item(), inner.sigmas[-1].item()
    assert list(z["names"])[1:] == list(osch.SCHEDULERS) and z["names"][0] == "automatic"
    for i, name in enumerate(z["names"]):
        if name != "automatic":
            assert bool(z["need_inner_model"][i]) == osch.SCHEDULERS[name][1], name
    checked = 0
    for n in (5, 11, 20, 50):
        for name, (fn, need_inner) in osch.SCHEDULERS.items():
            key = f"{name}_{n}"
            if key not in z.files:
                continue
            got = fn(n, smin, smax, inner) if need_inner else fn(n, smin, smax)
            np.testing.assert_allclose(torch.as_tensor(got).float().numpy(), z[key], rtol=1e-6, atol=1e-7, err_msg=key)
            checked += 1
    assert checked == 4 * 8
    np.testing.assert_allclose(osch.align_your_steps(20, smin, smax, is_sdxl=True).numpy(), z["align_your_steps_sdxl_20"], rtol=1e-6)
    # third-party get_sigmas_* (restated): end points and monotonicity
    for fn in (osch.get_sigmas_exponential, osch.get_sigmas_polyexponential, kd.get_sigmas_karras):
        s = fn(20, smin, smax)
        assert s.shape == (21,) and abs(s[0].item() - smax) < 1e-3 and abs(s[-2].item() - smin) < 1e-5 and s[-1] == 0
        assert bool((s[:-1] > s[1:]).all())


def test_lora_layer_names_match_reference_function(golden_dir):
    """oracle.lora.convert_diffusers_name_to_compvis == extensions-builtin/Lora/networks.py:56-120 on 141 kohya-style keys."""
    import json
    from oracle import lora as olora
    z = json.load(open(os.path.join(golden_dir, "lora_names.json")))
    for k, want in z["sd1"].items():
        assert olora.convert_diffusers_name_to_compvis(k, False) == want, k
    for k, want in z["sd2"].items():
        assert olora.convert_diffusers_name_to_compvis(k, True) == want, k


def test_lora_merge_arithmetic():
    """W' = W + (up @ down).reshape(W.shape) * alpha/dim * multiplier (network_lora.py:65-80, network.py:167-216), for a linear,
    a 1x1 conv given as 2-D factors and a 3x3 LoCon layer; two networks add up in list order."""
    from oracle import lora as olora
    g = torch.Generator().manual_seed(0)
    sd = {"model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight": torch.randn(64, 64, generator=g),
          "model.diffusion_model.input_blocks.1.1.proj_in.weight": torch.randn(64, 64, 1, 1, generator=g),
          "model.diffusion_model.input_blocks.1.0.in_layers.2.weight": torch.randn(64, 64, 3, 3, generator=g),
          "model.diffusion_model.input_blocks.1.0.in_layers.2.bias": torch.randn(64, generator=g)}
    lora = {"lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_up.weight": torch.randn(64, 4, generator=g),
            "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_down.weight": torch.randn(4, 64, generator=g),
            "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.alpha": torch.tensor(2.0),
            "lora_unet_down_blocks_0_attentions_0_proj_in.lora_up.weight": torch.randn(64, 8, 1, 1, generator=g),
            "lora_unet_down_blocks_0_attentions_0_proj_in.lora_down.weight": torch.randn(8, 64, 1, 1, generator=g),
            "lora_unet_down_blocks_0_resnets_0_conv1.lora_up.weight": torch.randn(64, 4, 1, 1, generator=g),
            "lora_unet_down_blocks_0_resnets_0_conv1.lora_down.weight": torch.randn(4, 64, 3, 3, generator=g),
            "lora_unet_down_blocks_0_resnets_0_conv1.alpha": torch.tensor(4.0),
            "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight": torch.randn(8, 2, generator=g)}
    out = olora.merge(sd, [(lora, 0.8)])
    k = "model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"
    up, down = lora["lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_up.weight"], \
        lora["lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_down.weight"]
    np.testing.assert_allclose(out[k].numpy(), (sd[k] + up @ down * (2.0 / 4) * 0.8).numpy(), rtol=1e-6, atol=1e-6)
    k = "model.diffusion_model.input_blocks.1.0.in_layers.2.weight"
    up, down = lora["lora_unet_down_blocks_0_resnets_0_conv1.lora_up.weight"], lora["lora_unet_down_blocks_0_resnets_0_conv1.lora_down.weight"]
    want = sd[k] + (up.reshape(64, 4) @ down.reshape(4, -1)).reshape(64, 64, 3, 3) * (4.0 / 4) * 0.8
    np.testing.assert_allclose(out[k].numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    assert torch.equal(out["model.diffusion_model.input_blocks.1.0.in_layers.2.bias"], sd["model.diffusion_model.input_blocks.1.0.in_layers.2.bias"])
    twice = olora.merge(sd, [(lora, 0.8), (lora, -0.8)])
    for kk in sd:
        np.testing.assert_allclose(twice[kk].numpy(), sd[kk].numpy(), atol=2e-5)
    _, failed = olora.group_network(lora, olora.layer_mapping(sd))
    assert list(failed) == ["lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight"]


def _seed_clip_like_fixture(model, ci):
    """Same seeding as tests/golden/make_golden.py::gen_clip (parameters visited in state-dict order)."""
    g = torch.Generator().manual_seed(4000 + ci)
    with torch.no_grad():
        for name, prm in model.state_dict().items():
            if name.endswith("embedding.weight"):
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.5)
            elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("layer_norm.weight"):
                prm.copy_(1.0 + 0.02 * torch.randn(prm.shape, generator=g))
            elif name.endswith(".bias"):
                prm.copy_(0.02 * torch.randn(prm.shape, generator=g))
            else:
                prm.copy_(torch.randn(prm.shape, generator=g) * prm.shape[-1] ** -0.5)


def test_clip_text_model_matches_reference_twin(golden_dir):
    """oracle.clip.ClipTextModel == modules/models/sd3/other_impls.py CLIPTextModel_ (last hidden state, clip-skip-2 + final
    norm, pooled), quick_gelu and gelu."""
    from oracle import clip as oclip
    z = np.load(os.path.join(golden_dir, "clip_text.npz"))
    for ci, act in enumerate(("quick_gelu", "gelu")):
        cfg = oclip.ClipConfig(vocab_size=49408, hidden=128, layers=3, heads=2, intermediate=256, act=act)
        m = oclip.ClipTextModel(cfg).eval()
        _seed_clip_like_fixture(m, ci)
        tok = torch.from_numpy(z[f"c{ci}_tokens"])
        last, pooled = m(tok, skip=1, return_pooled=True)
        np.testing.assert_allclose(last.numpy(), z[f"c{ci}_last"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(pooled.numpy(), z[f"c{ci}_pooled"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(m(tok, skip=2).numpy(), z[f"c{ci}_skip2"], rtol=0, atol=2e-5)


def test_clip_text_model_matches_installed_transformers():
    """The third-party network itself, when importable here: transformers' CLIPTextModel on a small random configuration —
    last_hidden_state, hidden_states[-2] + final_layer_norm (the clip-skip branch of sd_hijack_clip.py:354-356), pooler_output."""
    transformers = pytest.importorskip("transformers")
    from oracle import clip as oclip
    hf_cfg = transformers.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                         num_attention_heads=2, max_position_embeddings=77, hidden_act="quick_gelu",
                                         eos_token_id=999, bos_token_id=998, pad_token_id=999)
    torch.manual_seed(0)
    hf = transformers.CLIPTextModel(hf_cfg).eval()
    cfg = oclip.ClipConfig(vocab_size=1000, hidden=128, layers=3, heads=2, intermediate=256)
    # transformers 4.x keeps the network under ".text_model.", 5.x flattened it: normalise to the 4.x (checkpoint) layout
    sd = {"cond_stage_model.transformer." + (k if k.startswith("text_model.") else "text_model." + k): v
          for k, v in hf.state_dict().items()}
    m = oclip.build_clip(cfg, sd)
    tok = torch.randint(0, 998, (2, 77), generator=torch.Generator().manual_seed(1))
    tok[:, 0] = 998
    tok[0, 30:] = 999
    tok[1, 76] = 999
    with torch.no_grad():
        out = hf(input_ids=tok, output_hidden_states=True)
        np.testing.assert_allclose(m(tok).numpy(), out.last_hidden_state.numpy(), rtol=0, atol=2e-5)
        want2 = getattr(hf, "text_model", hf).final_layer_norm(out.hidden_states[-2])
        np.testing.assert_allclose(m(tok, skip=2).numpy(), want2.numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(m(tok, skip=2, apply_final_ln=False).numpy(), out.hidden_states[-2].numpy(), rtol=0, atol=2e-5)
        _, pooled = m(tok, return_pooled=True)
        np.testing.assert_allclose(pooled.numpy(), out.pooler_output.numpy(), rtol=0, atol=2e-5)


def test_schedule_known_answers():
    """SURVEY.md appendix A.3 (in-tree hints modules/shared_options.py:396-397, sd_schedulers.py:60-63)."""
    ac = kd.make_alphas_cumprod()
    assert abs(ac[0].item() - 0.99915) < 1e-6 and abs(ac[999].item() - 0.0046601) < 1e-6
    den = kd.CompVisDenoiser(None, ac)
    assert abs(den.sigma_min.item() - 0.0291672) < 1e-6
    assert abs(den.sigma_max.item() - 14.614641) < 1e-4
    want = [14.6146, 10.7468, 8.0815, 6.2049, 4.8557, 3.8654, 3.1238, 2.5572, 2.1157, 1.7648, 1.4806, 1.2458,
            1.0481, 0.8784, 0.7297, 0.5964, 0.4736, 0.3555, 0.2322, 0.0292, 0.0]
    np.testing.assert_allclose(den.get_sigmas(20).numpy(), want, atol=6e-5)
    ks = kd.get_sigmas_karras(50, den.sigma_min.item(), den.sigma_max.item())
    np.testing.assert_allclose(ks[:5].numpy(), [14.6146, 13.4292, 12.3272, 11.3036, 10.3538], atol=2e-4)
    np.testing.assert_allclose(ks[-4:].numpy(), [0.0434, 0.0357, 0.0292, 0.0], atol=1e-4)
    assert kd.ddim_timesteps(20).tolist() == list(range(1, 1000, 50))
    # sigma_to_t inverts t_to_sigma on the grid and in between
    t = torch.tensor([0.0, 10.5, 500.25, 998.0])
    np.testing.assert_allclose(den.sigma_to_t(den.t_to_sigma(t)).numpy(), t.numpy(), atol=2e-3)
    assert kd.setup_img2img_steps(20, 0.75) == (26, 19)


def test_structural_checksums():
    with torch.device("meta"):
        n15 = sum(p.numel() for p in ounet.UNetModel(ounet.sd15_config()).parameters())
        nxl = sum(p.numel() for p in ounet.UNetModel(ounet.sdxl_base_config()).parameters())
        nv = sum(p.numel() for p in ovae.AutoencoderKL(ovae.sd15_vae_config()).parameters())
    assert n15 == 859_520_964 and nxl == 2_567_463_684 and nv == 83_653_863


def test_timestep_embedding_cos_first():
    """modules/sd_hijack_unet.py:58-78"""
    e = ounet.timestep_embedding(torch.tensor([0.0, 3.0]), 320)
    assert e.shape == (2, 320)
    assert torch.all(e[0, :160] == 1) and torch.all(e[0, 160:] == 0)
    assert abs(e[1, 0].item() - math.cos(3.0)) < 1e-6 and abs(e[1, 160].item() - math.sin(3.0)) < 1e-6


def test_cfg_combine_and_euler_ancestral_step():
    """Known-answer on an analytic denoiser: one Euler-a step by hand."""
    x = seeded((2, 4, 4, 4), 1)
    sig = torch.tensor([2.0, 1.0, 0.0])
    noise = seeded((2, 4, 4, 4), 2)
    model = lambda x, s, **kw: x * 0.5
    out = kd.sample_euler_ancestral(model, x, sig[:2], {}, lambda: noise)   # one step 2.0 -> 1.0
    su = min(1.0, (1.0 * (4.0 - 1.0) / 4.0) ** 0.5)
    sd = (1.0 - su * su) ** 0.5
    want = x + (x - 0.5 * x) / 2.0 * (sd - 2.0) + noise * su
    np.testing.assert_allclose(out.numpy(), want.numpy(), atol=1e-6)
    xo = torch.stack([torch.full((1,), 3.0), torch.full((1,), 5.0), torch.full((1,), 1.0), torch.full((1,), 2.0)])
    den = kd.CFGDenoiser.combine_denoised(xo, [[(0, 1.0)], [(1, 1.0)]], 2, 7.0)
    assert den.flatten().tolist() == [1.0 + (3.0 - 1.0) * 7.0, 2.0 + (5.0 - 2.0) * 7.0]


def test_hypernetwork_module_and_chain_match_reference(golden_dir):
    """oracle.hypernetwork.HypernetworkModule / apply_hypernetworks == the reference's class and functions
    (modules/hypernetworks/hypernetwork.py:25-113, 358-379), exec'd from its own text by tests/golden/make_golden.py:gen_hypernetwork."""
    from oracle import hypernetwork as ohn
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "hypernetwork.npz"))
    mods = {}
    with torch.no_grad():
        for k, (name, (dim, ls, act, ln, ao, ds)) in enumerate(mg.HN_CASES.items()):
            m = ohn.HypernetworkModule(dim, None, ls, act, ln, ao, ds)
            seeded_module_weights(m, 6000 + k)
            m.multiplier = 0.7
            assert rel_l2(m(seeded((2, 10, dim), 6100 + k)), z[name]) < 1e-6, name
            mods[name] = m
        a, b = ohn.Hypernetwork({}), ohn.Hypernetwork({})
        a.layers = {64: (mods["relu_121"], mods["lin_121"])}
        b.layers = {64: (mods["tanh_131_ao"], mods["swish_ln_1221_ao"]), 128: (mods["elu_drop_1221"], mods["sigmoid_121"])}
        ck, cv = ohn.apply_hypernetworks([a, b], seeded((2, 7, 64), 6200))
        assert rel_l2(ck, z["chain_k"]) < 1e-6 and rel_l2(cv, z["chain_v"]) < 1e-6
