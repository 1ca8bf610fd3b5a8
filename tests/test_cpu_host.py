"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, host logic matches the oracle,
and the multi-process path works over gloo with world_size 2."""
import importlib
import os
import types
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from oracle import kdiffusion as okd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sub(name):
    return importlib.import_module("stable-diffusion-webui_amd." + name)


def test_library_loads_and_exports_every_declared_symbol():
    lib = sub("_lib")
    declared = lib.declared_symbols()
    assert len(declared) >= 30
    for s in declared:
        assert hasattr(lib.lib, s), f"libsdmi.so does not export {s}"
    assert set(declared) == set(lib._SIGS), "ctypes signature table out of sync with include/sdmi.h"
    assert lib.lib.sdmi_version() == 100


def test_product_path_fails_loudly_without_gpu():
    lib = sub("_lib")
    if lib.device_ok():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.SdmiError):
        sub("engine").Engine(0)
    with pytest.raises(lib.SdmiError):
        sub("ops").philox_randn((4,), 0, 0, "cpu")


def test_schema_matches_oracle_modules():
    schema = sub("schema")
    from oracle import unet as ou, vae as ov
    with torch.device("meta"):
        pairs = [(schema.unet_schema(schema.sd15_unet()), ou.UNetModel(ou.sd15_config())),
                 (schema.unet_schema(schema.sdxl_unet()), ou.UNetModel(ou.sdxl_base_config())),
                 (schema.unet_schema(schema.tiny_unet()), ou.UNetModel(ou.tiny_config())),
                 (schema.vae_schema(schema.sd15_vae()), ov.AutoencoderKL(ov.sd15_vae_config()))]
    for entries, mod in pairs:
        want = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        got = {k: s for k, s, _ in entries}
        assert got == want


def test_host_schedule_matches_oracle():
    ss = sub("sd_samplers")

    class M:
        alphas_cumprod = sub("schema").make_alphas_cumprod()
    prod = ss.CompVisDenoiser(M())
    ora = okd.CompVisDenoiser(None, okd.make_alphas_cumprod())
    assert torch.equal(prod.sigmas, ora.sigmas)
    for n in (5, 20, 26, 50):
        assert torch.equal(prod.get_sigmas(n), ora.get_sigmas(n))
        assert torch.equal(ss.get_sigmas_karras(n, prod.sigmas[0].item(), prod.sigmas[-1].item()),
                           okd.get_sigmas_karras(n, ora.sigmas[0].item(), ora.sigmas[-1].item()))
    s = torch.tensor([14.6146, 3.3, 0.5, 0.0292])
    assert torch.equal(prod.sigma_to_t(s), ora.sigma_to_t(s))
    a = ss.get_ancestral_step(torch.tensor(2.0), torch.tensor(1.0))
    b = okd.get_ancestral_step(torch.tensor(2.0), torch.tensor(1.0))
    assert float(a[0]) == float(b[0]) and float(a[1]) == float(b[1])


def test_sampler_registry_and_sigma_selection():
    ss = sub("sd_samplers")
    assert {"Euler a", "Euler", "DPM++ 2M", "DDIM"} <= set(ss.all_samplers_map)
    assert ss.find_sampler_config("k_euler_a").name == "Euler a"

    class M:
        alphas_cumprod = sub("schema").make_alphas_cumprod()
        engine = None

    class P:
        scheduler = None
        is_hr_pass = False
        sampler_noise_scheduler_override = None
        extra_generation_params = {}
    s = ss.create_sampler("DPM++ 2M", M())
    sig = s.get_sigmas(P(), 50)
    assert sig.shape == (51,) and abs(float(sig[1]) - 13.4292) < 2e-4      # Karras by default for DPM++ 2M
    assert P.extra_generation_params == {"Schedule type": "Karras"}        # the infotext key of sd_samplers_kdiffusion.py:103-104
    s = ss.create_sampler("Euler a", M())
    assert abs(float(s.get_sigmas(P(), 20)[1]) - 10.7468) < 1e-4            # model schedule for Euler a
    P.scheduler = "Karras"
    assert abs(float(s.get_sigmas(P(), 50)[1]) - 13.4292) < 2e-4
    with pytest.raises(AssertionError):
        ss.create_sampler("no such sampler", M())

    class P2:
        steps = 20
        denoising_strength = 0.75
    assert ss.setup_img2img_steps(P2(), 20) == (26, 19)


def test_host_schedulers_match_reference_functions(golden_dir):
    """The product's sd_schedulers module (host side, like the reference's) against the same reference-generated fixture the
    oracle is pinned with, plus the table itself (names, labels, default_rho, need_inner_model, aliases)."""
    hs, ss = sub("sd_schedulers"), sub("sd_samplers")
    z = np.load(os.path.join(golden_dir, "schedulers.npz"))
    assert [x.name for x in hs.schedulers] == list(z["names"]) and [x.label for x in hs.schedulers] == list(z["labels"])
    assert [x.default_rho for x in hs.schedulers] == list(z["default_rho"])
    assert [x.need_inner_model for x in hs.schedulers] == list(z["need_inner_model"])
    assert hs.schedulers_map["SGM Uniform"] is hs.schedulers_map["sgm_uniform"]

    class M:
        alphas_cumprod = sub("schema").make_alphas_cumprod()
    inner = ss.CompVisDenoiser(M())
    smin, smax = inner.sigmas[0].item(), inner.sigmas[-1].item()
    for n in (5, 11, 20, 50):
        for sch in hs.schedulers:
            key = f"{sch.name}_{n}"
            if key not in z.files:
                continue
            got = sch.function(n, smin, smax, inner, "cpu") if sch.need_inner_model else sch.function(n, smin, smax, "cpu")
            np.testing.assert_allclose(got.float().numpy(), z[key], rtol=1e-6, atol=1e-7, err_msg=key)
    from oracle import schedulers as osch
    for n in (7, 30):
        assert torch.equal(hs.get_sigmas_exponential(n, smin, smax), osch.get_sigmas_exponential(n, smin, smax))
        assert torch.equal(hs.get_sigmas_polyexponential(n, smin, smax, 1.0), osch.get_sigmas_polyexponential(n, smin, smax, 1.0))


def test_sampler_table_matches_reference_rows():
    """Labels, aliases and options of the implemented rows of modules/sd_samplers_kdiffusion.py:11-27 and
    modules/sd_samplers_timesteps.py:12-17; scheduler selection and the discarded penultimate sigma follow :79-132."""
    ss = sub("sd_samplers")
    rows = {x.name: x for x in ss.all_samplers}
    assert rows["DPM2"].options == {'scheduler': 'karras', 'discard_next_to_last_sigma': True, "second_order": True}
    assert rows["DPM2 a"].aliases == ['k_dpm_2_a'] and rows["DPM++ 2S a"].options["uses_ensd"] is True
    assert rows["Heun"].options == {"second_order": True} and rows["LMS"].aliases == ['k_lms']
    assert {"PLMS", "DDIM"} <= set(rows)
    assert ss.sampler_extra_params['sample_heun'] == ['s_churn', 's_tmin', 's_tmax', 's_noise']

    class M:
        alphas_cumprod = sub("schema").make_alphas_cumprod()
        engine = None

    class P:
        scheduler = None
        is_hr_pass = False
        sampler_noise_scheduler_override = None
        extra_generation_params = {}
    from oracle import pipeline as opipe
    ora = okd.CompVisDenoiser(None, okd.make_alphas_cumprod())
    for name, key in (("DPM2", "dpm_2"), ("DPM2 a", "dpm_2_a"), ("Heun", "heun"), ("LMS", "lms"), ("DPM++ 2S a", "dpmpp_2s_a")):
        got = ss.create_sampler(name, M()).get_sigmas(P(), 12)
        want = opipe.get_sigmas(ora, key, 12)
        assert torch.equal(got, want), name
    assert ss.create_sampler("DPM2", M()).get_sigmas(P(), 12).shape == (13,)       # 13 + 1 sigmas, penultimate dropped
    P.scheduler = "Align Your Steps"
    assert abs(float(ss.create_sampler("Euler", M()).get_sigmas(P(), 11)[1]) - 6.475) < 1e-5
    P.scheduler = "no such scheduler"
    with pytest.raises(NotImplementedError):
        ss.create_sampler("Euler", M()).get_sigmas(P(), 10)


def test_host_unipc_coefficients_match_reference(golden_dir, monkeypatch):
    """The host UniPC (sd_samplers.unipc) folds each predictor / corrector update into ONE linear combination evaluated by
    sdmi_lincomb.  Here the device launch is replaced by the same sum in torch (this test only — the product has no such
    path) so that the host schedule / coefficient logic is checked against the reference-generated fixture on the CPU."""
    ss = sub("sd_samplers")
    z = np.load(os.path.join(golden_dir, "unipc.npz"))
    monkeypatch.setattr(ss, "_lc", lambda out, terms, coefs: out.copy_(sum(float(c) * t for c, t in zip(coefs, terms))))
    ac = sub("schema").make_alphas_cumprod()

    class Model:
        inner_model = type("I", (), {"inner_model": type("M", (), {"alphas_cumprod": ac})})

        def __call__(self, x, t, **_):
            return torch.tanh(0.7 * x + (t / 1000.0)[:, None, None, None]) * 0.9 + 0.05 * x

    from tests.test_oracle_pins import _unipc_cases, seeded
    opts = ss.shared.opts
    keep = (opts.uni_pc_variant, opts.uni_pc_skip_type, opts.uni_pc_order, opts.uni_pc_lower_order_final)
    try:
        for ci, ts, kw in _unipc_cases(z):
            opts.uni_pc_variant, opts.uni_pc_skip_type = kw["variant"], kw["skip_type"]
            opts.uni_pc_order, opts.uni_pc_lower_order_final = kw["order"], kw["lower_order_final"]
            dens = []
            batch = int(z[f"c{ci}_batch"][0])          # vary_coeff fixtures are batch 1: the reference's own code breaks beyond that
            out = ss.unipc(Model(), seeded((batch, 4, 8, 8), 990 + ci), ts, extra_args={}, callback=lambda d: dens.append(d['denoised']),
                           is_img2img=kw["is_img2img"])
            # vary_coeff: the reference inverts its small systems in fp32 (torch.linalg per step), the host solves them once in float64 —
            # the fp32 error of the reference's own inverse is what the looser bound covers
            np.testing.assert_allclose(out.numpy(), z[f"c{ci}_out"], rtol=0, atol=1e-3 if kw["variant"] == "vary_coeff" else 2e-4)
            assert len(dens) == int(z[f"c{ci}_n_callbacks"][0]) and dens[-1] is None
    finally:
        opts.uni_pc_variant, opts.uni_pc_skip_type, opts.uni_pc_order, opts.uni_pc_lower_order_final = keep
    assert ss.find_sampler_config("unipc").name == "UniPC"


def test_host_lcm_schedule_and_folded_scalings_match_reference(golden_dir):
    """Host LCMCompVisDenoiser: sigma table / get_sigmas / sigma_to_t equal the reference-generated fixture, and the folded
    affine scalings reproduce its forward: denoised = eps * c_out + x * c_skip with eps evaluated at x * c_in."""
    ss = sub("sd_samplers")
    z = np.load(os.path.join(golden_dir, "lcm.npz"))
    M = type("M", (), {"alphas_cumprod": sub("schema").make_alphas_cumprod(), "engine": None})
    s = ss.create_sampler("k_lcm", M())
    assert s.config.name == "LCM" and isinstance(s.model_wrap, ss.LCMCompVisDenoiser) and s.model_wrap_cfg.inner_model is s.model_wrap
    den = s.model_wrap
    assert np.array_equal(den.sigmas.numpy(), z["sigmas"])
    assert np.array_equal(den.sigma_to_t(torch.tensor(z["probe_sigma"])).numpy(), z["probe_t"])
    P = type("P", (), {"scheduler": None, "is_hr_pass": False, "sampler_noise_scheduler_override": None, "extra_generation_params": {}})
    for ci, steps in enumerate([4, 8]):
        np.testing.assert_allclose(s.get_sigmas(P(), steps).numpy(), z[f"c{ci}_sigmas"], rtol=1e-6)
    from tests.test_oracle_pins import seeded
    x = seeded((2, 4, 8, 8), 4100)
    for k, sg in enumerate([14.6, 2.2, 0.4]):
        c_skip, c_out, c_in = den.get_scalings(torch.tensor(sg))
        t = den.sigma_to_t(torch.full((2,), sg))
        xin = x * sg
        eps = torch.tanh(0.6 * (xin * c_in) + (t.float() / 1000.0)[:, None, None, None]) * 0.8 + 0.1 * (xin * c_in)
        np.testing.assert_allclose((eps * c_out + xin * c_skip).numpy(), z[f"forward{k}"], rtol=0, atol=2e-5)


def test_host_upscaler_handoff_matches_reference(golden_dir):
    """upscaler.resize_image / Upscaler.upscale / built-in scalers reproduce the reference-generated uint8 images, and
    StableDiffusionProcessingTxt2Img.calculate_target_resolution the reference's size / truncation arithmetic."""
    from PIL import Image
    from tests.test_oracle_pins import _golden_module
    from oracle import pipeline as opipe
    up, shared, processing = sub("upscaler"), sub("shared"), sub("processing")
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "resize_image.npz"))
    shared.sd_upscalers = up.builtin_upscalers()
    assert [x.name for x in shared.sd_upscalers] == ["None", "Lanczos", "Nearest"]
    for k, (w, h, name) in enumerate(mg.RESIZE_CASES):
        assert np.array_equal(np.array(up.resize_image(0, Image.fromarray(z["base"]), w, h, upscaler_name=name)), z[f"r{k}"]), (w, h, name)
    for kw in (dict(), dict(hr_resize_x=1000), dict(hr_resize_y=900), dict(hr_resize_x=1024, hr_resize_y=1024),
               dict(hr_resize_x=1024, hr_resize_y=640), dict(hr_scale=1.5)):
        for wh in ((512, 768), (768, 512), (512, 512)):
            p = processing.StableDiffusionProcessingTxt2Img(width=wh[0], height=wh[1], enable_hr=True, **kw)
            p.calculate_target_resolution()
            want = opipe.hires_target_resolution(wh[0], wh[1], kw.get("hr_scale", 2.0), kw.get("hr_resize_x", 0), kw.get("hr_resize_y", 0))
            assert (p.hr_upscale_to_x, p.hr_upscale_to_y, p.truncate_x, p.truncate_y) == want
    p = processing.StableDiffusionProcessingTxt2Img(enable_hr=True, hr_upscaler="ESRGAN_4x")
    with pytest.raises(Exception, match="could not find upscaler"):
        p.init(None, None, None)


def test_host_prompt_parser_containers_match_reference(golden_dir):
    """prompt_parser (host mirror): same results as the reference-generated fixture for AND splitting and the per-step
    reconstruction; get_learned_conditioning / get_multicond_learned_conditioning build the containers from an encoder and
    share the encoding of equal prompts; selection_key changes exactly when the selected entries change."""
    import json
    from tests.test_oracle_pins import _golden_module
    hp = sub("prompt_parser")
    mg = _golden_module()
    z = np.load(os.path.join(golden_dir, "prompt_cond.npz"))
    meta = json.load(open(os.path.join(golden_dir, "prompt_cond.json")))
    r = hp.get_multicond_prompt_list(mg.MULTICOND_PROMPTS)
    assert [[list(x) for x in e] for e in r[0]] == meta["res_indexes"] and list(r[1]) == meta["flat"] and r[2] == meta["indexes"]
    for dict_conds in (False, True):
        tag = "dict_" if dict_conds else ""
        multi, uncond = mg.prompt_cond_schedules(dict_conds)
        c = hp.MulticondLearnedConditioning((3,), [[hp.ComposableScheduledPromptConditioning(
            [hp.ScheduledPromptConditioning(e, t) for e, t in sch], w) for sch, w in img] for img in multi])
        uc = [[hp.ScheduledPromptConditioning(e, t) for e, t in sch] for sch in uncond]
        keys = []
        for step in (0, 2, 3, 4, 5, 9, 10, 25):
            conds_list, stacked = hp.reconstruct_multicond_batch(c, step)
            u = hp.reconstruct_cond_batch(uc, step)
            keys.append((hp.selection_key(c, step), hp.selection_key(uc, step)))
            assert [[list(x) for x in e] for e in conds_list] == meta[f"{tag}conds_list_{step}"]
            if dict_conds:
                assert tuple(stacked.shape) == tuple(stacked["crossattn"].shape)
                for k in ("crossattn", "vector"):
                    assert np.array_equal(stacked[k].numpy(), z[f"{tag}c_{k}_{step}"]) and np.array_equal(u[k].numpy(), z[f"{tag}uc_{k}_{step}"])
            else:
                assert np.array_equal(stacked.numpy(), z[f"c_{step}"]) and np.array_equal(u.numpy(), z[f"uc_{step}"])
        # cond selection: steps {0,2} | 3 | 4 | {5..9} | 10 | 25 (past every end: back to entry 0); uncond changes after step 3
        assert keys[0][0] == keys[1][0] != keys[2][0] and keys[3][0] != keys[4][0] and keys[6][0] != keys[7][0]
        assert keys[0][1] == keys[1][1] == keys[2][1] != keys[3][1]
        foreign = type("MulticondLearnedConditioning", (), {})()          # the webui's own class: same attributes, other type
        foreign.shape, foreign.batch = c.shape, c.batch
        assert hp.is_multicond(foreign) and hp.selection_key(foreign, 3) == hp.selection_key(c, 3) and not hp.is_multicond(torch.zeros(2, 3))
        assert hp.reconstruct_multicond_batch(foreign, 3)[0] == hp.reconstruct_multicond_batch(c, 3)[0]
        sl = hp.slice_conds(c, 1, 3, "cpu")
        assert isinstance(sl, hp.MulticondLearnedConditioning) and sl.shape == (2,) and sl.batch[0] is c.batch[1]

    class Enc:
        calls = []

        def get_learned_conditioning(self, texts):
            self.calls.append(list(texts))
            return torch.stack([torch.full((4, 2), float(len(t))) for t in texts])
    enc = Enc()
    mc = hp.get_multicond_learned_conditioning(enc, ["a cat AND a dog :0.5", "a cat"], 20)
    assert mc.shape == (2,) and [len(x) for x in mc.batch] == [2, 1] and mc.batch[0][1].weight == 0.5
    assert mc.batch[0][0].schedules is mc.batch[1][0].schedules and mc.batch[0][0].schedules[0].end_at_step == 20
    assert enc.calls == [["a cat"], [" a dog"]]
    sch = hp.get_learned_conditioning(enc, ["x"], 20, prompt_schedules=[[[5, "ab"], [20, "abcd"]]])
    assert [e.end_at_step for e in sch[0]] == [5, 20] and float(sch[0][1].cond[0, 0]) == 4.0


def test_host_alpha_schedule_override_matches_reference(golden_dir):
    """sd_models.apply_alpha_schedule_override (modules/sd_models.py:647-668): "Zero Terminal SNR" and the fp16 downcast start
    from alphas_cumprod_original every time, reproduce the reference-generated schedule, and change what the samplers see
    (sigma_max of the wrapped model)."""
    sdm, shared, ss = sub("sd_models"), sub("shared"), sub("sd_samplers")
    z = np.load(os.path.join(golden_dir, "zsnr.npz"))
    ac = sub("schema").make_alphas_cumprod()
    M = type("M", (), {})
    m = M()
    m.alphas_cumprod, m.alphas_cumprod_original, m.engine = ac.clone(), ac.clone(), None
    keep = (shared.opts.sd_noise_schedule, shared.opts.use_downcasted_alpha_bar)
    try:
        p = type("P", (), {"extra_generation_params": {}})()
        shared.opts.sd_noise_schedule = "Zero Terminal SNR"
        sdm.apply_alpha_schedule_override(m, p)
        np.testing.assert_allclose(m.alphas_cumprod.numpy(), z["fp32"], rtol=1e-6, atol=0)
        assert p.extra_generation_params == {"Noise Schedule": "Zero Terminal SNR"}
        assert float(ss.CompVisDenoiser(m).sigmas[-1]) > 4000          # sqrt((1 - a) / a) at a = 4.9e-8
        shared.opts.use_downcasted_alpha_bar = True
        sdm.apply_alpha_schedule_override(m)
        np.testing.assert_allclose(m.alphas_cumprod.float().numpy(), z["downcast"], rtol=2e-3, atol=1e-7)
        shared.opts.sd_noise_schedule, shared.opts.use_downcasted_alpha_bar = "Default", False
        sdm.apply_alpha_schedule_override(m)
        assert torch.equal(m.alphas_cumprod, ac) and abs(float(ss.CompVisDenoiser(m).sigmas[-1]) - 14.6146) < 1e-3
    finally:
        shared.opts.sd_noise_schedule, shared.opts.use_downcasted_alpha_bar = keep


def test_host_dpm_fast_matches_oracle_restatement(monkeypatch):
    """sd_samplers.sample_dpm_fast (folded lincomb coefficients, host fp32 step sizes) against the oracle's op-by-op DPMSolver
    restatement on an analytic denoiser — orders 3 / 2 / 1 and 3 / 3 / remainder, with and without the ancestral noise.  The
    device launch is replaced by the same sum in torch for this test only."""
    ss = sub("sd_samplers")
    monkeypatch.setattr(ss, "_lc", lambda out, terms, coefs: out.copy_(sum(float(c) * t for c, t in zip(coefs, terms))))
    from tests.test_oracle_pins import seeded
    den = okd.CompVisDenoiser(None, okd.make_alphas_cumprod())
    smin, smax = den.sigmas[0].item(), den.sigmas[-1].item()

    def model(x, sigma, **kw):
        s = sigma[:, None, None, None]
        return x / (1 + s * s) + torch.tanh(0.5 * x) * (s * s / (1 + s * s)) * 0.3

    for n, eta in ((6, 0.0), (6, 1.0), (8, 1.0), (10, 0.6), (3, 1.0)):
        x0 = seeded((2, 4, 8, 8), 9800 + n) * smax
        d1 = iter([seeded((2, 4, 8, 8), 9850 + i) for i in range(12)])
        d2 = iter([seeded((2, 4, 8, 8), 9850 + i) for i in range(12)])
        seen = []
        want = okd.sample_dpm_fast(model, x0.clone(), smin, smax, n, {}, lambda: next(d1), eta=eta, s_noise=0.9)
        got = ss.sample_dpm_fast(model, x0.clone(), smin, smax, n, extra_args={}, callback=lambda d: seen.append(float(d["sigma"])),
                                 eta=eta, s_noise=0.9, noise_sampler=lambda *a: next(d2))
        assert float((got - want).norm() / want.norm()) < 2e-5, (n, eta)
        assert len(seen) == n // 3 + 1                        # floor(n / 3) + 1 solver steps, one callback each
    assert ss.find_sampler_config("k_dpm_fast").name == "DPM fast" and ss.sampler_extra_params["sample_dpm_fast"] == ["s_noise"]


def test_host_lora_names_and_grouping_match_reference(golden_dir):
    """networks.convert_diffusers_name_to_compvis against the reference-generated fixture, and load_network's grouping /
    layer lookup (extensions-builtin/Lora/networks.py:183-240) on the tiny UNet's layer map."""
    import json
    nets, schema = sub("networks"), sub("schema")
    z = json.load(open(os.path.join(golden_dir, "lora_names.json")))
    for k, want in z["sd1"].items():
        assert nets.convert_diffusers_name_to_compvis(k, False) == want, k
    for k, want in z["sd2"].items():
        assert nets.convert_diffusers_name_to_compvis(k, True) == want, k

    class M:
        unet_cfg = schema.tiny_unet()
    m = M()
    mapping = nets.assign_network_names_to_compvis_modules(m)
    assert mapping["diffusion_model_input_blocks_1_1_transformer_blocks_0_attn1_to_q"][0] == "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"
    # norm layers are addressable too (LyCORIS norm modules rewrite their gain / shift through sdmi_unet_update_vector)
    assert mapping["diffusion_model_input_blocks_1_0_in_layers_0"] == ("input_blocks.1.0.in_layers.0.weight", (64,))
    g = torch.Generator().manual_seed(0)
    c = mapping["diffusion_model_input_blocks_1_1_proj_in"][1][0]
    sd = {"lora_unet_down_blocks_0_attentions_0_proj_in.lora_up.weight": torch.randn(c, 4, 1, 1, generator=g),
          "lora_unet_down_blocks_0_attentions_0_proj_in.lora_down.weight": torch.randn(4, c, 1, 1, generator=g),
          "lora_unet_down_blocks_0_attentions_0_proj_in.alpha": torch.tensor(2.0),
          "lora_unet_input_blocks_1_1_proj_out.lora_B.weight": torch.randn(c, 4, generator=g),        # compvis-named + A/B naming
          "lora_unet_input_blocks_1_1_proj_out.lora_A.weight": torch.randn(4, c, generator=g),
          "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight": torch.randn(8, 2, generator=g)}
    net = nets.load_network("t", sd, m)
    assert set(net.modules) == {"diffusion_model_input_blocks_1_1_proj_in", "diffusion_model_input_blocks_1_1_proj_out"}
    mod = net.modules["diffusion_model_input_blocks_1_1_proj_in"]
    assert mod.dim == 4 and mod.calc_scale() == 0.5 and mod.engine_key == "input_blocks.1.1.proj_in.weight"
    assert net.modules["diffusion_model_input_blocks_1_1_proj_out"].calc_scale() == 1.0
    assert list(net.keys_failed_to_match) == ["lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight"]
    bad = dict(sd)
    bad["lora_unet_down_blocks_0_attentions_0_proj_in.lora_up.weight"] = torch.randn(c + 8, 4, 1, 1, generator=g)
    with pytest.raises(AssertionError):
        nets.load_network("bad", bad, m)


def test_shard_range_partitions_exactly():
    par = sub("parallel")
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [par.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_geglu_bias_packing_matches_kernel_row_order():
    ops = sub("ops")
    o = 256
    b = torch.arange(o, dtype=torch.float32)
    packed = ops.pack_bias(b, o, geglu=True)
    # packed row 64g + r (r < 32) holds value channel 32g + r; row 64g + 32 + r holds gate channel o/2 + 32g + r
    for g in range(o // 64):
        assert packed[64 * g: 64 * g + 32].tolist() == list(range(32 * g, 32 * g + 32))
        assert packed[64 * g + 32: 64 * g + 64].tolist() == list(range(o // 2 + 32 * g, o // 2 + 32 * g + 32))


WORKER = textwrap.dedent("""
    import importlib, os, sys, torch
    sys.path.insert(0, {root!r})
    par = importlib.import_module("stable-diffusion-webui_amd.parallel")
    rank, local_rank, world = par.init_distributed("gloo")
    g = torch.Generator().manual_seed(5)
    sd = None
    if rank == 0:
        sd = {{"a.weight": torch.randn(300, 700, generator=g).half(), "b.bias": torch.randn(1000, generator=g),
              "alphas_cumprod": torch.rand(1000, generator=g)}}
    for algo in ("scatter_allgather", "broadcast"):
        out = par.broadcast_state_dict(sd, src=0, device="cpu", algo=algo)
        g2 = torch.Generator().manual_seed(5)
        ref = {{"a.weight": torch.randn(300, 700, generator=g2).half(), "b.bias": torch.randn(1000, generator=g2),
               "alphas_cumprod": torch.rand(1000, generator=g2)}}
        for k in ref:
            assert out[k].dtype == ref[k].dtype and torch.equal(out[k], ref[k]), (algo, k)
    n = 5
    lo, hi = par.shard_range(n, world, rank)
    mine = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
    counts = [par.shard_range(n, world, r)[1] - par.shard_range(n, world, r)[0] for r in range(world)]
    allv = par.gather_to_rank0(mine, counts)
    if rank == 0:
        assert allv[:, 0].tolist() == [0., 1., 2., 3., 4.]
    assert par.max_over_ranks(float(rank)) == float(world - 1)
    par.barrier()
    print("RANK_OK", rank)
""")


def test_weight_broadcast_and_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    import socket
    last = ""
    for attempt in range(3):                                  # a rendezvous port can be taken between probing and use: retry
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
        procs = []
        for r in range(2):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
            outs.append(out.decode())
        if all(p.returncode == 0 and f"RANK_OK {r}" in o for r, (p, o) in enumerate(zip(procs, outs))):
            return
        last = "\n".join(outs)
        if "AssertionError" in last:                          # a real mismatch, not a rendezvous problem
            break
    raise AssertionError(last)


SHARD_WORKER = textwrap.dedent("""
    import importlib, os, sys
    import numpy as np
    import torch
    sys.path.insert(0, {root!r})
    par = importlib.import_module("stable-diffusion-webui_amd.parallel")
    processing = importlib.import_module("stable-diffusion-webui_amd.processing")
    rank, local_rank, world = par.init_distributed("gloo")
    assert world == 2

    calls = []
    def fake_runner(p):
        # stands in for process_images (no GPU here): one deterministic "image" per (seed, cond row), batch by batch
        n = p.batch_size * p.n_iter
        seeds = list(p.seed) if isinstance(p.seed, (list, tuple)) else [p.seed + i for i in range(n)]
        assert p.c.shape[0] == n and p.uc.shape[0] == n and len(seeds) == n
        imgs = []
        for it in range(p.n_iter):
            calls.append(p.batch_size)
            for i in range(it * p.batch_size, (it + 1) * p.batch_size):
                v = (seeds[i] * 7 + int(p.c[i].sum().item()) * 3 + int(p.uc[i].sum().item())) % 251
                f = 2 if getattr(p, "enable_hr", False) else 1      # a hires job's final images are not p.height x p.width
                imgs.append(np.full((p.height * f, p.width * f, 3), v, dtype=np.uint8))
        return processing.Processed(p, imgs, seeds[0], seeds, None)

    def job(bs, n_iter, hr=False):
        n = bs * n_iter
        c = torch.arange(n, dtype=torch.float32)[:, None, None].repeat(1, 3, 2)
        return processing.StableDiffusionProcessingTxt2Img(sd_model=None, c=c, uc=c * 2 + 1, seed=4242, batch_size=bs, n_iter=n_iter,
                                                           steps=2, width=8, height=8, sampler_name="Euler a", enable_hr=hr)
    # hires job with fewer images than ranks: the empty rank's gather buffer must have the FINAL image size — the one case that costs
    # a second collective (the size from rank 0); a job with an image on every rank is ONE gather and nothing else
    res = par.process_images_sharded(job(1, 1, hr=True), runner=fake_runner)
    assert par.COLLECTIVES["job"] == 2
    if rank == 0:
        assert len(res.images) == 1 and res.images[0].shape == (16, 16, 3)
    for bs, n_iter in ((2, 2), (2, 3), (1, 1), (4, 1)):
        calls.clear()
        whole = fake_runner(job(bs, n_iter))
        calls.clear()
        before = par.COLLECTIVES["job"]
        res = par.process_images_sharded(job(bs, n_iter), runner=fake_runner)
        assert par.COLLECTIVES["job"] - before == (1 if bs * n_iter >= world else 2), (bs, n_iter)
        lo, hi = par.shard_range(bs * n_iter, world, rank)
        assert sum(calls) == hi - lo and all(c <= bs for c in calls), (calls, lo, hi)
        if rank == 0:
            assert res.shard == (0, bs * n_iter) and len(res.images) == bs * n_iter
            assert all(np.array_equal(a, b) for a, b in zip(res.images, whole.images)), (bs, n_iter)
            assert res.all_seeds == whole.all_seeds
        # replaying one rank's slice in a single process gives that rank's images
        solo = par.process_images_sharded(job(bs, n_iter), runner=fake_runner, world=2, rank=1)
        lo1, hi1 = par.shard_range(bs * n_iter, 2, 1)
        assert all(np.array_equal(a, b) for a, b in zip(solo.images, whole.images[lo1:hi1]))
    par.barrier()
    print("RANK_OK", rank)
""")


def _run_world2(script):
    import socket
    last = ""
    for attempt in range(3):                                  # a rendezvous port can be taken between probing and use: retry
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
        procs = []
        for r in range(2):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
            outs.append(out.decode())
        if all(p.returncode == 0 and f"RANK_OK {r}" in o for r, (p, o) in enumerate(zip(procs, outs))):
            return
        last = "\n".join(outs)
        if "AssertionError" in last:                          # a real mismatch, not a rendezvous problem
            break
    raise AssertionError(last)


def test_weight_blob_collective_is_chosen_up_front_from_the_backend():
    """broadcast_blob must not pick its fallback by catching an exception on some ranks.  The choice is a pure function of
    (backend, size, request), evaluated identically on every rank."""
    par = sub("parallel")
    assert par.blob_algorithm("nccl", 2 << 30) == "scatter_allgather" and par.blob_algorithm("gloo", 2 << 20) == "scatter_allgather"
    assert par.blob_algorithm("mpi", 2 << 30) == "broadcast" and par.blob_algorithm("ucc", 2 << 30) == "broadcast"
    assert par.blob_algorithm("nccl", 1000) == "broadcast" and par.blob_algorithm("nccl", 2 << 30, "broadcast") == "broadcast"
    import inspect
    assert "except" not in inspect.getsource(par.broadcast_blob)


def test_process_images_sharded_world_size_2_gloo(tmp_path):
    """the job [0, batch_size * n_iter) split contiguously over 2 ranks (incl. ragged and empty shards),
    global seeds kept, per-call batch size never above p.batch_size, uint8 images gathered on rank 0 in job order."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER.format(root=ROOT))
    _run_world2(script)


def test_bench_refuses_to_claim_gpus_it_does_not_have():
    """bench.py --gpus N outside torchrun spawns the N ranks itself; with fewer than N devices visible it must fail, not print
    n_gpus: N for a dp1 run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and b"refusing" in r.stderr and b"n_gpus" not in r.stdout


def test_bench_cpu_baseline_is_bounded_and_isolated(monkeypatch):
    """bench.py's CPU baseline runs in a child process under a wall limit, with as many threads as the process may really use: a host
    that reports 256 hardware threads but grants 16 (a GPU box of round 2) hung the all-host-threads version for the whole bench
    timeout.  (a) the thread count honours the affinity mask; (b) the child produces the baseline object (tiny model); (c) a child that
    exceeds the wall limit costs that limit and yields a null value instead of losing the bench line."""
    import json
    import time
    monkeypatch.syspath_prepend(ROOT)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--model", "tiny", "--size", "128", "--sampler-steps", "3", "--cpu-baseline-budget", "20"])
    bench = importlib.import_module("bench")
    assert 1 <= bench.usable_cpus() <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert bench.usable_cpus() <= len(os.sched_getaffinity(0))
    args = bench.parse()
    out = bench.cpu_baseline_isolated(args)
    assert out["kind"] == "port" and out["unit"] == "images/s" and out["value"] and out["value"] > 0, out
    assert out["cores"] == bench.usable_cpus() and "usable" in out["sample"]
    json.dumps(out)
    args.cpu_baseline_timeout = 0.05
    t0 = time.time()
    late = bench.cpu_baseline_isolated(args)
    assert late["value"] is None and "wall limit" in late["sample"] and time.time() - t0 < 30


_ASM_CACHE = {}


def _gfx950_assembly(name, extra_flags=()):
    """csrc/<name>.hip cross-compiled to gfx950 assembly; extra_flags: what csrc/build.sh adds for that file.  csrc/build.sh leaves the
    device assembly of the library's own compile in csrc/build/asm/<name>.s (-save-temps=obj: same flags, same code object): it is
    used when it is newer than every source of csrc/, so that the suite does not repeat the 5-minute compile of gemm.hip; otherwise the
    file is compiled here (once per test session) and left there for the next one."""
    import shutil
    import tempfile
    if name not in _ASM_CACHE:
        csrc = os.path.join(ROOT, "stable-diffusion-webui_amd", "csrc")
        kept = os.path.join(csrc, "build", "asm", name + ".s")
        sources = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".inc", ".cpp", ".sh"))]
        if os.path.exists(kept) and os.path.getsize(kept) > 0 and os.path.getmtime(kept) >= max(os.path.getmtime(f) for f in sources):
            _ASM_CACHE[name] = open(kept).read()
            return _ASM_CACHE[name]
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("hipcc not available")
        out = os.path.join(tempfile.mkdtemp(prefix="sdmi_asm_"), name + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *extra_flags, "-S", "--cuda-device-only", "-o", out,
                        os.path.join(csrc, name + ".hip")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        _ASM_CACHE[name] = open(out).read()
        try:
            os.makedirs(os.path.dirname(kept), exist_ok=True)
            shutil.move(out, kept)
        except OSError:
            pass
    return _ASM_CACHE[name]


def test_the_k_loops_of_the_4_wave_gemm_tiles_stay_lean():
    """Round 5's finding kept as a rule.  The 4-wave tiles were bound by ISSUE, not by latency or bandwidth: the general gather rebuilt
    every LDS-direct load's address per K step — 315 instructions around the 16 MFMAs of a 128x64 tile (2.5 workgroups per CU x 20 steps x
    ~1400 issue cycles = the 30 us every tile, occupancy and prefetch depth had measured on M4096 N1280 K1280).  The running-pointer walks
    (gemm.hip LIN for 1x1 launches, LIN3 for the plain 3x3 convs) must keep their K loops at a fraction of that; counted over the basic
    blocks hipcc marks as loop bodies."""
    import re
    text = _gfx950_assembly("gemm")

    def loop_instructions(kname):
        body = re.search(r"^%s:[^\n]*\n(.*?)\.Lfunc_end" % re.escape(kname), text, re.S | re.M).group(1)
        n = mfma = 0
        inside = head = False
        for line in body.split("\n"):
            t = line.strip()
            if not t:
                continue
            if t.startswith(".LBB") or t.startswith("; %bb."):
                inside = "in Loop" in t or "Loop Header" in t
                head = True
                continue
            if head and t.startswith(";") and ("in Loop" in t or "Loop Header" in t):
                inside = True             # (-save-temps keeps the IR block name on the label line; the loop note follows on its own line)
                continue
            if t.startswith((";", ".")):
                continue
            head = False
            if inside:
                n += 1
                mfma += t.startswith("v_mfma")
        return n, mfma
    # tile -> (MFMAs per K step and wave, cap of the linear walk, cap of the lean 3x3 walk)       measured: 83 / 134, 113 / 166, 65 / 103
    for tile, (per_step, cap_lin, cap_lin3) in {"Li128ELi64ELi4ELi1ELi64E": (16, 100, 160), "Li128ELi128ELi2ELi2ELi64E": (32, 135, 200),
                                                "Li64ELi64ELi4ELi1ELi64E": (8, 80, 125)}.items():
        base = "_ZN4sdmi16gemm_mfma_kernelI" + tile + "Lb1ELb0ELb0ELb0ELb0ELi0ELi2E"
        general, m0 = loop_instructions(base + "Lb0ELb0EEEvNS_5GemmPE")
        lin, m1 = loop_instructions(base + "Lb1ELb0EEEvNS_5GemmPE")
        lin3, m2 = loop_instructions(base + "Lb0ELb1EEEvNS_5GemmPE")
        assert m0 == m1 == m2 == per_step, (tile, m0, m1, m2)                  # the loop found IS the K loop, not unrolled
        assert lin <= cap_lin and lin3 <= cap_lin3, (tile, lin, lin3)
        assert general >= 1.4 * lin3 and general >= 2 * lin, (tile, general, lin, lin3)


def test_kernels_compile_without_scratch_or_spills():
    """Code-object metadata of every kernel in csrc/*.hip: no private (scratch) segment, no VGPR / SGPR spills — a spill in the
    GEMM or attention loops costs more than any tuning gained (the first attention rewrite went to scratch through captured
    uint4 arrays) — and the register budgets the occupancy assumptions rest on (the 8-wave ping-pong GEMM runs two waves per
    SIMD: <= 256 of the 512 unified VGPRs; the HBM-bound norm kernels stay <= 128 so several waves per SIMD hide latency)."""
    import re
    seen = {}
    for name in ("gemm", "attention", "norm", "elementwise"):
        text = _gfx950_assembly(name)
        meta = text[text.index("amdhsa.kernels:"):]
        for block in meta.split("  - .agpr_count:")[1:]:
            kname = re.search(r"\.name:\s+(\S+)", block).group(1)
            field = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, block).group(1))
            # not on the product path: the s_memtime-instrumented ping-pong instantiation (TIMING = true, tools/gemm_sections.py)
            # and the register-staged two-stage GEMM (GLDS = false; kept as the bitwise cross-check of the LDS-direct path)
            if re.search(r"gemm_mfma_pingpong_kernelILi\d+ELi\d+ELb[01]ELb1E", kname) or re.search(r"gemm_mfma_kernelI(Li\d+E){5}Lb0E", kname):
                continue
            seen[kname] = field("vgpr_count")
            assert field("private_segment_fixed_size") == 0, f"{kname} uses scratch"
            # (SGPR spills go to VGPR lanes, not memory; tolerated only in the one-thread-per-output cross-check kernel, whose
            # kernel-argument struct alone crowds the scalar file)
            # ... and in the GroupNorm-statistics instantiations of the ping-pong GEMM (last template flag), where a handful of
            # kernel-argument SGPRs are parked in VGPR lanes BEFORE the K loop and read back after it — checked below)
            stats_pp = re.search(r"gemm_mfma_pingpong_kernelILi\d+ELi\d+E(Lb[01]E){4}Lb1ELi0EEEv", kname) is not None
            # ... the same epilogue on the row-shared 3x3 walk (gemm_mfma_pingpong_dx_kernel<BM, BN, STATS = true>)
            stats_dx = re.search(r"gemm_mfma_pingpong_dx_kernelILi\d+ELi\d+ELb1EEEv", kname) is not None
            # ... and in the LayerNorm consumer / producer forms (last template argument 1 / 2), same rule: parked before the K loop
            ln_pp = re.search(r"gemm_mfma_pingpong_kernelILi\d+ELi\d+E(Lb[01]E){5}Li[12]EEEv", kname) is not None
            # ... and, since round 5 (the (hi, lo) stream branches of the shared epilogue: EP_HILO), a few kernel-argument SGPRs of the other
            # ping-pong instantiations too — same rule: parked before the K loop, read back behind it, never inside
            any_pp = "gemm_mfma_pingpong" in kname
            # ... and the 4-wave kernel's lean 3x3 walk (last two template flags false, true), plain and GroupNorm-statistics forms: the same
            # few kernel-argument SGPRs, parked ahead of the first MFMA and read back behind the last
            stats_lin3 = re.search(r"gemm_mfma_kernelI(Li\d+E){5}Lb1E(Lb[01]E){4}Li0ELi2ELb0ELb1EEEv", kname) is not None
            assert field("vgpr_spill_count") == 0, f"{kname} spills registers"
            # ... and, since round 6 (gemm_epilogue_hilo: a second, block-uniform region of the epilogue behind the EP_HILO flag), the other
            # 4-wave MFMA instantiations as well — the same few kernel-argument SGPRs, and the same rule checked below: none of that traffic
            # between the first and the last MFMA
            any_4w = "gemm_mfma_kernel" in kname
            assert field("sgpr_spill_count") == 0 or "generic" in kname or (stats_pp and field("sgpr_spill_count") <= 12) or (ln_pp and field("sgpr_spill_count") <= 24) or \
                (stats_dx and field("sgpr_spill_count") <= 12) or (any_pp and field("sgpr_spill_count") <= 12) or (any_4w and field("sgpr_spill_count") <= 12), \
                f"{kname} spills registers"
            if (stats_pp or ln_pp or stats_dx or any_pp) and field("sgpr_spill_count"):
                body = re.search(r"^%s:[^\n]*\n(.*?)\.Lfunc_end" % re.escape(kname), text, re.S | re.M).group(1)
                loop = body[body.index("s_setprio 1"):body.rindex("s_setprio 0")]
                assert "v_readlane" not in loop and "v_writelane" not in loop, f"{kname}: SGPR spill traffic inside the K loop"
            if (stats_lin3 or any_4w) and field("sgpr_spill_count"):
                body = re.search(r"^%s:[^\n]*\n(.*?)\.Lfunc_end" % re.escape(kname), text, re.S | re.M).group(1)
                loop = body[body.index("v_mfma_f32_16x16x32_f16"):body.rindex("v_mfma_f32_16x16x32_f16")]
                assert "v_readlane" not in loop and "v_writelane" not in loop, f"{kname}: SGPR spill traffic inside the K loop"
            assert field("vgpr_count") <= (256 if "pingpong" in kname else 512), kname     # 8-wave workgroups: 2 waves per SIMD
            # (gn_fused_small_kernel is the opposite design on purpose: one (image, group) slice held entirely in registers, every load
            # of a thread issued up front — its latency hiding is the 4 .. 24 loads in flight per thread, not the wave count)
            if ("gn_" in kname and "gn_fused_small" not in kname) or "layernorm" in kname:
                # (the (hi, lo) input forms of the accuracy mode — last template argument true — hold two fp16 vectors per value: <= 192)
                hilo = re.search(r"(layernorm_kernelILi\d+ELi\d+ELb1E|gn_(stats|apply)_kernelILb1E)", kname) is not None
                assert field("vgpr_count") <= (192 if hilo else 128), (kname, field("vgpr_count"))
    assert any("gemm_mfma_pingpong_kernel" in k for k in seen) and any("attn_mfma_kernel" in k for k in seen) and len(seen) > 40
    # rowchain.hip (round 5): one wave per SIMD by design (row fragments + O^T accumulators: up to 512 registers); the prologue / epilogue
    # may park registers, the chunk loops — the basic blocks that carry the MFMAs — must not touch scratch in the default (feed-forward)
    # chain, and stay within a handful of dword reloads per head pair in the opt-in cross-attention chain
    text = _gfx950_assembly("rowchain", extra_flags=("-fno-honor-nans",))
    for kname, cap in (("rowchain_ff_kernel", 0),):
        m = re.search(r"^(_ZN4sdmi\d+%s\w+):[^\n]*\n(.*?)\.Lfunc_end" % kname, text, re.S | re.M)
        assert m, kname
        blocks = re.split(r"\n\.LBB\d+_\d+:", m.group(2))
        loops = [b for b in blocks if b.count("v_mfma_f32_32x32x16_f16") >= 100]
        assert loops, f"{kname}: no chunk loop found"
        for b in loops:
            assert b.count("scratch_") <= cap, f"{kname}: {b.count('scratch_')} scratch accesses inside a chunk loop"
        meta = text[text.index("amdhsa.kernels:"):]
        blk = next(bk for bk in meta.split("  - .agpr_count:")[1:] if kname in bk)
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 512


def test_pingpong_gemm_isa_keeps_counted_waits(tmp_path):
    """The ping-pong GEMM's pipelining rests on COUNTED `s_waitcnt vmcnt(N)` (256-row tiles: N1 = 4 + BN/64, N2 = N1 + BN/128;
    128-row tile: N3 = BN/64 + 2) and raw
    s_barrier / s_setprio in its K loop; a compiler that folded them into vmcnt(0) would silently serialise the loads.
    Cross-compile the kernel to gfx950 assembly and check the loop body."""
    import re
    text = _gfx950_assembly("gemm")
    for bm, bn, waits, phases, kord, st in [(bm, bn, w, ph, k, st) for (bm, bn, w, ph) in ((256, 256, (8, 10), 4), (256, 320, (9, 11), 4), (128, 320, (7,), 2))
                                            for k in (0, 1) for st in (0, 1)]:  # k = 1: the channel-block-major instantiation the 3x3 convs run;
                                                                                # st = 1: the GroupNorm-statistics epilogue variant
        m = re.search(r"^_ZN4sdmi25gemm_mfma_pingpong_kernelILi%dELi%dELb0ELb0ELb0ELb%dELb%dELi0EEEvNS_5GemmPE:[^\n]*\n(.*?)\.Lfunc_end" % (bm, bn, kord, st), text, re.S | re.M)
        assert m, f"ping-pong kernel <{bm},{bn}> not found in the assembly"
        body = m.group(1)
        first, last = body.index("s_setprio 1"), body.rindex("s_setprio 0")
        loop = body[first:last]
        assert loop.count("s_setprio 1") == phases and loop.count("v_mfma_f32_16x16x32_f16") == phases * 4 * (bn // 64)
        for n in waits:
            assert f"s_waitcnt vmcnt({n})" in loop, (bm, bn, n)
        assert loop.count("s_barrier") == 2 * (phases - 1)   # (barrier b, barrier a) pairs between the first and last MFMA sections
        assert "scratch_" not in loop, "register spill inside the K loop (scratch traffic would disturb the vmcnt accounting)"


def test_row_shared_conv_kernel_isa_is_one_loop_with_counted_waits():
    """gemm_mfma_pingpong_dx_kernel (the default 3x3 walk): its K loop must stay ONE copy of four (two) phases whose only waits are the
    counted ones of the header — BU, BU + 2, BU + 4 (PH = 2: BU, BU + 1, BU + 2) and the tail's 0.  Two round-4 variants compiled into
    something else and lost 30-40 % without a single spill being reported: dx as a compile-time constant tripled the body (accumulators
    renamed across it), a second group order behind a runtime flag made the compiler clone the loop.  Checked on the cross-compiled ISA:
    MFMA count of the loop = one K tile, the counted waits present, no scratch, no lane spill traffic."""
    import re
    text = _gfx950_assembly("gemm")
    for bm, bn, phases, st in [(256, 320, 4, 0), (256, 320, 4, 1), (256, 256, 4, 0), (128, 320, 2, 0), (128, 320, 2, 1)]:
        m = re.search(r"^_ZN4sdmi28gemm_mfma_pingpong_dx_kernelILi%dELi%dELb%dEEEvNS_5GemmPE:[^\n]*\n(.*?)\.Lfunc_end" % (bm, bn, st), text, re.S | re.M)
        assert m, f"row-shared kernel <{bm},{bn},{st}> not found in the assembly"
        body = m.group(1)
        loop = body[body.index("s_setprio 1"):body.rindex("s_setprio 0")]
        bu, na = bn // 64, 4 if phases == 4 else 2
        assert loop.count("s_setprio 1") == phases and loop.count("v_mfma_f32_16x16x32_f16") == phases * 4 * (bn // 64), (bm, bn, st)
        for n in (bu, bu + na // 2, bu + na):
            assert f"s_waitcnt vmcnt({n})" in loop, (bm, bn, st, n)
        assert set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)) <= {str(v) for v in (0, bu, bu + na // 2, bu + na)}, (bm, bn, st)
        assert "scratch_" not in loop and "v_readlane" not in loop and "v_writelane" not in loop, (bm, bn, st)
        assert body.count("v_mfma_f32_16x16x32_f16") == phases * 4 * (bn // 64), "the K loop was cloned / unrolled"


def test_oracle_pipeline_batch_invariance():
    """Image i of a batch equals that image generated alone (per-image generators, modules/rng.py:108)."""
    schema = sub("schema")
    from oracle import pipeline as opipe, unet as ou, vae as ov
    sd = schema.synthetic_state_dict(schema.tiny_unet(), None, dtype=torch.float32)
    om = opipe.OracleModel(sd, ou.tiny_config(), None)
    g = torch.Generator().manual_seed(3)
    cond, uncond = torch.randn(2, 77, 64, generator=g), torch.randn(2, 77, 64, generator=g)
    both = opipe.sample(om, cond, uncond, [1000, 1001], 3, "euler_a", 7.0, (8, 8))
    solo = opipe.sample(om, cond[1:], uncond[1:], [1001], 3, "euler_a", 7.0, (8, 8))
    assert float((both[1] - solo[0]).abs().max()) < 1e-4 * float(both.abs().max())


# ------------------------------------------------------------------------------------------------------------
# checkpoint loader (row a15) and extension packaging (row B0) — host logic, no GPU
# ------------------------------------------------------------------------------------------------------------
def test_read_state_dict_safetensors_round_trip_and_key_fixups(tmp_path):
    """modules/sd_models.py:262-281, 312-329: a pytorch-lightning style checkpoint with the OLD CLIP key layout, written as
    .safetensors and as .ckpt, comes back with cond_stage_model.transformer.text_model.* keys and every tensor intact."""
    import safetensors.torch
    sd_models, schema = sub("sd_models"), sub("schema")
    sd = schema.synthetic_state_dict(schema.tiny_unet(), schema.tiny_vae(), dtype=torch.float16)
    old = dict(sd)
    g = torch.Generator().manual_seed(3)
    old["cond_stage_model.transformer.embeddings.position_embedding.weight"] = torch.randn(77, 64, generator=g)
    old["cond_stage_model.transformer.encoder.layers.0.mlp.fc1.weight"] = torch.randn(8, 64, generator=g).half()
    old["cond_stage_model.transformer.final_layer_norm.bias"] = torch.randn(64, generator=g)
    st = tmp_path / "model.safetensors"
    safetensors.torch.save_file({k: v.contiguous() for k, v in old.items()}, str(st))
    ck = tmp_path / "model.ckpt"
    torch.save({"state_dict": old, "global_step": 7}, str(ck))
    for path in (st, ck):
        got = sd_models.read_state_dict(str(path))
        assert "state_dict" not in got
        assert "cond_stage_model.transformer.text_model.embeddings.position_embedding.weight" in got
        assert "cond_stage_model.transformer.text_model.encoder.layers.0.mlp.fc1.weight" in got
        assert "cond_stage_model.transformer.text_model.final_layer_norm.bias" in got
        assert not any(k.startswith("cond_stage_model.transformer.embeddings.") for k in got)
        for k, v in sd.items():
            assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    # the mmap-less variant (opts.disable_mmap_load_safetensors) reads the same bytes
    shared = sub("shared")
    shared.opts.disable_mmap_load_safetensors = True
    try:
        got2 = sd_models.read_state_dict(str(st))
    finally:
        shared.opts.disable_mmap_load_safetensors = False
    assert all(torch.equal(got2[k], v) for k, v in sd.items())
    # SD 2.1 Turbo (SGM layout): conditioner.embedders.0.* -> cond_stage_model.*
    turbo = {"conditioner.embedders.0.model.ln_final.weight": torch.ones(1024), "conditioner.embedders.0.model.x": torch.zeros(2)}
    out = sd_models.get_state_dict_from_checkpoint(dict(turbo))
    assert set(out) == {"cond_stage_model.model.ln_final.weight", "cond_stage_model.model.x"}
    # standalone VAE file: loss / EMA bookkeeping keys dropped (modules/sd_vae.py:188-191)
    vae_only = {k[len(schema.VAE_PREFIX):]: v for k, v in sd.items() if k.startswith(schema.VAE_PREFIX)}
    vae_only["loss.logvar"] = torch.zeros(1)
    vae_only["model_ema.decay"] = torch.zeros(1)
    vp = tmp_path / "ext.vae.safetensors"
    safetensors.torch.save_file({k: v.contiguous() for k, v in vae_only.items()}, str(vp))
    vd = sd_models.load_vae_dict(str(vp))
    assert "loss.logvar" not in vd and "model_ema.decay" not in vd and "decoder.conv_in.weight" in vd
    assert sd_models.guess_unet_config(got).model_channels == schema.sd15_unet().model_channels


def test_extension_script_registers_three_callbacks_against_stubbed_webui(monkeypatch):
    """Boundary B0: extension/scripts/mi355x_engine.py imported exactly as the webui's script loader would (modules.* resolved to
    the webui's modules — stubbed here), must register on_list_unets / on_list_optimizers / on_model_loaded, and the registered
    callables must produce an SdUnetOption per checkpoint and the SdOptimization row."""
    import types
    reg = {"unets": [], "optimizers": [], "model_loaded": []}
    cb = types.ModuleType("modules.script_callbacks")
    cb.on_list_unets = lambda f: reg["unets"].append(f)
    cb.on_list_optimizers = lambda f: reg["optimizers"].append(f)
    cb.on_model_loaded = lambda f: reg["model_loaded"].append(f)
    sdm = types.ModuleType("modules.sd_models")
    info = types.SimpleNamespace(filename="/models/Stable-diffusion/v1-5.safetensors", model_name="v1-5")
    sdm.checkpoints_list = {"v1-5.safetensors [abc]": info}
    reads = []
    sdm.read_state_dict = lambda fn, map_location=None: reads.append((fn, map_location)) or {"k": 1}
    shared_stub = types.ModuleType("modules.shared")
    root = types.ModuleType("modules")
    root.script_callbacks, root.sd_models, root.shared = cb, sdm, shared_stub
    for name, mod in (("modules", root), ("modules.script_callbacks", cb), ("modules.sd_models", sdm), ("modules.shared", shared_stub)):
        monkeypatch.setitem(sys.modules, name, mod)
    import importlib.util
    path = os.path.join(ROOT, "stable-diffusion-webui_amd", "extension", "scripts", "mi355x_engine.py")
    spec = importlib.util.spec_from_file_location("mi355x_engine_script", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert len(reg["unets"]) == 1 and len(reg["optimizers"]) == 1 and len(reg["model_loaded"]) == 1
    options = []
    reg["unets"][0](options)
    sd_unet = sub("sd_unet")
    assert len(options) == 1 and isinstance(options[0], sd_unet.SdUnetOption) and options[0].model_name == "v1-5"
    assert options[0].label == "[MI355X] v1-5"
    assert options[0]._provider() == {"k": 1} and reads == [("/models/Stable-diffusion/v1-5.safetensors", "cpu")]   # lazily, on activation
    opts = []
    reg["optimizers"][0](opts)
    assert len(opts) == 1 and opts[0].name == "mi355x" and opts[0].cmd_opt == "opt_mi355x_attention" if hasattr(opts[0], "cmd_opt") else True
    # importing the script twice (webui "Reload UI" clears callbacks and re-imports) registers again without side effects
    mod2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod2)
    assert len(reg["unets"]) == 2
    assert mod.registered == {}                               # this stub webui has no sd_samplers / Lora modules: the script still loads


def test_extension_script_registers_samplers_lora_and_clip_hooks_against_stubbed_webui(monkeypatch):
    """Boundaries B3 / B5 / B6 as the shipped script installs them: the rows of modules.sd_samplers.all_samplers keep their names,
    aliases and options but dispatch to the engine sampler while the engine UNet is sd_unet.current_unet (and to the stock constructor
    otherwise); networks.load_networks of the built-in Lora extension is wrapped (stock call first, engine merge only with an active
    engine UNet); a second import of the script does not stack wrappers."""
    import collections
    import types
    amd = sub("sd_samplers")
    SamplerData = collections.namedtuple("SamplerData", ["name", "constructor", "aliases", "options"])
    stock_calls = []
    mk = lambda name: (lambda model: stock_calls.append((name, model)) or types.SimpleNamespace(stock=name))
    samplers_mod = types.ModuleType("modules.sd_samplers")
    samplers_mod.all_samplers = [SamplerData("Euler a", mk("Euler a"), ["k_euler_a"], {"uses_ensd": True}),
                                 SamplerData("DPM++ 2M", mk("DPM++ 2M"), ["k_dpmpp_2m"], {"scheduler": "karras"}),
                                 SamplerData("Some third-party sampler", mk("x"), [], {})]
    samplers_mod.all_samplers_map = {x.name: x for x in samplers_mod.all_samplers}
    set_calls = []
    samplers_mod.set_samplers = lambda: set_calls.append(len(samplers_mod.all_samplers))
    unet_mod = types.ModuleType("modules.sd_unet")
    unet_mod.current_unet = None
    cb = types.ModuleType("modules.script_callbacks")
    cb.on_list_unets = cb.on_list_optimizers = cb.on_model_loaded = lambda f: None
    sdm = types.ModuleType("modules.sd_models")
    sdm.checkpoints_list = {}
    reads = []
    sdm.read_state_dict = lambda fn, map_location=None: reads.append(fn) or {}
    shared_stub = types.ModuleType("modules.shared")
    shared_stub.sd_model = types.SimpleNamespace(alphas_cumprod=torch.linspace(0.99, 0.01, 1000))
    lora = types.ModuleType("networks")
    lora_calls = []
    lora.load_networks = lambda names, te=None, un=None, dyn=None: lora_calls.append((tuple(names), te, un, dyn))
    lora.available_network_aliases = {"myLora": types.SimpleNamespace(filename="/models/Lora/myLora.safetensors")}
    lora.available_networks = {}
    root = types.ModuleType("modules")
    root.script_callbacks, root.sd_models, root.shared, root.sd_samplers, root.sd_unet = cb, sdm, shared_stub, samplers_mod, unet_mod
    for name, m in (("modules", root), ("modules.script_callbacks", cb), ("modules.sd_models", sdm), ("modules.shared", shared_stub),
                    ("modules.sd_samplers", samplers_mod), ("modules.sd_unet", unet_mod), ("networks", lora)):
        monkeypatch.setitem(sys.modules, name, m)
    import importlib.util
    path = os.path.join(ROOT, "stable-diffusion-webui_amd", "extension", "scripts", "mi355x_engine.py")
    spec = importlib.util.spec_from_file_location("mi355x_engine_script_b", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.registered == {"samplers": ["Euler a", "DPM++ 2M"], "lora": True} and set_calls == [3]
    rows = samplers_mod.all_samplers
    assert [r.name for r in rows] == ["Euler a", "DPM++ 2M", "Some third-party sampler"]
    assert rows[0].aliases == ["k_euler_a"] and rows[1].options == {"scheduler": "karras"} and samplers_mod.all_samplers_map["Euler a"] is rows[0]
    # no engine UNet active: the stock constructor
    assert rows[0].constructor("torch-model").stock == "Euler a" and stock_calls == [("Euler a", "torch-model")]
    # engine UNet active: the engine sampler over a view of (webui model, engine)
    fake_engine = types.SimpleNamespace(device=0)
    unet = sub("sd_unet").Mi355xUnet(lambda: {}, unet_cfg=sub("schema").tiny_unet())
    unet.engine, unet._sd = fake_engine, {"model.diffusion_model.x.weight": torch.ones(2)}
    unet_mod.current_unet = unet
    s = rows[1].constructor(shared_stub.sd_model)
    assert isinstance(s, amd.KDiffusionSampler) and s.sd_model.engine is fake_engine and len(stock_calls) == 1
    assert torch.equal(s.sd_model.alphas_cumprod, shared_stub.sd_model.alphas_cumprod)
    assert torch.equal(s.sd_model.unet_checkpoint_tensor("x.weight"), torch.ones(2))
    # Lora: stock first; the engine merge reads the same file through the webui's own reader
    amd_net = sub("networks")
    merged = []
    monkeypatch.setattr(amd_net, "load_networks", lambda view, names, sds, te, un, dyn: merged.append((view.engine, names, te, un, dyn)))
    lora.load_networks(["myLora"], [0.5], [0.8], [None])
    assert lora_calls == [(("myLora",), [0.5], [0.8], [None])] and reads == ["/models/Lora/myLora.safetensors"]
    assert merged == [(fake_engine, ["myLora"], [0.5], [0.8], [None])]
    # the same request again (ExtraNetworkLora.activate runs per job): stock bookkeeping, no re-read, no re-merge; new multipliers merge
    # again from the cached file; a name that collides with an alias resolves by file name only (networks.py:303)
    lora.load_networks(["myLora"], [0.5], [0.8], [None])
    assert len(lora_calls) == 2 and len(reads) == 1 and len(merged) == 1
    lora.load_networks(["myLora"], [0.5], [0.3], [None])
    assert len(reads) == 1 and merged[-1] == (fake_engine, ["myLora"], [0.5], [0.3], [None])
    lora.forbidden_network_aliases = {"mylora": 1}
    lora.load_networks(["myLora"], [0.5], [0.3], [None])
    assert len(merged) == 3 and merged[-1][1] == []           # not in available_networks under that name: nothing of it is merged
    del lora.forbidden_network_aliases
    unet_mod.current_unet = None
    n_calls = len(lora_calls)
    lora.load_networks(["myLora"], [1.0], [1.0], [None])
    assert len(lora_calls) == n_calls + 1 and len(merged) == 3          # torch UNet active: the stock path only
    # "Reload UI": a second import wraps the ORIGINAL constructors / loader again, not the wrappers
    mod2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod2)
    assert samplers_mod.all_samplers[0].constructor("m2").stock == "Euler a"
    n_calls = len(lora_calls)
    lora.load_networks(["myLora"])
    assert len(lora_calls) == n_calls + 1
    # B6: a model without the transformers CLIP wrapper is left alone
    bridge = sub("webui_bridge")
    assert bridge.install_clip_hook(types.SimpleNamespace(cond_stage_model=None)) is None
    # B6 + Lora: a loaded network that touches the text encoder with a non-zero multiplier sends prompts through the torch tower
    # (the packed tower never runs the patched Linear forwards that apply those deltas)
    csm = torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.Linear(2, 2))
    te_mod, unet_mod_ = types.SimpleNamespace(sd_module=csm[1]), types.SimpleNamespace(sd_module=torch.nn.Linear(2, 2))
    net = types.SimpleNamespace(te_multiplier=0.7, modules={"lora_te_x": te_mod, "lora_unet_y": unet_mod_})
    lora.loaded_networks = []
    assert not bridge.text_encoder_networks_active(csm, lora)
    lora.loaded_networks = [types.SimpleNamespace(te_multiplier=1.0, modules={"lora_unet_y": unet_mod_})]
    assert not bridge.text_encoder_networks_active(csm, lora)           # UNet-only network
    lora.loaded_networks.append(net)
    assert bridge.text_encoder_networks_active(csm, lora) and bridge.text_encoder_networks_active(csm)      # (resolved from sys.modules)
    net.te_multiplier = 0
    assert not bridge.text_encoder_networks_active(csm, lora)


def test_hypernetwork_file_is_loaded_without_unpickling_arbitrary_objects(tmp_path):
    """Hypernetwork.load reads a user-supplied .pt with torch.load(weights_only=True): a real hypernetwork file (tensors, tuples, ints,
    strs, lists, bools — modules/hypernetworks/hypernetwork.py:246-300) loads, a pickle that would execute code on load is refused
    (the reference routes torch.load through modules/safe.py's restricted unpickler for the same reason)."""
    import pickle
    hn = sub("hypernetwork")
    g = torch.Generator().manual_seed(5)
    mk = lambda dim: {"linear.0.weight": torch.randn(2 * dim, dim, generator=g), "linear.0.bias": torch.zeros(2 * dim),
                      "linear.1.weight": torch.randn(dim, 2 * dim, generator=g), "linear.1.bias": torch.zeros(dim)}
    state = {"layer_structure": [1, 2, 1], "activation_func": "linear", "is_layer_norm": False, "activate_output": False,
             "dropout_structure": None, "name": "tiny_hn", "step": 100, "sd_checkpoint": "abc", "sd_checkpoint_name": "m",
             64: (mk(64), mk(64)), 128: (mk(128), mk(128))}
    good = tmp_path / "tiny_hn.pt"
    torch.save(state, str(good))
    h = hn.Hypernetwork().load(str(good))
    assert h.name == "tiny_hn" and sorted(h.layers) == [64, 128] and len(h.layers[64]) == 2

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    bad = tmp_path / "evil.pt"
    torch.save({"layer_structure": [1, 2, 1], "payload": Evil()}, str(bad))
    with pytest.raises(pickle.UnpicklingError):
        hn.Hypernetwork().load(str(bad))


def test_img2img_frontend_helpers_match_reference_fixtures(golden_dir):
    """The PIL side of img2img / inpainting against fixtures produced by the reference's own files (tests/golden/make_golden.py
    gen_img2img_frontend): resize_image modes 1 (cover) and 2 (fit + border fill) incl. an L-mode mask and named upscalers
    (modules/images.py:293-326), modules/masking.py (crop regions with padding, the all-black cases, the aspect-ratio expansion, the
    blur-ladder fill — bit-identical uint8 images) and create_binary_mask / uncrop / apply_overlay (modules/processing.py:70-98)."""
    from PIL import Image, ImageOps
    from tests.test_oracle_pins import _golden_module
    mg = _golden_module()
    up, masking, shared = sub("upscaler"), sub("masking"), sub("shared")
    z = np.load(os.path.join(golden_dir, "img2img_frontend.npz"))
    img, mask, rgba, black = mg.frontend_inputs()
    shared.sd_upscalers[:] = up.builtin_upscalers() if hasattr(up, "builtin_upscalers") else shared.sd_upscalers
    for k, (mode, w, h, name) in enumerate(mg.FRONTEND_RESIZE_CASES):
        assert np.array_equal(np.array(up.resize_image(mode, img, w, h, upscaler_name=name)), z[f"resize{k}"]), (mode, w, h, name)
    assert np.array_equal(np.array(up.resize_image(2, mask, 64, 64)), z["resize_mask_m2"])
    for k, pad in enumerate((0, 4, 32)):
        assert tuple(z[f"crop_v2_{k}"]) == masking.get_crop_region_v2(mask, pad)
        assert tuple(z[f"crop_{k}"]) == masking.get_crop_region(mask, pad)
        assert tuple(z[f"crop_black_{k}"]) == masking.get_crop_region(black, pad)
        assert tuple(z[f"crop_v2_{k}"]) == masking.get_crop_region_v2(np.array(mask), pad)        # arrays are accepted too
    assert masking.get_crop_region_v2(black, 3) is None
    for k, (box, pw, ph, iw, ih) in enumerate(mg.FRONTEND_CROP_CASES):
        assert tuple(int(v) for v in z[f"expand{k}"]) == tuple(int(v) for v in masking.expand_crop_region(box, pw, ph, iw, ih)), k
    assert np.array_equal(np.array(masking.fill(img, mask)), z["fill"])
    for tag, m, rnd in (("rgba_round", rgba, True), ("rgba_soft", rgba, False), ("l", mask, True), ("rgb", img, True)):
        assert np.array_equal(np.array(masking.create_binary_mask(m, round=rnd)), z[f"binary_{tag}"]), tag
    overlay = Image.new('RGBa', (img.width, img.height))
    overlay.paste(img.convert("RGBA").convert("RGBa"), mask=ImageOps.invert(mask.convert('L')))
    overlay = overlay.convert('RGBA')
    gen = Image.fromarray(np.random.RandomState(99).randint(0, 256, size=(80, 96, 3)).astype(np.uint8))
    a_img, a_orig = masking.apply_overlay(gen, None, overlay)
    assert np.array_equal(np.array(a_img), z["overlay_full"]) and np.array_equal(np.array(a_orig), z["overlay_full_orig"])
    small = Image.fromarray(np.random.RandomState(98).randint(0, 256, size=(64, 64, 3)).astype(np.uint8))
    b_img, b_orig = masking.apply_overlay(small, (30, 20, 40, 25), overlay)
    assert np.array_equal(np.array(b_img), z["overlay_paste"]) and np.array_equal(np.array(b_orig), z["overlay_paste_orig"])
    assert masking.apply_overlay(gen, None, None)[0] is gen


def test_mask_blur_kernel_is_the_one_cv2_would_build():
    """The separable Gaussian of the mask blur (modules/processing.py:1621-1631 calls cv2.GaussianBlur(mask, (k, 1), sigma) with
    k = 2 * int(2.5 * sigma + 0.5) + 1; cv2 is absent here, so the reference itself cannot run this step — parity unpinned for it):
    kernel size formula, normalisation, symmetry, BORDER_REFLECT_101 handling, and agreement with scipy's independently written
    1-D Gaussian (same taps when truncated at the same radius, mirror border) to one grey level."""
    from scipy.ndimage import correlate1d
    masking = sub("masking")
    for sigma, k in ((1, 7), (4, 21), (7, 37), (0.6, 5)):
        w = masking.gaussian_kernel_1d(sigma)
        assert len(w) == k and abs(w.sum() - 1.0) < 1e-12 and np.allclose(w, w[::-1])
        assert np.allclose(w[k // 2 + 1] / w[k // 2], np.exp(-1.0 / (2 * sigma ** 2)))
    g = np.random.RandomState(5)
    m = (g.rand(40, 56) > 0.7).astype(np.uint8) * 255
    for sigma in (2, 4):
        for axis in (0, 1):
            got = masking.gaussian_blur_axis(m, sigma, axis)
            want = correlate1d(m.astype(np.float64), masking.gaussian_kernel_1d(sigma), axis=axis, mode="mirror")
            assert got.dtype == np.uint8 and got.shape == m.shape
            assert np.abs(got.astype(np.float64) - want).max() <= 0.5 + 1e-9
    assert np.array_equal(masking.gaussian_blur_axis(np.full((8, 8), 200, np.uint8), 3, 1), np.full((8, 8), 200, np.uint8))
    from PIL import Image
    out = masking.blur_mask(Image.fromarray(m), 4, 4)
    assert out.size == (56, 40) and np.array(out).max() <= 255 and 0 < np.array(out).mean() < 255
    assert masking.blur_mask(Image.fromarray(m), 0, 0).tobytes() == Image.fromarray(m).tobytes()


def test_bench_pmc_child_command_line(monkeypatch):
    """bench.py --pmc-traffic profiles ONE job of the same workload in child processes: the child's flags keep the workload selection and
    drop everything that shapes the timed region or would start children of its own; without measured passes on the box `traffic` is null
    (a figure from another box is never reported)."""
    monkeypatch.syspath_prepend(ROOT)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")
    got = bench.pmc_child_args(["--config", "c3", "--steps", "5", "--warmup=2", "--pmc-traffic", "--gpus", "1", "--verify-shards", "--no-cpu-baseline",
                                "--pmc-timeout", "60"])
    assert got == ["--config", "c3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--no-dropin", "--no-pmc-traffic"]
    args = bench.parse()
    assert args.pmc_traffic is None                       # resolved in main(): on for the named c1 workload where rocprofv3 exists
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-pmc-traffic"])
    assert bench.parse().pmc_traffic is False
    monkeypatch.setattr(sys, "argv", ["bench.py", "--pmc-traffic"])
    assert bench.parse().pmc_traffic is True
    path = os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
    if not os.path.exists(path):
        assert bench.pmc_traffic(args) == (None, None)


def test_restart_plan_matches_the_pinned_oracle_restatement():
    """sd_samplers.restart_plan (the repo's own form of the schedule builder) against oracle.kdiffusion.restart_step_list, which
    tests/test_oracle_pins.py pins to modules/sd_samplers_extra.py: same pairs, same fp32 bits, for no / one / two restarts and an
    explicit restart_list whose target lies below the trigger (never climbs: no ladder)."""
    ss = sub("sd_samplers")
    from oracle import kdiffusion as kd
    smin, smax = 0.0291672, 14.614641
    for steps, restart_list in ((8, None), (22, None), (40, None), (12, {0.5: [4, 2, 3.0]}), (12, {3.0: [4, 1, 0.5]})):
        sig = kd.get_sigmas_karras(steps, smin, smax)
        got, want = ss.restart_plan(sig.clone(), restart_list), kd.restart_step_list(sig.clone(), None if restart_list is None else dict(restart_list))
        assert len(got) == len(want)
        assert all(torch.equal(a0, b0) and torch.equal(a1, b1) for (a0, a1), (b0, b1) in zip(got, want))
    assert len(ss.restart_plan(kd.get_sigmas_karras(22, smin, smax))) == 13 + 9


def _stub_webui(monkeypatch, opts=None):
    """A webui small enough for a CPU test: modules.{shared, script_callbacks, sd_models, sd_samplers, sd_unet, sd_samplers_common,
    scripts} with the attributes the extension script and the engine samplers touch."""
    import collections
    import types
    class SamplerData(collections.namedtuple("SamplerData", ["name", "constructor", "aliases", "options"])):
        def total_steps(self, steps):                         # modules/sd_samplers_common.py:14-18
            return steps * 2 if self.options.get("second_order", False) else steps
    w = types.SimpleNamespace(stock_calls=[], tqdm=[], stored=[])

    def mk(name):
        def ctor(model):
            w.stock_calls.append((name, model))
            calls = []
            return types.SimpleNamespace(stock=name, config=None, calls=calls,
                                         sample=lambda p, *a, **k: calls.append(("sample", p)) or "stock-samples",
                                         sample_img2img=lambda p, *a, **k: calls.append(("sample_img2img", p)) or "stock-samples")
        return ctor
    samplers_mod = types.ModuleType("modules.sd_samplers")
    samplers_mod.all_samplers = [SamplerData("Euler a", mk("Euler a"), ["k_euler_a"], {"uses_ensd": True}),
                                 SamplerData("DPM++ 2M", mk("DPM++ 2M"), ["k_dpmpp_2m"], {"scheduler": "karras"}),
                                 SamplerData("DDIM", mk("DDIM"), ["ddim"], {})]
    samplers_mod.all_samplers_map = {x.name: x for x in samplers_mod.all_samplers}
    samplers_mod.set_samplers = lambda: None
    unet_mod = types.ModuleType("modules.sd_unet")
    unet_mod.current_unet = None
    cb = types.ModuleType("modules.script_callbacks")
    cb.on_list_unets = cb.on_list_optimizers = cb.on_model_loaded = lambda f: None
    cb.callback_map = dict(callbacks_cfg_denoiser=[], callbacks_cfg_denoised=[], callbacks_cfg_after_cfg=[], callbacks_extra_noise=[],
                           callbacks_model_loaded=["something unrelated"])
    sdm = types.ModuleType("modules.sd_models")
    sdm.checkpoints_list = {}
    sdm.read_state_dict = lambda fn, map_location=None: {}
    shared_stub = types.ModuleType("modules.shared")
    shared_stub.sd_model = types.SimpleNamespace(alphas_cumprod=sub("schema").make_alphas_cumprod(), model=torch.nn.Module())
    shared_stub.opts = types.SimpleNamespace(**(opts or {}))
    shared_stub.state = types.SimpleNamespace(interrupted=False, skipped=False, sampling_step=0, sampling_steps=0, current_latent=None)
    shared_stub.cmd_opts = types.SimpleNamespace(disable_nan_check=True)
    shared_stub.total_tqdm = types.SimpleNamespace(update=lambda: w.tqdm.append(1))
    common = types.ModuleType("modules.sd_samplers_common")
    common.store_latent = lambda decoded: w.stored.append(decoded)
    scripts_mod = types.ModuleType("modules.scripts")
    scripts_mod.MaskBlendArgs = type("MaskBlendArgs", (), {"__init__": lambda self, *a, **k: setattr(self, "args", (a, k))})
    root = types.ModuleType("modules")
    mods = {"script_callbacks": cb, "sd_models": sdm, "shared": shared_stub, "sd_samplers": samplers_mod, "sd_unet": unet_mod,
            "sd_samplers_common": common, "scripts": scripts_mod}
    monkeypatch.setitem(sys.modules, "modules", root)
    for name, m in mods.items():
        setattr(root, name, m)
        monkeypatch.setitem(sys.modules, "modules." + name, m)
    import importlib.util
    path = os.path.join(ROOT, "stable-diffusion-webui_amd", "extension", "scripts", "mi355x_engine.py")
    spec = importlib.util.spec_from_file_location("mi355x_engine_script_c", path)
    w.script = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(w.script)
    w.__dict__.update(mods)
    # an "active" engine UNet (no GPU here: the engine handle is a stand-in, nothing below launches a kernel)
    unet = sub("sd_unet").Mi355xUnet(lambda: {}, unet_cfg=sub("schema").tiny_unet())
    unet.engine, unet._sd = types.SimpleNamespace(device=0), {}
    w.unet = unet
    return w


@pytest.fixture
def webui(monkeypatch):
    w = _stub_webui(monkeypatch, opts=dict(eta_ancestral=0.5, sigma_min=0.05, always_discard_next_to_last_sigma=True,
                                           live_previews_enable=True, show_progress_every_n_steps=2, live_preview_content="Combined"))
    yield w
    sub("shared").unbind_webui()


def _job(**kw):
    import types
    p = types.SimpleNamespace(scheduler="Karras", hr_scheduler=None, is_hr_pass=False, sampler_noise_scheduler_override=None,
                              extra_generation_params={}, eta=None, s_min_uncond=0.0, steps=20, batch_size=1, iteration=0, rng=None,
                              get_token_merging_ratio=lambda for_hr=False: 0.0)
    p.__dict__.update(kw)
    return p


def test_engine_samplers_read_the_webuis_own_opts_and_state(webui):
    """VERDICT r3 missing #1: once the extension script ran, ``stable-diffusion-webui_amd.shared.opts / state / cmd_opts`` ARE the
    webui's objects (names the webui lacks fall back to the reference defaults), so the user's eta / sigma_min / "always discard
    next-to-last sigma", Interrupt, the progress counters and the live-preview store reach the engine samplers as they reach the
    stock ones (modules/sd_samplers_kdiffusion.py:79-132, modules/sd_samplers_common.py:256-281, 303-305,
    modules/sd_samplers_cfg_denoiser.py:157-158)."""
    amd_shared, ss = sub("shared"), sub("sd_samplers")
    assert amd_shared.webui is webui.shared and amd_shared.opts.eta_ancestral == 0.5 and amd_shared.opts.uni_pc_order == 3
    assert amd_shared.cmd_opts.disable_nan_check is True and amd_shared.cmd_opts.no_half is False
    webui.shared.opts.eta_ancestral = 0.25                    # a settings change AFTER the bind is seen: the view reads through
    assert amd_shared.opts.eta_ancestral == 0.25
    webui.sd_unet.current_unet = webui.unet
    s = webui.sd_samplers.all_samplers_map["Euler a"].constructor(webui.shared.sd_model)
    s.config = webui.sd_samplers.all_samplers_map["Euler a"]
    assert isinstance(s, ss.KDiffusionSampler) and webui.stock_calls == []
    p = _job()
    sig = s.get_sigmas(p, 20)
    assert sig.shape == (21,) and abs(float(sig[-2]) - 0.05) > 1e-3          # 21 + 1 levels from sigma_min 0.05, the penultimate (0.05) dropped
    ref = ss.get_sigmas_karras(n=21, sigma_min=0.05, sigma_max=float(s.model_wrap.sigmas[-1]), device="cpu")
    assert torch.equal(sig, torch.cat([ref[:-2], ref[-1:]]))
    assert p.extra_generation_params == {"Discard penultimate sigma": True, "Schedule type": "Karras", "Schedule min sigma": 0.05}
    kw = s.initialize(p)
    assert kw["eta"] == 0.25 and s.eta == 0.25 and p.extra_generation_params["Eta"] == 0.25
    # progress: the webui's state object and its console bar
    assert s.launch_sampling(20, lambda: "done") == "done" and webui.shared.state.sampling_steps == 20
    s.callback_state({"i": 7})
    assert webui.shared.state.sampling_step == 7 and webui.tqdm == [1]
    # Interrupt / Skip: checked before anything else in CFGDenoiser.forward; launch_sampling returns the last latent
    s.last_latent = "last-latent"
    for flag in ("interrupted", "skipped"):
        setattr(webui.shared.state, flag, True)
        with pytest.raises(ss.InterruptedException):
            s.model_wrap_cfg.forward(None, None, None, None, 7.0)
        assert s.launch_sampling(20, lambda: s.model_wrap_cfg.forward(None, None, None, None, 7.0)) == "last-latent"
        setattr(webui.shared.state, flag, False)
    # the live-preview store is the webui's own function
    amd_shared.store_latent("x0")
    assert webui.stored == ["x0"] and amd_shared.MaskBlendArgs is webui.scripts.MaskBlendArgs
    amd_shared.unbind_webui()
    assert amd_shared.opts.eta_ancestral == 1.0 and amd_shared.webui is None and amd_shared.MaskBlendArgs is not webui.scripts.MaskBlendArgs


def test_sampler_rows_fan_out_over_the_devices_of_one_webui_process(webui, monkeypatch):
    """VERDICT r5 missing #2 (modules/call_queue.py:8-13, modules/cmd_args.py:106): inside a webui the batch of ONE
    sampling call is spread over opts.mi355x_devices at the sampler row — contiguous row ranges, one worker + one engine UNet replica per
    device, per-range ImageRNG over the range's own seeds in the state the batch's generator is in, conds / img2img tensors / SDXL dict
    conds / MulticondLearnedConditioning sliced alike, LoRA merges re-applied on a replica — and the concatenation is what the
    single-device call returns.  (The engines are stand-ins here; the real ones run the same split in tests/test_gpu_models.py's
    device-pool test.)"""
    import types
    bridge, amd_shared, amd_rng = sub("webui_bridge"), sub("shared"), sub("rng")
    webui.sd_unet.current_unet = webui.unet
    view = bridge.engine_model_view(webui.shared.sd_model, webui.sd_unet)
    made, merged = [], []

    class FakeSampler:                                        # a "sampling loop" whose output depends on everything that is per row
        def __init__(self, v):
            self.view, self.config = v, None
            made.append(v.engine.device)

        def _run(self, p, x, cond, uncond, extra):
            c = cond["crossattn"] if isinstance(cond, dict) else (torch.stack([b[0] for b in cond.batch]) if hasattr(cond, "batch") else cond)
            u = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in uncond]) if isinstance(uncond, list) else uncond
            noise = p.rng.next()                              # the step noise of an ancestral sampler
            seeds = torch.tensor(p.seeds, dtype=torch.float32).view(-1, 1, 1, 1)
            return 2.0 * x + c.mean(dim=(1, 2)).view(-1, 1, 1, 1) - u.mean(dim=1).view(-1, 1, 1, 1) + noise + 1e-3 * seeds + extra

        def sample(self, p, x, cond, uncond, steps=None, image_conditioning=None):
            return self._run(p, x, cond, uncond, 0.0 if image_conditioning is None else image_conditioning.mean(dim=(1, 2, 3)).view(-1, 1, 1, 1))

        def sample_img2img(self, p, x, noise, cond, uncond, steps=None, image_conditioning=None):
            return self._run(p, x, cond, uncond, noise * 0.5 + p.init_latent * 0.25)
    replicas = {}

    def fake_unet_on_device(unet, device, nth=0):
        if device == unet.engine.device and nth == 0:
            return unet
        return replicas.setdefault(device, types.SimpleNamespace(engine=types.SimpleNamespace(device=device), unet_cfg=unet.unet_cfg, checkpoint=lambda: {}))
    monkeypatch.setattr(bridge, "_unet_on_device", fake_unet_on_device)
    monkeypatch.setattr(bridge, "_torch_device", lambda d: torch.device("cpu"))
    monkeypatch.setattr(bridge, "serial_device_workers", True)

    class CpuGenerator:                                       # the package's Generator draws on the device (Philox kernel): a host stand-in
        def __init__(self, seed, device):
            self.g = torch.Generator().manual_seed(int(seed))

        def randn(self, shape):
            return torch.randn(tuple(shape), generator=self.g)
    monkeypatch.setattr(amd_rng, "Generator", CpuGenerator)
    n, shape = 5, (4, 8, 8)
    seeds, subseeds = [100 + i for i in range(n)], [900 + i for i in range(n)]
    g = torch.Generator().manual_seed(3)
    cond, x = torch.randn(n, 77, 16, generator=g), None
    uncond = [[float(i), float(i) + 0.5] for i in range(n)]   # a per-image list (the uncond schedules' container)

    def job():
        r = amd_rng.ImageRNG(shape, seeds, subseeds, 0.0, device="cpu")
        p = _job(batch_size=n, seeds=list(seeds), subseeds=list(subseeds), rng=r)
        return p, r.next()                                    # the webui draws x before the sampler runs
    p, x = job()
    whole = FakeSampler(view).sample(p, x, cond, uncond, image_conditioning=torch.ones(n, 5, 8, 8) * torch.arange(n).view(-1, 1, 1, 1))
    made.clear()
    view._networks_applied = (("some-lora",), lambda v: merged.append(v.engine.device))
    p, x = job()
    split = bridge.sample_over_devices(FakeSampler, view, "sample", p, (x, cond, uncond),
                                       dict(image_conditioning=torch.ones(n, 5, 8, 8) * torch.arange(n).view(-1, 1, 1, 1)), [0, 1, 2])
    assert torch.equal(split, whole) and made == [0, 1, 2] and merged == [1, 2, 0]         # 5 rows over 3 devices: 2 + 2 + 1; LoRA merged on the replicas
                                                                                           # (then once more on the primary: the loader's module state)
    split2 = bridge.sample_over_devices(FakeSampler, view, "sample", job()[0], (x, cond, uncond),
                                        dict(image_conditioning=torch.ones(n, 5, 8, 8) * torch.arange(n).view(-1, 1, 1, 1)), [0, 1, 2])
    assert torch.equal(split2, whole) and merged == [1, 2, 0]          # the same merges are not applied twice
    # SDXL dict conds, a MulticondLearnedConditioning-shaped container, img2img tensors on p
    class Multi:
        def __init__(self, shape, batch):
            self.shape, self.batch = shape, batch
    dict_cond = {"crossattn": cond, "vector": torch.randn(n, 8, generator=g)}
    multi = Multi((n,), [[cond[i]] for i in range(n)])
    for c in (dict_cond, multi):
        p, x = job()
        a = FakeSampler(view).sample(p, x, c, uncond)
        p, x = job()
        assert torch.equal(bridge.sample_over_devices(FakeSampler, view, "sample", p, (x, c, uncond), {}, [0, 1]), a)
    init, noise = torch.randn(n, 4, 8, 8, generator=g), torch.randn(n, 4, 8, 8, generator=g)
    p, x = job()
    p.init_latent = init
    a = FakeSampler(view).sample_img2img(p, x, noise, cond, uncond)
    p, x = job()
    p.init_latent = init
    assert torch.equal(bridge.sample_over_devices(FakeSampler, view, "sample_img2img", p, (x, noise, cond, uncond), {}, [0, 1]), a)
    # the row's own call dispatches on opts.mi355x_devices (and not for a single image, nor without the option)
    seen = []
    monkeypatch.setattr(bridge, "sample_over_devices", lambda make, v, name, p, args, kwargs, devs: seen.append((name, list(devs), args[0].shape[0])) or "fanned-out")
    row = webui.sd_samplers.all_samplers_map["Euler a"]
    s = row.constructor(webui.shared.sd_model)
    s.config = row
    webui.shared.opts.mi355x_devices = "0,1"
    try:
        p, x = job()
        assert s.sample(p, x, cond, uncond) == "fanned-out" and seen == [("sample", [0, 1], n)]
        one = amd_rng.ImageRNG(shape, seeds[:1], device="cpu")
        assert bridge.job_needs_stock_sampler(_job(), webui.shared.sd_model) is None
        monkeypatch.setattr(s, "sample", s.sample)          # (keep the wrapper; a one-image call must take the fused path: checked by count)
        n_seen = len(seen)
        try:
            s.sample(_job(batch_size=1, seeds=seeds[:1], rng=one), one.next(), cond[:1], uncond[:1])
        except Exception:
            pass                                              # the stand-in engine cannot sample; what matters is that nothing fanned out
        assert len(seen) == n_seen
    finally:
        del webui.shared.opts.mi355x_devices


def test_refiner_checkpoint_switch_on_the_engine_path_inside_a_webui(webui):
    """VERDICT r5 missing #3 (modules/sd_samplers_common.py:158-202): with the webui's checkpoint loader bound
    (webui_bridge.install_refiner_switch), a job that names a refiner checkpoint STAYS on the engine sampler; at the switch point the
    webui reloads its model (reload_model_weights -> apply_unet activates the refiner checkpoint's engine UNet), p.setup_conds() re-encodes
    the prompts, and the sampler continues on the new engine with the new conds, a fresh wrapped denoiser and no cached context — in the
    reference's order and with its infotext keys; before the switch ratio, on the same checkpoint, or on the wrong hires pass nothing moves."""
    import types
    ss, bridge, amd_shared = sub("sd_samplers"), sub("webui_bridge"), sub("shared")
    webui.sd_unet.current_unet = webui.unet
    base_info, ref_info = types.SimpleNamespace(short_title="base"), types.SimpleNamespace(short_title="refiner [abc]")
    webui.shared.sd_model.sd_checkpoint_info = base_info
    refiner_unet = sub("sd_unet").Mi355xUnet(lambda: {}, unet_cfg=sub("schema").tiny_unet())
    refiner_unet.engine, refiner_unet._sd = types.SimpleNamespace(device=0), {}
    log = []

    def reload_model_weights(sd_model=None, info=None, forced_reload=False):
        log.append(("reload", info.short_title))
        webui.shared.sd_model.sd_checkpoint_info = info       # modules/sd_models.py:940-1000: same object, new weights ...
        webui.sd_unet.current_unet = refiner_unet             # ... and apply_unet() activates the checkpoint's own option
    webui.sd_models.reload_model_weights = reload_model_weights

    class Skip:
        def __enter__(self): log.append("skip-config-on")
        def __exit__(self, *a): log.append("skip-config-off")
    webui.sd_models.SkipWritingToConfig = Skip
    devices = types.SimpleNamespace(torch_gc=lambda: log.append("gc"))
    try:
        bridge.install_refiner_switch(webui.sd_models, webui.shared, webui.sd_unet, devices)
        row = webui.sd_samplers.all_samplers_map["Euler a"]
        s = row.constructor(webui.shared.sd_model)
        s.config = row
        assert isinstance(s, ss.KDiffusionSampler)
        p = _job(refiner_checkpoint_info=ref_info, refiner_switch_at=0.5, enable_hr=False)
        p.setup_conds = lambda: log.append("setup_conds")
        p.get_conds = lambda: ("new-cond", "new-uncond")
        assert bridge.job_needs_stock_sampler(p, webui.shared.sd_model) is None      # no hand-over to the stock sampler any more
        den = s.model_wrap_cfg
        den.p, den.step, den.total_steps = p, 4, 20
        s.sampler_extra_args = {"cond": "old-cond", "uncond": "old-uncond", "y": "old-y", "uy": "old-uy"}
        amd_shared.opts.refiner_switch_by_sample_steps = True
        wrap_before = den.inner_model
        den._ctx_key = "cached"
        assert ss.apply_refiner(den) is False and log == []   # 4 / 20 < 0.5
        den.step = 10
        assert ss.apply_refiner(den) is True
        assert log == ["skip-config-on", ("reload", "refiner [abc]"), "skip-config-off", "gc", "setup_conds"]
        assert s.sd_model.engine is refiner_unet.engine and amd_shared.sd_model is s.sd_model
        assert s.sampler_extra_args == {"cond": "new-cond", "uncond": "new-uncond"} and den._ctx_key is None and den.model_wrap is None
        assert den.inner_model is not wrap_before
        assert p.extra_generation_params["Refiner"] == "refiner [abc]" and p.extra_generation_params["Refiner switch at"] == 0.5
        assert p.extra_generation_params["Refiner switch by sampling steps"] is True
        n = len(log)
        assert ss.apply_refiner(den) is False and len(log) == n       # the model already IS the refiner checkpoint
        # hires fix: "second pass" only (the default) leaves the first pass alone
        webui.shared.sd_model.sd_checkpoint_info = base_info
        amd_shared.opts.hires_fix_refiner_pass = "second pass"
        p2 = _job(refiner_checkpoint_info=ref_info, refiner_switch_at=0.0, enable_hr=True, is_hr_pass=False)
        den.p = p2
        assert ss.apply_refiner(den) is False and len(log) == n
        # the webui activated no engine UNet for the refiner checkpoint: a loud stop, never a silent run on the wrong weights
        p3 = _job(refiner_checkpoint_info=ref_info, refiner_switch_at=0.0, enable_hr=False)
        p3.setup_conds, p3.get_conds = (lambda: None), (lambda: ("c", "uc"))
        den.p = p3

        def reload_to_none(sd_model=None, info=None, forced_reload=False):
            webui.shared.sd_model.sd_checkpoint_info = info
            webui.sd_unet.current_unet = None
        webui.sd_models.reload_model_weights = reload_to_none
        with pytest.raises(RuntimeError, match="did not activate an engine UNet"):
            ss.apply_refiner(den)
    finally:
        bridge.uninstall_refiner_switch()
        for k in ("refiner_switch_by_sample_steps", "hires_fix_refiner_pass"):
            if hasattr(webui.shared.opts, k):
                delattr(webui.shared.opts, k)


def test_engine_sampler_rows_fall_back_when_the_webui_job_needs_torch_side_hooks(webui):
    """VERDICT r3 missing #1 / #2, r4 missing #6: a row builds the STOCK sampler while any cfg_denoiser / cfg_denoised / cfg_after_cfg /
    extra_noise script callback is registered (modules/sd_samplers_cfg_denoiser.py:212, 279, 307; sd_samplers_kdiffusion.py:146-151); ToMe
    (modules/sd_models.py:1011-1034) and Hypertile (extensions-builtin/hypertile) jobs are never silently ignored: the sampling call is
    forwarded to a stock sampler built on the spot and every UNet evaluation under it goes to the webui's own — patched — torch UNet
    (SURVEY.md section 7 (vi)); a refiner-checkpoint job is forwarded the same way."""
    import types
    ss, bridge = sub("sd_samplers"), sub("webui_bridge")
    webui.sd_unet.current_unet = webui.unet
    row = webui.sd_samplers.all_samplers_map["DPM++ 2M"]
    model = webui.shared.sd_model
    assert isinstance(row.constructor(model), ss.KDiffusionSampler) and webui.stock_calls == []
    for i, name in enumerate(bridge.SAMPLER_CALLBACK_LISTS):
        webui.script_callbacks.callback_map[name].append(object())
        assert bridge.registered_sampler_callbacks(webui.script_callbacks) == [name]
        assert row.constructor(model).stock == "DPM++ 2M" and len(webui.stock_calls) == i + 1
        webui.script_callbacks.callback_map[name].clear()
    assert isinstance(row.constructor(model), ss.KDiffusionSampler)
    x = torch.zeros(1, 4, 8, 8)

    # the webui's torch UNet, as modules/sd_unet.py:86-93 leaves it: forward goes to current_unet while one is set
    seen = []

    class TorchUnet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1))

        def forward(self, x, timesteps=None, context=None, *args, **kwargs):
            if webui.sd_unet.current_unet is not None:
                return webui.sd_unet.current_unet.forward(x, timesteps, context, *args, **kwargs)
            seen.append((tuple(x.shape), kwargs.get("y")))
            return x + 1.0
    model.model.diffusion_model = TorchUnet()

    def forwarded(reason, p=None, **model_attrs):
        """sample / sample_img2img of an engine row land in a stock sampler, with the reason noted"""
        s = row.constructor(model)
        s.config = row
        for k, v in model_attrs.items():
            setattr(model, k, v)
        try:
            n = len(webui.stock_calls)
            assert s.sample(p or _job(), x, None, None) == "stock-samples" and reason in s.stock_reason
            assert s.sample_img2img(p or _job(), x, x, None, None) == "stock-samples" and len(webui.stock_calls) == n + 2
        finally:
            for k in model_attrs:
                delattr(model, k)
    # ToMe applied to the model (first pass: processing.py:841 runs before the sampler is built): the UNet call reaches the torch UNet,
    # with the option restored afterwards ...
    model.applied_token_merged_ratio = 0.5
    out = model.model.diffusion_model(x, torch.zeros(1), torch.zeros(1, 77, 64), y=None)
    assert torch.equal(out, x + 1.0) and seen == [((1, 4, 8, 8), None)] and webui.sd_unet.current_unet is webui.unet
    assert "token merging is active" in webui.unet.torch_fallback_reason
    del model.applied_token_merged_ratio
    forwarded("token merging is active", applied_token_merged_ratio=0.5)
    # ... or only requested for the hires pass (applied at processing.py:1442, AFTER the hires sampler was built)
    forwarded("token merging is requested", _job(is_hr_pass=True, get_token_merging_ratio=lambda for_hr=False: 0.3 if for_hr else 0.0))
    # Hypertile: the options, and the live module flags hypertile_hook_model leaves on the torch UNet
    webui.shared.opts.hypertile_enable_unet = True
    forwarded("Hypertile")
    webui.shared.opts.hypertile_enable_unet = False
    webui.shared.opts.hypertile_enable_unet_secondpass = True
    forwarded("second pass", _job(is_hr_pass=True))
    s_ok = row.constructor(model)                             # first pass of the same settings: nothing is tiled yet
    assert bridge.patched_unet_reason(model) is None and bridge.job_needs_stock_sampler(_job(), model) is None
    webui.shared.opts.hypertile_enable_unet_secondpass = False
    attn = torch.nn.Linear(2, 2)
    model.model.add_module("attn1", attn)
    setattr(attn, "__webui_hypertile_params", types.SimpleNamespace(enabled=False))
    setattr(model.model, "__webui_hypertile_layers", {"attn1": 1})
    assert not bridge.hypertile_unet_active(model)
    getattr(attn, "__webui_hypertile_params").enabled = True
    assert bridge.hypertile_unet_active(model) and "Hypertile" in bridge.patched_unet_reason(model)
    assert torch.equal(webui.unet.forward(x, torch.zeros(1), torch.zeros(1, 77, 64)), x + 1.0) and len(seen) == 2
    assert "Hypertile" in webui.unet.torch_fallback_reason and webui.sd_unet.current_unet is webui.unet
    # a torch UNet that cannot be reached (no diffusion_model on the bound model) is still a loud refusal, never a silent engine run
    saved, model.model.diffusion_model = model.model.diffusion_model, None
    with pytest.raises(NotImplementedError, match="SD Unet to None"):
        webui.unet.forward(x, torch.zeros(1), torch.zeros(1, 77, 64))
    model.model.diffusion_model = saved
    getattr(attn, "__webui_hypertile_params").enabled = False
    # refiner checkpoint with the webui's loader out of reach (this stub has no reload_model_weights): the stock sampler of the same
    # row takes the call
    assert ss.webui_refiner_switch is None
    n_stock = len(webui.stock_calls)
    s = row.constructor(model)
    s.config = row
    p = _job(refiner_checkpoint_info=types.SimpleNamespace(short_title="refiner"))
    assert s.sample(p, x, None, None) == "stock-samples" and len(webui.stock_calls) == n_stock + 1
    assert s.stock_delegate.config is row and s.stock_delegate.calls == [("sample", p)] and s.stock_reason == "refiner checkpoint switch"
    # the B4 decode hook leaves a Hypertile-tiled VAE to torch
    assert bridge.hypertile_active(torch.nn.Module()) is False


def test_sampler_rows_and_name_resolution_match_the_reference(golden_dir):
    """The row ORDER, names and options of the reference's three sampler tables, and get_sampler_and_scheduler (modules/sd_samplers.py:
    105-126: "DPM++ 2M Karras" -> ("DPM++ 2M", "Karras"), unknown names -> the first row / Automatic) against the reference function
    executed over the reference's own tables (tests/golden/make_golden.py::gen_sampler_names); process_images applies it to the job
    (fix_p_invalid_sampler_and_scheduler, modules/processing.py:842) together with the job's option overrides."""
    import json
    ss = sub("sd_samplers")
    z = json.load(open(os.path.join(golden_dir, "sampler_names.json")))
    assert [[r.name, r.options] for r in ss.all_samplers] == z["rows"]
    for (name, sched), conv, want in z["cases"]:
        assert list(ss.get_sampler_and_scheduler(name, sched, convert_automatic=conv)) == want, (name, sched, conv)
    p = types.SimpleNamespace(sampler_name="DPM++ 2M Karras", scheduler=None)
    ss.fix_p_invalid_sampler_and_scheduler(p)
    assert (p.sampler_name, p.scheduler) == ("DPM++ 2M", "Karras")
    # the job's option overrides: set for the job, restored afterwards (also when the job raises), unknown keys refused
    processing, shared = sub("processing"), sub("shared")
    seen = {}

    def fake_inner(pp):
        seen.update(eta=shared.opts.eta_ancestral, ensd=shared.opts.eta_noise_seed_delta, sampler=pp.sampler_name, scheduler=pp.scheduler)
        if getattr(pp, "boom", False):
            raise RuntimeError("job failed")
        return "done"
    orig = processing._process_images_inner
    processing._process_images_inner = fake_inner
    try:
        before = (shared.opts.eta_ancestral, shared.opts.eta_noise_seed_delta)
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=None, sampler_name="Euler a SGMUniform",
                                                        override_settings={"eta_ancestral": 0.5, "eta_noise_seed_delta": 31337, "sd_model_checkpoint": "x"})
        assert processing.process_images(p) == "done"
        assert seen == {"eta": 0.5, "ensd": 31337, "sampler": "Euler a", "scheduler": "SGM Uniform"}
        assert (shared.opts.eta_ancestral, shared.opts.eta_noise_seed_delta) == before
        p.boom = True
        with pytest.raises(RuntimeError, match="job failed"):
            processing.process_images(p)
        assert (shared.opts.eta_ancestral, shared.opts.eta_noise_seed_delta) == before
        p2 = processing.StableDiffusionProcessingTxt2Img(sd_model=None, sampler_name="Euler", override_settings={"eta_ancestral": 0.25},
                                                         override_settings_restore_afterwards=False)
        processing.process_images(p2)
        assert shared.opts.eta_ancestral == 0.25
        shared.opts.eta_ancestral = before[0]
        with pytest.raises(KeyError):
            processing.process_images(processing.StableDiffusionProcessingTxt2Img(sd_model=None, override_settings={"no_such_option": 1}))
    finally:
        processing._process_images_inner = orig


def test_process_images_iteration_loop_obeys_skip_and_interrupt(monkeypatch):
    """modules/processing.py:935-939: a pending Skip is cleared when the next batch starts (it ended the previous one), Interrupt / stop
    ends the job before the next batch; what was finished is returned.  The device work of a batch is stubbed out."""
    processing, shared = sub("processing"), sub("shared")
    calls = []

    class P(processing.StableDiffusionProcessingTxt2Img):
        def sample(self, conditioning, unconditional_conditioning, seeds, subseeds, subseed_strength, prompts):
            calls.append((self.iteration, list(seeds)))
            if self.iteration == 0:
                shared.state.skipped = True                   # "Skip" pressed during the first batch
            if self.iteration == 1:
                shared.state.interrupted = True               # "Interrupt" during the second
            return torch.zeros(len(seeds), 4, 8, 8)

    eng = types.SimpleNamespace(set_option=lambda *a: None)
    model = types.SimpleNamespace(engine=eng, device=torch.device("cpu"), alphas_cumprod=sub("schema").make_alphas_cumprod())
    monkeypatch.setattr(processing, "ImageRNG", lambda *a, **kw: None)
    monkeypatch.setattr(processing, "decode_latent_batch", lambda m, x, **kw: torch.zeros(x.shape[0], 3, 64, 64))
    monkeypatch.setattr(processing.ops, "image_to_u8", lambda x: torch.zeros(x.shape[0], 64, 64, 3, dtype=torch.uint8))
    monkeypatch.setattr(processing.sd_models, "apply_alpha_schedule_override", lambda m, p=None: None)
    monkeypatch.setattr(shared.state, "interrupted", False, raising=False)
    monkeypatch.setattr(shared.state, "skipped", False, raising=False)
    c = torch.zeros(6, 77, 8)
    res = processing.process_images(P(sd_model=model, c=c, uc=c, seed=50, batch_size=2, n_iter=3, sampler_name="Euler a", width=64, height=64))
    assert calls == [(0, [50, 51]), (1, [52, 53])]            # the third batch never started
    assert len(res.images) == 4 and res.all_seeds == [50, 51, 52, 53, 54, 55] and shared.state.skipped is False
    # interrupted before the first batch: an empty result, not an error
    res = processing.process_images(P(sd_model=model, c=c, uc=c, seed=50, batch_size=2, n_iter=3, sampler_name="Euler a", width=64, height=64))
    assert res.images == [] and res.latents is None and res.images_device is None


def test_clip_hook_binds_sd2_and_sdxl_text_towers(monkeypatch):
    """B6 for every text-encoder wrapper the webui builds (modules/sd_hijack.py:205-243): SD 2.x's FrozenOpenCLIPEmbedderWithCustomWords and
    both SDXL embedders get ``encode_with_transformers`` rebound to an engine tower of the right configuration, fed with the webui's own
    (textual-inversion patched) token embeddings; a text-encoder LoRA on one tower sends THAT tower's prompts through torch; uninstall
    restores the wrappers.  The engine and the packed tower are stand-ins here (no GPU): what is tested is the binding."""
    import types
    bridge, schema = sub("webui_bridge"), sub("schema")
    made = []

    class FakeEngine:
        def __init__(self, device):
            self.device, self.closed = device, False

        def close(self):
            self.closed = True

    class FakeEncoder:
        def __init__(self, engine, cfg, state_dict, prefix=None, slot=0, layer="last", layer_idx=None):
            self.engine, self.cfg, self.sd, self.layer, self.layer_idx, self.calls = engine, cfg, state_dict, layer, layer_idx, []
            made.append(self)

        def _rec(self, how, tokens, inputs_embeds):
            self.calls.append((how, tuple(tokens.shape), None if inputs_embeds is None else tuple(inputs_embeds.shape)))
            return torch.zeros(tokens.shape[0], tokens.shape[1], 4)
        encode_with_transformers = lambda self, t, inputs_embeds=None: self._rec("clip_l", t, inputs_embeds)
        encode_with_transformers_sdxl = lambda self, t, inputs_embeds=None: self._rec("clip_l_sdxl", t, inputs_embeds)
        encode_with_transformer_openclip = lambda self, t, inputs_embeds=None: self._rec("openclip", t, inputs_embeds)
        encode_with_transformer_openclip2 = lambda self, t, inputs_embeds=None: self._rec("openclip2", t, inputs_embeds)
    monkeypatch.setattr(sub("engine"), "Engine", FakeEngine)
    monkeypatch.setattr(sub("sd_hijack_clip"), "Mi355xClipTextEncoder", FakeEncoder)

    class EmbeddingsWithFixes(torch.nn.Module):               # modules/sd_hijack.py: keeps the nn.Embedding as .wrapped
        def __init__(self, wrapped):
            super().__init__()
            self.wrapped = wrapped

        def forward(self, ids):
            return self.wrapped(ids)

    def open_clip_tower(width, layers=2):                     # the attributes of open_clip's text tower the hook and the key map read
        m = torch.nn.Module()
        m.token_embedding = EmbeddingsWithFixes(torch.nn.Embedding(50, width))
        m.positional_embedding = torch.nn.Parameter(torch.zeros(77, width))
        m.transformer = torch.nn.Module()
        blocks = []
        for _ in range(layers):
            b = torch.nn.Module()
            b.ln_1, b.ln_2 = torch.nn.LayerNorm(width), torch.nn.LayerNorm(width)
            b.attn = torch.nn.MultiheadAttention(width, 2)
            b.mlp = torch.nn.Module()
            b.mlp.c_fc, b.mlp.c_proj = torch.nn.Linear(width, 4 * width), torch.nn.Linear(4 * width, width)
            blocks.append(b)
        m.transformer.resblocks = torch.nn.ModuleList(blocks)
        m.ln_final = torch.nn.LayerNorm(width)
        m.text_projection = torch.nn.Parameter(torch.zeros(width, width))
        return m

    def hf_tower(width=8):                                    # transformers' CLIPTextModel.text_model, as far as the hook reads it
        tm = torch.nn.Module()
        tm.embeddings = torch.nn.Module()
        tm.embeddings.token_embedding = EmbeddingsWithFixes(torch.nn.Embedding(50, width))
        tm.final_layer_norm = torch.nn.LayerNorm(width)
        return tm

    def wrapper(clsname, wrapped):
        cls = type(clsname, (torch.nn.Module,), {"forward": lambda self, x: x})
        w = cls()
        w.wrapped = wrapped
        w.stock_calls = []
        w.encode_with_transformers = lambda tokens, w=w: w.stock_calls.append(tuple(tokens.shape)) or "torch"
        return w

    tokens = torch.zeros(3, 77, dtype=torch.long)
    # ---- SD 2.x: one OpenCLIP-H tower, penultimate layer
    inner = torch.nn.Module()
    inner.model, inner.layer = open_clip_tower(16), "penultimate"
    sd2 = types.SimpleNamespace(cond_stage_model=wrapper("FrozenOpenCLIPEmbedderWithCustomWords", inner), is_sdxl=False)
    enc = bridge.install_clip_hook(sd2, lora_networks=types.SimpleNamespace(loaded_networks=[]))
    assert enc is made[-1] and enc.cfg == schema.openclip_h() and enc.layer == "penultimate"
    keys = set(enc.sd)
    assert schema.CLIP_PREFIX + "embeddings.token_embedding.weight" in keys and schema.CLIP_PREFIX + "encoder.layers.1.self_attn.q_proj.weight" in keys
    assert schema.CLIP_PREFIX + "text_projection.weight" in keys and schema.CLIP_PREFIX + "final_layer_norm.bias" in keys
    sd2.cond_stage_model.encode_with_transformers(tokens)
    assert enc.calls == [("openclip", (3, 77), (3, 77, 16))] and sd2.cond_stage_model.stock_calls == []
    bridge.uninstall_clip_hook(sd2)
    assert enc.engine.closed and sd2.cond_stage_model.encode_with_transformers(tokens) == "torch" and not hasattr(sd2.cond_stage_model, "_mi355x_clip")
    # ---- SDXL: conditioner.embedders = [CLIP-L read at hidden_states[11], OpenCLIP-bigG with the pooled projection, a non-text embedder]
    l_inner = torch.nn.Module()
    l_inner.transformer = torch.nn.Module()
    l_inner.transformer.text_model = hf_tower()
    l_inner.layer, l_inner.layer_idx = "hidden", 11
    g_inner = torch.nn.Module()
    g_inner.model, g_inner.layer, g_inner.legacy = open_clip_tower(1280, layers=1), "penultimate", False
    emb_l = wrapper("FrozenCLIPEmbedderForSDXLWithCustomWords", l_inner)
    emb_g = wrapper("FrozenOpenCLIPEmbedder2WithCustomWords", g_inner)
    conditioner = types.SimpleNamespace(embedders=[emb_l, emb_g, types.SimpleNamespace(note="ConcatTimestepEmbedderND: no text")])
    sdxl = types.SimpleNamespace(cond_stage_model=conditioner, is_sdxl=True)
    lora = types.SimpleNamespace(loaded_networks=[])
    encs = bridge.install_clip_hook(sdxl, lora_networks=lora)
    assert isinstance(encs, list) and len(encs) == 2
    assert encs[0].cfg == schema.sd15_clip() and (encs[0].layer, encs[0].layer_idx) == ("hidden", 11)
    assert encs[1].cfg == schema.openclip_bigg()
    assert schema.CLIP_PREFIX + "embeddings.token_embedding.weight" in encs[0].sd          # (EmbeddingsWithFixes' .wrapped. level removed)
    emb_l.encode_with_transformers(tokens)
    emb_g.encode_with_transformers(tokens)
    assert encs[0].calls == [("clip_l_sdxl", (3, 77), (3, 77, 8))] and encs[1].calls == [("openclip2", (3, 77), (3, 77, 1280))]
    # a text-encoder LoRA on the bigG tower.  A Lora extension without network_apply_weights: its prompts go through torch, CLIP-L stays on
    # the engine ...
    lora.loaded_networks = [types.SimpleNamespace(name="style", te_multiplier=0.8, dyn_dim=None,
                                                  modules={"lora_te2_x": types.SimpleNamespace(sd_module=g_inner.model.ln_final)})]
    assert bridge.text_encoder_networks_request(emb_g, lora) == (("style", 0.8, None),) and bridge.text_encoder_networks_request(emb_l, lora) == ()
    emb_l.encode_with_transformers(tokens)
    assert emb_g.encode_with_transformers(tokens) == "torch"
    assert len(encs[0].calls) == 2 and len(encs[1].calls) == 1 and emb_g.stock_calls == [(3, 77)]
    # ... with it (extensions-builtin/Lora/networks.py:411: restore the backup, add the loaded networks' deltas IN PLACE) the deltas are merged
    # into the torch tower's weights once per change of the set, and the tower is packed again from them: the prompt stays on the engine
    applied = []

    def network_apply_weights(module):                        # what the stub merges: +1 on the layer the network names
        applied.append(module)
        if module is g_inner.model.ln_final:
            with torch.no_grad():
                module.weight.copy_(torch.full_like(module.weight, 2.0 if lora.loaded_networks else 1.0))
    lora.network_apply_weights = network_apply_weights
    g_inner.model.ln_final.network_layer_name = "1_model_ln_final"
    n_made = len(made)
    assert emb_g.encode_with_transformers(tokens).shape == (3, 77, 4) and emb_g.stock_calls == [(3, 77)]      # engine, not torch
    merged = made[-1]
    assert len(made) == n_made + 1 and applied == [g_inner.model.ln_final] and encs[1].engine.closed and emb_g._mi355x_clip is merged
    assert float(merged.sd[schema.CLIP_PREFIX + "final_layer_norm.weight"].mean()) == 2.0 and merged.calls[-1][0] == "openclip2"
    emb_g.encode_with_transformers(tokens)                    # same set of networks: no new merge, no new tower
    assert len(made) == n_made + 1 and len(applied) == 1 and len(merged.calls) == 2
    lora.loaded_networks = []                                 # the network is unloaded: weights restored through the same entry point, tower repacked
    emb_g.encode_with_transformers(tokens)
    assert len(made) == n_made + 2 and merged.engine.closed and float(made[-1].sd[schema.CLIP_PREFIX + "final_layer_norm.weight"].mean()) == 1.0
    emb_l.encode_with_transformers(tokens)                    # the CLIP-L tower was never touched
    assert emb_l._mi355x_clip is encs[0] and not encs[0].engine.closed
    # installing again replaces the towers (checkpoint reload), uninstall restores both wrappers
    current = [emb_l._mi355x_clip, emb_g._mi355x_clip]
    encs2 = bridge.install_clip_hook(sdxl, lora_networks=lora)
    assert all(e.engine.closed for e in current) and encs2[0] is not encs[0]
    bridge.uninstall_clip_hook(sdxl)
    assert emb_l.encode_with_transformers(tokens) == "torch" and all(e.engine.closed for e in encs2)
