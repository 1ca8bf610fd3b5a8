#!/usr/bin/env python3
"""Generate golden fixtures from the importable pieces of the reference (run in the authoring container,
where /root/reference exists; the GPU box only sees the committed .npz files).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Reference files executed — loaded by path, unmodified, with the webui modules they import stubbed; where a file cannot be
imported at all (lark / gradio / ldm at module level) the named functions are exec'd from the file's own text:
  modules/rng_philox.py                          -> philox.npz
  modules/rng.py (+ rng_philox.py)               -> image_rng.npz          (ImageRNG: variation seeds, seed-resize, ENSD)
  modules/sub_quadratic_attention.py             -> subquad_attention.npz
  modules/models/sd3/sd3_impls.py                -> vae_decoder.npz, vae_encoder.npz   (VAEDecoder / VAEEncoder, z_channels=4)
  modules/models/sd3/sd3_impls.py (sample_euler) -> euler_twin.npz         (in-tree copy of k-diffusion's Euler sampler)
  modules/models/sd3/other_impls.py              -> clip_text.npz          (CLIP text transformer)
  modules/sd_samplers_timesteps_impl.py          -> ddim.npz, plms.npz     (ddim, ddim_cfgpp, plms)
  ... + modules/models/diffusion/uni_pc/uni_pc.py -> unipc.npz              (unipc() / UniPCCFG over the real solver)
  modules/sd_samplers_extra.py                   -> restart.npz
  modules/sd_samplers_lcm.py                     -> lcm.npz
  modules/sd_schedulers.py                       -> schedulers.npz
  modules/sd_samplers_cfg_denoiser.py            -> cfg_denoiser.npz       (CFGDenoiser.forward, 20 scenarios)
  modules/sd_models.py           [text]          -> zsnr.npz               (rescale_zero_terminal_snr_abar)
  modules/sd_samplers_common.py  [text]          -> refiner.npz            (apply_refiner), + images_tensor_to_samples below
  modules/processing.py          [text]          -> image_conditioning.npz (txt2img / inpainting / edit image conditioning)
  modules/images.py, upscaler.py [text]          -> resize_image.npz       (resize_image mode 0, Upscaler loop, PIL scalers)
  modules/prompt_parser.py       [text]          -> prompt_cond.npz/.json  (AND splitting, reconstruct_*cond_batch)
  modules/sd_hijack_unet.py, hypernetworks/hypernetwork.py [text] -> unet_twins.npz (timestep embedding, SpatialTransformer and
                                                    baseline attention forwards run on the oracle's modules)
  extensions-builtin/Lora/networks.py [text]     -> lora_names.json        (convert_diffusers_name_to_compvis)
  extensions-builtin/Lora/network*.py, lyco_helpers.py -> lyco.npz         (calc_updown of every module type but OFT)
Weights / inputs are produced by ``seeded()`` below (CPU torch.Generator, N(0,1) scaled), so a fixture stores
only seeds + the reference's outputs; tests regenerate the same inputs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("SD_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def seeded(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def seeded_module_weights(module, seed):
    """Fill every parameter of ``module`` (in state_dict order) with seeded values; returns nothing."""
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


def gen_philox():
    m = load_by_path("ref_rng_philox", "modules/rng_philox.py")
    out = {}
    cases = [(0, (3, 4), 1), (1000, (4, 8, 8), 3), (1007, (4, 16, 16), 2), (2 ** 32 + 5, (2, 5), 2), (123456789, (1, 4, 64, 64), 1)]
    for ci, (seed, shape, draws) in enumerate(cases):
        g = m.Generator(seed)
        for d in range(draws):
            out[f"c{ci}_seed{seed}_draw{d}"] = g.randn(shape)
    out["docstring_seed0"] = np.array([[-0.92466259, -0.42534415, -2.6438457, 0.14518388],
                                       [-0.12086647, -0.57972564, -0.62285122, -0.32838709],
                                       [-1.07454231, -0.36314407, -1.67105067, 2.26550497]], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "philox.npz"), **out)
    print("philox.npz", len(out))


def gen_subquad():
    m = load_by_path("ref_subquad", "modules/sub_quadratic_attention.py")
    out = {}
    for ci, (bh, n, mk, d, qc, kc) in enumerate([(4, 96, 96, 40, 32, 32), (2, 64, 77, 80, 64, 16), (3, 50, 130, 160, 16, 64)]):
        q, k, v = seeded((bh, n, d), 10 + ci), seeded((bh, mk, d), 20 + ci), seeded((bh, mk, d), 30 + ci)
        o = m.efficient_dot_product_attention(q, k, v, query_chunk_size=qc, kv_chunk_size=kc, use_checkpoint=False)
        out[f"c{ci}_shape"] = np.array([bh, n, mk, d])
        out[f"c{ci}_out"] = o.numpy()
    np.savez_compressed(os.path.join(OUT, "subquad_attention.npz"), **out)
    print("subquad_attention.npz")


def gen_vae():
    stub = types.ModuleType("modules.models.sd3.mmdit")
    stub.MMDiT = object
    for name in ("modules", "modules.models", "modules.models.sd3"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["modules.models.sd3.mmdit"] = stub
    m = load_by_path("ref_sd3_impls", "modules/models/sd3/sd3_impls.py")
    with torch.no_grad():
        # full-size SD1.5 decoder (49,490,179 params) on an 8x8 latent, and a small one on 16x16
        dec = m.VAEDecoder(z_channels=4)
        seeded_module_weights(dec, 777)
        z = seeded((1, 4, 8, 8), 778)
        out = {"full_nparams": np.array(sum(p.numel() for p in dec.parameters())),
               "full_out": dec(z).numpy()}
        dec_s = m.VAEDecoder(ch=64, ch_mult=(1, 2), num_res_blocks=1, z_channels=4)
        seeded_module_weights(dec_s, 779)
        z = seeded((2, 4, 16, 16), 780)
        out["small_out"] = dec_s(z).numpy()
        np.savez_compressed(os.path.join(OUT, "vae_decoder.npz"), **out)
        enc = m.VAEEncoder(ch=64, ch_mult=(1, 2), num_res_blocks=1, z_channels=4)
        seeded_module_weights(enc, 781)
        x = seeded((2, 3, 32, 32), 782)
        np.savez_compressed(os.path.join(OUT, "vae_encoder.npz"), small_out=enc(x).numpy())
    print("vae_decoder.npz vae_encoder.npz")


def gen_vae_512():
    """The reference's own full-size VAEDecoder at the BENCH shape (64x64 latent -> 512x512 image); the fixture keeps every 4th
    pixel of each axis (3 x 128 x 128 fp32, ~190 KB) — tests/test_gpu_c1_parity.py compares the same sub-lattice."""
    stub = types.ModuleType("modules.models.sd3.mmdit")
    stub.MMDiT = object
    for name in ("modules", "modules.models", "modules.models.sd3"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["modules.models.sd3.mmdit"] = stub
    m = load_by_path("ref_sd3_impls", "modules/models/sd3/sd3_impls.py")
    with torch.no_grad():
        dec = m.VAEDecoder(z_channels=4)
        seeded_module_weights(dec, 777)
        out = dec(seeded((1, 4, 64, 64), 778))
        np.savez_compressed(os.path.join(OUT, "vae_decoder_512.npz"), full_out_512_sub4=out[:, :, ::4, ::4].contiguous().numpy(),
                            full_out_512_mean=np.array(float(out.mean())), full_out_512_std=np.array(float(out.std())))
    print("vae_decoder_512.npz")


def gen_ddim():
    # stub the modules sd_samplers_timesteps_impl imports (k_diffusion is third-party and absent here)
    kd = types.ModuleType("k_diffusion")
    kds = types.ModuleType("k_diffusion.sampling")
    kd.sampling = kds
    sys.modules["k_diffusion"] = kd
    sys.modules["k_diffusion.sampling"] = kds
    for name in ("modules", "modules.models", "modules.models.diffusion"):
        sys.modules.setdefault(name, types.ModuleType(name))
    shared = types.ModuleType("modules.shared")
    sys.modules["modules.shared"] = shared
    sys.modules["modules"].shared = shared
    unipc_pkg = types.ModuleType("modules.models.diffusion.uni_pc")
    unipc_pkg.uni_pc = types.ModuleType("modules.models.diffusion.uni_pc.uni_pc")
    unipc_pkg.uni_pc.UniPC = object          # only subclassed at import time, never used by ddim()
    sys.modules["modules.models.diffusion.uni_pc"] = unipc_pkg
    tu = load_by_path("modules.torch_utils", "modules/torch_utils.py")
    sys.modules["modules"].torch_utils = tu
    try:
        import tqdm  # noqa: F401
    except ImportError:
        t = types.ModuleType("tqdm")
        t.trange = lambda n, disable=None: range(n)
        sys.modules["tqdm"] = t
    impl = load_by_path("ref_timesteps_impl", "modules/sd_samplers_timesteps_impl.py")

    betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, 1000, dtype=torch.float64) ** 2
    alphas_cumprod = torch.tensor(np.cumprod(1.0 - betas.numpy(), axis=0), dtype=torch.float32)

    class Inner2:
        pass

    class Inner1:
        pass

    class Model:
        """model.inner_model.inner_model.alphas_cumprod as in the reference; a fixed analytic eps function."""
        def __init__(self):
            self.inner_model = Inner1()
            self.inner_model.inner_model = Inner2()
            self.inner_model.inner_model.alphas_cumprod = alphas_cumprod

        def __call__(self, x, t, **kw):
            return torch.tanh(0.7 * x + (t / 1000.0)[:, None, None, None]) * 0.9 + 0.05 * x

    out = {}
    for ci, (steps, eta) in enumerate([(20, 0.0), (7, 0.5)]):
        noise_draws = [seeded((2, 4, 8, 8), 900 + i) for i in range(steps + 2)]
        it = iter(noise_draws)

        class TH:
            @staticmethod
            def randn_like(x):
                return next(it)
        kds.torch = TH
        timesteps = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
        x0 = seeded((2, 4, 8, 8), 890 + ci)
        res = impl.ddim(Model(), x0.clone(), timesteps, extra_args={}, disable=True, eta=eta)
        out[f"c{ci}_steps_eta"] = np.array([steps, eta])
        out[f"c{ci}_out"] = res.numpy()
    # DDIM CFG++ (same file, :43-82): the model object also exposes last_noise_uncond
    class ModelPP(Model):
        def __call__(self, x, t, **kw):
            self.last_noise_uncond = torch.sin(0.3 * x) * 0.8 - 0.1 * (t / 1000.0)[:, None, None, None]
            return super().__call__(x, t, **kw)
    for ci, (steps, eta) in enumerate([(12, 0.0), (9, 0.7)]):
        noise_draws = [seeded((2, 4, 8, 8), 950 + i) for i in range(steps + 2)]
        it = iter(noise_draws)

        class TH2:
            @staticmethod
            def randn_like(x):
                return next(it)
        kds.torch = TH2
        timesteps = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
        mpp = ModelPP()
        res = impl.ddim_cfgpp(mpp, seeded((2, 4, 8, 8), 940 + ci), timesteps, extra_args={}, disable=True, eta=eta)
        assert mpp.cond_scale_miltiplier == 1 / 12.5 and mpp.need_last_noise_uncond is True
        out[f"cfgpp{ci}_steps_eta"] = np.array([steps, eta])
        out[f"cfgpp{ci}_out"] = res.numpy()
    np.savez_compressed(os.path.join(OUT, "ddim.npz"), **out)
    print("ddim.npz")
    # PLMS (same file, :85-137): deterministic, no noise
    out = {}
    for ci, steps in enumerate([20, 6]):
        timesteps = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
        x0 = seeded((2, 4, 8, 8), 870 + ci)
        res = impl.plms(Model(), x0.clone(), timesteps, extra_args={}, disable=True)
        out[f"c{ci}_steps"] = np.array([steps])
        out[f"c{ci}_out"] = res.numpy()
    np.savez_compressed(os.path.join(OUT, "plms.npz"), **out)
    print("plms.npz")


def gen_schedulers():
    """Execute modules/sd_schedulers.py.  Its only third-party dependency is the k-diffusion denoiser object passed in as
    ``inner_model`` (sigmas / sigma_to_t / t_to_sigma / get_sigmas) and k_diffusion.sampling's three get_sigmas_* functions
    referenced by the table; the oracle's restatements stand in for both (they are what the fixture is NOT pinning)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import kdiffusion as okd, schedulers as osch
    kd = types.ModuleType("k_diffusion")
    kds = types.ModuleType("k_diffusion.sampling")
    kds.get_sigmas_karras = lambda n, sigma_min, sigma_max, rho=7.0, device="cpu": okd.get_sigmas_karras(n, sigma_min, sigma_max, rho)
    kds.get_sigmas_exponential = lambda n, sigma_min, sigma_max, device="cpu": osch.get_sigmas_exponential(n, sigma_min, sigma_max)
    kds.get_sigmas_polyexponential = lambda n, sigma_min, sigma_max, rho=1.0, device="cpu": osch.get_sigmas_polyexponential(n, sigma_min, sigma_max, rho)
    kd.sampling = kds
    sys.modules["k_diffusion"] = kd
    sys.modules["k_diffusion.sampling"] = kds
    sys.modules.setdefault("modules", types.ModuleType("modules"))
    shared = types.ModuleType("modules.shared")

    class Opts:
        beta_dist_alpha = 0.6
        beta_dist_beta = 0.6

    class SdModel:
        is_sdxl = False
    shared.opts, shared.sd_model = Opts(), SdModel()
    sys.modules["modules.shared"] = shared
    sys.modules["modules"].shared = shared
    ref = load_by_path("ref_sd_schedulers", "modules/sd_schedulers.py")
    inner = okd.CompVisDenoiser(None, okd.make_alphas_cumprod())
    smin, smax = inner.sigmas[0].item(), inner.sigmas[-1].item()
    out = {"names": np.array([s.name for s in ref.schedulers]), "labels": np.array([s.label for s in ref.schedulers]),
           "default_rho": np.array([s.default_rho for s in ref.schedulers]),
           "need_inner_model": np.array([s.need_inner_model for s in ref.schedulers])}
    for n in (5, 11, 20, 50):
        for sch in ref.schedulers:
            if sch.function is None or sch.name in ("karras", "exponential", "polyexponential"):
                continue
            if sch.need_inner_model:
                sig = sch.function(n, smin, smax, inner, "cpu")
            else:
                sig = sch.function(n, smin, smax, "cpu")
            out[f"{sch.name}_{n}"] = torch.as_tensor(sig).float().numpy()
    SdModel.is_sdxl = True
    out["align_your_steps_sdxl_20"] = ref.get_align_your_steps_sigmas(20, smin, smax, "cpu").numpy()
    np.savez_compressed(os.path.join(OUT, "schedulers.npz"), **out)
    print("schedulers.npz")


def gen_clip():
    """Execute the reference's plain-torch CLIP text model (modules/models/sd3/other_impls.py:61-150) on a small configuration
    with seeded weights: last_hidden_state, the penultimate hidden state with the final norm (clip skip 2), pooled output; both
    activations.  Its two imports from outside torch are stubbed: transformers' tokenizers (unused) and
    sd_hijack.TextualInversionEmbeddings (a plain nn.Embedding when no embedding is registered)."""
    tr = types.ModuleType("transformers")
    tr.CLIPTokenizer = object
    tr.T5TokenizerFast = object
    sys.modules["transformers"] = tr
    sys.modules.setdefault("modules", types.ModuleType("modules"))
    hij = types.ModuleType("modules.sd_hijack")

    class TextualInversionEmbeddings(torch.nn.Embedding):
        def __init__(self, num_embeddings, embedding_dim, textual_inversion_key='clip_l', **kwargs):
            super().__init__(num_embeddings, embedding_dim, **kwargs)
    hij.TextualInversionEmbeddings = TextualInversionEmbeddings
    sys.modules["modules.sd_hijack"] = hij
    sys.modules["modules"].sd_hijack = hij
    oi = load_by_path("ref_other_impls", "modules/models/sd3/other_impls.py")
    out = {}
    with torch.no_grad():
        for ci, act in enumerate(("quick_gelu", "gelu")):
            cfgd = {"num_hidden_layers": 3, "hidden_size": 128, "num_attention_heads": 2, "intermediate_size": 256, "hidden_act": act}
            m = oi.CLIPTextModel_(cfgd, torch.float32, "cpu")
            # vocab is fixed at 49408 in the reference class; seed every parameter in state-dict order
            g = torch.Generator().manual_seed(4000 + ci)
            for name, prm in m.state_dict().items():
                if name.endswith("embedding.weight"):
                    prm.copy_(torch.randn(prm.shape, generator=g) * 0.5)
                elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("layer_norm.weight"):
                    prm.copy_(1.0 + 0.02 * torch.randn(prm.shape, generator=g))
                elif name.endswith(".bias"):
                    prm.copy_(0.02 * torch.randn(prm.shape, generator=g))
                else:
                    prm.copy_(torch.randn(prm.shape, generator=g) * prm.shape[-1] ** -0.5)
            tok = torch.randint(0, 49407, (2, 77), generator=torch.Generator().manual_seed(4100 + ci))
            tok[:, 0] = 49406
            tok[0, 20:] = 49407
            tok[1, 50:] = 49407
            x, inter, pooled = m(tok.clone(), intermediate_output=-2)
            out[f"c{ci}_tokens"] = tok.numpy()
            out[f"c{ci}_last"] = x.numpy()
            out[f"c{ci}_skip2"] = inter.numpy()
            out[f"c{ci}_pooled"] = pooled.numpy()
    np.savez_compressed(os.path.join(OUT, "clip_text.npz"), **out)
    print("clip_text.npz")


def gen_restart():
    """Execute modules/sd_samplers_extra.py restart_sampler on the analytic denoiser; k_diffusion.sampling is stubbed with the
    oracle's to_d / get_sigmas_karras (third-party functions the fixture does not pin) and a scripted randn_like."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import kdiffusion as okd
    kd = types.ModuleType("k_diffusion")
    kds = types.ModuleType("k_diffusion.sampling")
    kds.to_d = okd.to_d
    kds.get_sigmas_karras = lambda n, sigma_min, sigma_max, rho=7.0, device="cpu": okd.get_sigmas_karras(n, sigma_min, sigma_max, rho)
    kd.sampling = kds
    sys.modules["k_diffusion"] = kd
    sys.modules["k_diffusion.sampling"] = kds
    try:
        import tqdm  # noqa: F401
    except ImportError:
        t = types.ModuleType("tqdm")
        t.tqdm = lambda it, disable=None: it
        sys.modules["tqdm"] = t
    ref = load_by_path("ref_samplers_extra", "modules/sd_samplers_extra.py")
    den = okd.CompVisDenoiser(None, okd.make_alphas_cumprod())
    smin, smax = den.sigmas[0].item(), den.sigmas[-1].item()

    def model(x, sigma, **kw):                              # a smooth "denoiser": pulls x toward tanh(x) more as sigma shrinks
        s = sigma[:, None, None, None]
        return x / (1 + s * s) + torch.tanh(0.5 * x) * (s * s / (1 + s * s)) * 0.3

    out = {}
    for ci, steps in enumerate([8, 22, 40]):
        draws = iter([seeded((2, 4, 8, 8), 3000 + 10 * ci + i) for i in range(8)])

        class TH:
            @staticmethod
            def randn_like(x):
                return next(draws)
        kds.torch = TH
        sigmas = okd.get_sigmas_karras(steps, smin, smax)
        x0 = seeded((2, 4, 8, 8), 2990 + ci) * sigmas[0]
        res = ref.restart_sampler(model, x0.clone(), sigmas, extra_args={}, disable=True)
        out[f"c{ci}_steps"] = np.array([steps])
        out[f"c{ci}_out"] = res.numpy()
    np.savez_compressed(os.path.join(OUT, "restart.npz"), **out)
    print("restart.npz")


def gen_lora_names():
    """Execute convert_diffusers_name_to_compvis (extensions-builtin/Lora/networks.py:40-120) on kohya-style LoRA keys of
    every UNet layer family.  The module imports the whole webui, so only the function and the three module-level objects it
    uses are exec'd from the file's own text (nothing is copied into this repository)."""
    import json
    import re
    src = open(os.path.join(REF, "extensions-builtin/Lora/networks.py")).read()
    a = src.index("re_digits = re.compile")
    b = src.index("def assign_network_names_to_compvis_modules")
    ns = {"re": re}
    exec(src[a:b], ns)
    conv = ns["convert_diffusers_name_to_compvis"]
    keys = ["lora_unet_conv_in", "lora_unet_conv_out", "lora_unet_time_embedding_linear_1", "lora_unet_time_embedding_linear_2"]
    for blk in range(4):
        for j in range(2):
            for sfx in ("conv1", "conv2", "time_emb_proj", "conv_shortcut", "norm1"):
                keys.append(f"lora_unet_down_blocks_{blk}_resnets_{j}_{sfx}")
            for sfx in ("proj_in", "proj_out", "transformer_blocks_0_attn1_to_q", "transformer_blocks_0_attn2_to_k",
                        "transformer_blocks_0_attn1_to_out_0", "transformer_blocks_0_ff_net_0_proj", "transformer_blocks_0_ff_net_2"):
                keys.append(f"lora_unet_down_blocks_{blk}_attentions_{j}_{sfx}")
        keys.append(f"lora_unet_down_blocks_{blk}_downsamplers_0_conv")
        for j in range(3):
            keys.append(f"lora_unet_up_blocks_{blk}_resnets_{j}_conv1")
            keys.append(f"lora_unet_up_blocks_{blk}_attentions_{j}_transformer_blocks_0_attn2_to_v")
        keys.append(f"lora_unet_up_blocks_{blk}_upsamplers_0_conv")
    keys += ["lora_unet_mid_block_resnets_0_conv1", "lora_unet_mid_block_resnets_1_conv2", "lora_unet_mid_block_attentions_0_proj_in",
             "lora_unet_mid_block_attentions_0_transformer_blocks_0_attn1_to_k",
             "lora_te_text_model_encoder_layers_3_self_attn_q_proj", "lora_te_text_model_encoder_layers_11_mlp_fc1",
             "lora_te2_text_model_encoder_layers_5_mlp_fc2", "lora_unet_input_blocks_4_1_transformer_blocks_1_attn1_to_q",
             "something_else_entirely"]
    out = {"sd1": {k: conv(k, False) for k in keys}, "sd2": {k: conv(k, True) for k in keys if "lora_te_" in k}}
    json.dump(out, open(os.path.join(OUT, "lora_names.json"), "w"), indent=0, sort_keys=True)
    print("lora_names.json", len(keys))


def gen_unipc():
    """modules/sd_samplers_timesteps_impl.py:144-190 (UniPCCFG + unipc()) over the real modules/models/diffusion/uni_pc/uni_pc.py,
    with a fixed analytic eps model; covers the three skip types, both B(h) variants and vary_coeff, orders 1-4, lower_order_final off
    and the img2img start."""
    kd = types.ModuleType("k_diffusion")
    kds = types.ModuleType("k_diffusion.sampling")
    kd.sampling = kds
    sys.modules["k_diffusion"] = kd
    sys.modules["k_diffusion.sampling"] = kds
    for name in ("modules", "modules.models", "modules.models.diffusion"):
        sys.modules.setdefault(name, types.ModuleType(name))
    shared = types.ModuleType("modules.shared")
    shared.opts = types.SimpleNamespace()
    sys.modules["modules.shared"] = shared
    sys.modules["modules"].shared = shared
    real = load_by_path("modules.models.diffusion.uni_pc.uni_pc", "modules/models/diffusion/uni_pc/uni_pc.py")
    unipc_pkg = types.ModuleType("modules.models.diffusion.uni_pc")
    unipc_pkg.uni_pc = real
    sys.modules["modules.models.diffusion.uni_pc"] = unipc_pkg
    tu = load_by_path("modules.torch_utils", "modules/torch_utils.py")
    sys.modules["modules"].torch_utils = tu
    impl = load_by_path("ref_timesteps_impl_unipc", "modules/sd_samplers_timesteps_impl.py")

    betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, 1000, dtype=torch.float64) ** 2
    alphas_cumprod = torch.tensor(np.cumprod(1.0 - betas.numpy(), axis=0), dtype=torch.float32)

    class Model:
        def __init__(self):
            self.inner_model = types.SimpleNamespace(inner_model=types.SimpleNamespace(alphas_cumprod=alphas_cumprod))
            self.ts = []

        def __call__(self, x, t, **kw):
            self.ts.append(float(t[0]))
            return torch.tanh(0.7 * x + (t / 1000.0)[:, None, None, None]) * 0.9 + 0.05 * x

    cases = [  # steps, variant, skip_type, order, lower_order_final, is_img2img(t_enc)
        (20, "bh1", "time_uniform", 3, True, 0),
        (8, "bh2", "time_uniform", 3, True, 0),
        (10, "bh1", "time_quadratic", 2, True, 0),
        (9, "bh2", "logSNR", 3, False, 0),
        (6, "bh1", "time_uniform", 1, True, 0),
        (12, "bh1", "time_uniform", 4, True, 0),
        (20, "bh1", "time_uniform", 3, True, 11),
        # vary_coeff: the reference's update multiplies the [B] schedule vectors into x without expanding them (uni_pc.py:584), so it
        # only runs at batch 1 (it raises a broadcasting error otherwise): pinned at batch 1
        (10, "vary_coeff", "time_uniform", 3, True, 0),
        (8, "vary_coeff", "time_quadratic", 2, True, 0),
        (7, "vary_coeff", "logSNR", 1, True, 0),
        (9, "vary_coeff", "time_uniform", 4, False, 0),
    ]
    out = {}
    for ci, (steps, variant, skip, order, lof, t_enc) in enumerate(cases):
        shared.opts.uni_pc_variant, shared.opts.uni_pc_skip_type = variant, skip
        shared.opts.uni_pc_order, shared.opts.uni_pc_lower_order_final = order, lof
        timesteps = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
        if t_enc:
            timesteps = timesteps[:t_enc]
        m = Model()
        dens = []
        batch = 1 if variant == "vary_coeff" else 2
        out[f"c{ci}_batch"] = np.array([batch])
        res = impl.unipc(m, seeded((batch, 4, 8, 8), 990 + ci), timesteps, extra_args={}, disable=True,
                         callback=lambda d: dens.append(None if d['denoised'] is None else d['denoised'].clone()),
                         is_img2img=bool(t_enc))
        out[f"c{ci}_cfg"] = np.array([steps, order, int(lof), t_enc])
        out[f"c{ci}_variant_skip"] = np.array([variant, skip])
        out[f"c{ci}_out"] = res.numpy()
        out[f"c{ci}_model_t"] = np.array(m.ts, dtype=np.float64)
        assert dens[-1] is None          # the final update runs without corrector: callback gets denoised=None (uni_pc.py:775-789)
        out[f"c{ci}_last_denoised"] = dens[-2].numpy()
        out[f"c{ci}_n_callbacks"] = np.array([len(dens)])
    np.savez_compressed(os.path.join(OUT, "unipc.npz"), **out)
    print("unipc.npz")


def gen_lcm():
    """Execute modules/sd_samplers_lcm.py (LCMCompVisDenoiser :10-63, sample_lcm :66-80).  k_diffusion is third-party and absent:
    its DiscreteEpsDDPMDenoiser base / append_dims / append_zero are supplied from the oracle's restatement of k-diffusion
    (not pinned by this fixture); everything the reference file itself adds — the 50-entry sigma table, its get_sigmas /
    sigma_to_t / t_to_sigma, the consistency-model output scaling and the sampling loop — is."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import kdiffusion as okd
    import collections

    class DiscreteEpsDDPMDenoiser(okd.DiscreteSchedule):
        def __init__(self, model, alphas_cumprod, quantize):
            super().__init__(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5, quantize)
            self.inner_model = model
            self.sigma_data = 1.

        def get_scalings(self, sigma):
            return -sigma, 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5

        def __call__(self, *a, **kw):
            return self.forward(*a, **kw)

    kd = types.ModuleType("k_diffusion")
    kds = types.ModuleType("k_diffusion.sampling")
    kdu = types.ModuleType("k_diffusion.utils")
    kde = types.ModuleType("k_diffusion.external")
    kdu.append_dims = okd.append_dims
    kds.append_zero = okd.append_zero
    kds.trange = lambda n, disable=None: range(n)
    kds.default_noise_sampler = lambda x: None
    kde.DiscreteEpsDDPMDenoiser = DiscreteEpsDDPMDenoiser
    kd.sampling, kd.utils, kd.external = kds, kdu, kde
    for n, m in (("k_diffusion", kd), ("k_diffusion.sampling", kds), ("k_diffusion.utils", kdu), ("k_diffusion.external", kde)):
        sys.modules[n] = m
    mods = sys.modules.setdefault("modules", types.ModuleType("modules"))
    ac = okd.make_alphas_cumprod()

    class SdModel:
        device = "cpu"
        alphas_cumprod = ac

        @staticmethod
        def apply_model(x, t, **kw):
            return torch.tanh(0.6 * x + (t.float() / 1000.0)[:, None, None, None]) * 0.8 + 0.1 * x

    shared = types.ModuleType("modules.shared")
    shared.sd_model = SdModel()
    cfgd = types.ModuleType("modules.sd_samplers_cfg_denoiser")
    cfgd.CFGDenoiser = type("CFGDenoiser", (), {})
    kdiff = types.ModuleType("modules.sd_samplers_kdiffusion")
    kdiff.KDiffusionSampler = type("KDiffusionSampler", (), {})
    common = types.ModuleType("modules.sd_samplers_common")
    common.SamplerData = collections.namedtuple('SamplerData', ['name', 'constructor', 'aliases', 'options'])
    for n, m in (("shared", shared), ("sd_samplers_cfg_denoiser", cfgd), ("sd_samplers_kdiffusion", kdiff), ("sd_samplers_common", common)):
        sys.modules["modules." + n] = m
        setattr(mods, n, m)
    ref = load_by_path("ref_samplers_lcm", "modules/sd_samplers_lcm.py")
    den = ref.LCMCompVisDenoiser(shared.sd_model)
    out = {"sigmas": den.sigmas.numpy(), "get_sigmas_all": den.get_sigmas().numpy()}
    probe = torch.tensor([0.03, 0.5, 1.7, 3.3, 14.6, 20.0])
    out["probe_sigma"] = probe.numpy()
    out["probe_t"] = den.sigma_to_t(probe).numpy()
    out["probe_t_to_sigma"] = den.t_to_sigma(torch.tensor([0., 19., 59., 333., 500.5, 999., 1200.])).numpy()
    x = seeded((2, 4, 8, 8), 4100)
    for k, sg in enumerate([14.6, 2.2, 0.4]):
        out[f"forward{k}"] = den(x * sg, torch.full((2,), sg)).numpy()
    for ci, steps in enumerate([4, 8]):
        sigmas = den.get_sigmas(steps)
        out[f"c{ci}_sigmas"] = sigmas.numpy()
        draws = iter([seeded((2, 4, 8, 8), 4200 + 10 * ci + i) for i in range(steps)])
        res = ref.sample_lcm(den, seeded((2, 4, 8, 8), 4190 + ci) * sigmas[0], sigmas, extra_args={}, disable=True,
                             noise_sampler=lambda a, b: next(draws))
        out[f"c{ci}_out"] = res.numpy()
    assert [x.name for x in ref.samplers_data_lcm] == ["LCM"] and ref.samplers_lcm[0][2] == ['k_lcm'] and ref.samplers_lcm[0][3] == {}
    np.savez_compressed(os.path.join(OUT, "lcm.npz"), **out)
    print("lcm.npz")


CFG_SCENARIOS = [
    # name, dict(b, conds_list | None, t_cond, t_uncond, cond_scale, s_min_uncond, sigma, step, total_steps, opts overrides, flags)
    ("plain", dict()),
    ("and", dict(conds_list=[[(0, 1.0), (1, 0.6)], [(2, 0.8)]])),
    ("ngms_odd", dict(s_min_uncond=5.0, sigma=2.0, step=1)),
    ("ngms_even", dict(s_min_uncond=5.0, sigma=2.0, step=2)),
    ("ngms_all", dict(s_min_uncond=5.0, sigma=2.0, step=2, opts=dict(s_min_uncond_all=True))),
    ("ngms_high_sigma", dict(s_min_uncond=1.0, sigma=2.0, step=1)),
    ("skip_early", dict(step=1, total_steps=10, opts=dict(skip_early_cond=0.3))),
    ("skip_early_and", dict(step=0, total_steps=10, opts=dict(skip_early_cond=0.3), conds_list=[[(0, 1.0), (1, 0.6)], [(2, 0.8)]])),
    ("mask_after", dict(mask=True)),
    ("mask_before", dict(mask=True, mask_before=True)),
    ("long_cond", dict(t_cond=16, t_uncond=8)),
    ("long_uncond", dict(t_cond=8, t_uncond=24)),
    ("long_cond_pad", dict(t_cond=16, t_uncond=8, opts=dict(pad_cond_uncond=True))),
    ("long_uncond_pad", dict(t_cond=8, t_uncond=24, opts=dict(pad_cond_uncond=True))),
    ("long_cond_pad_v0", dict(t_cond=16, t_uncond=8, opts=dict(pad_cond_uncond_v0=True))),
    ("long_uncond_pad_v0", dict(t_cond=8, t_uncond=24, opts=dict(pad_cond_uncond_v0=True))),
    ("cfgpp", dict(need_last_noise_uncond=True, cond_scale_miltiplier=1 / 12.5)),
    ("no_batch", dict(opts=dict(batch_cond_uncond=False))),
    ("edit", dict(edit=True, image_cfg_scale=1.5)),
    ("edit_scale_1", dict(edit=True, image_cfg_scale=1.0, conds_list=[[(0, 1.0), (1, 0.6)], [(2, 0.8)]])),
    # unCLIP checkpoints (conditioning_key "crossattn-adm", :192-194): image_cond is the CLIP image embedding [B, adm], handed to the
    # UNet as c_adm; the uncond rows get zeros
    ("unclip", dict(adm=True)),
    ("unclip_and_ngms", dict(adm=True, conds_list=[[(0, 1.0), (1, 0.6)], [(2, 0.8)]], s_min_uncond=5.0, sigma=2.0, step=1)),
]


def cfg_scenario_inputs(k, sc):
    """Seeded inputs of CFG scenario k (shared by the generator and the tests)."""
    b, c, hw, dim = 2, 4, 8, 6
    conds_list = sc.get("conds_list") or [[(i, 1.0)] for i in range(b)]
    n_cond = sum(len(x) for x in conds_list)
    return dict(
        x=seeded((b, c, hw, hw), 5000 + 10 * k), conds_list=conds_list,
        cond=seeded((n_cond, sc.get("t_cond", 8), dim), 5001 + 10 * k, 0.5),
        uncond=seeded((b, sc.get("t_uncond", 8), dim), 5002 + 10 * k, 0.5),
        image_cond=seeded((b, 12), 5003 + 10 * k) if sc.get("adm") else seeded((b, 4 if sc.get("edit") else 5, hw, hw), 5003 + 10 * k),
        init_latent=seeded((b, c, hw, hw), 5004 + 10 * k),
        mask=(seeded((b, 1, hw, hw), 5005 + 10 * k) > 0).float(), empty=seeded((1, 8, dim), 4999, 0.5),
        sigma=torch.full((b,), float(sc.get("sigma", 3.0))))


def cfg_inner_model(x_in, sigma_in, c_crossattn, c_concat):
    """Analytic stand-in for the wrapped UNet: depends on every input, and on the token COUNT of the context."""
    ctx = c_crossattn.sum(dim=(1, 2))[:, None, None, None]
    # (the reference leaves c_concat at full length when it drops the uncond rows, :212-214 — only the matching rows are read)
    if c_concat.dim() == 2:                               # unCLIP: c_adm [rows, adm] — a per-row vector, as the UNet's y
        return torch.tanh(0.5 * x_in + 0.2 * ctx + 0.1 * sigma_in[:, None, None, None]) + 0.05 * c_concat[:x_in.shape[0]].sum(1)[:, None, None, None] * x_in
    return torch.tanh(0.5 * x_in + 0.2 * ctx + 0.1 * sigma_in[:, None, None, None]) + 0.05 * c_concat[:x_in.shape[0], :4] * x_in


def gen_cfg_denoiser():
    """Execute CFGDenoiser.forward (modules/sd_samplers_cfg_denoiser.py:156-311) with the webui modules it imports stubbed
    (prompt_parser hands back the (conds_list, tensor) pair it is given; script callbacks are no-ops) over CFG_SCENARIOS:
    plain CFG, AND composition, skip-uncond (NGMS / skip-early), inpainting mask before / after, cond / uncond of different
    token counts with and without the two padding options, CFG++ bookkeeping, unbatched cond / uncond, InstructPix2Pix."""
    mods = sys.modules.setdefault("modules", types.ModuleType("modules"))
    pp = types.ModuleType("modules.prompt_parser")
    pp.reconstruct_multicond_batch = lambda c, step: c
    pp.reconstruct_cond_batch = lambda c, step: c
    common = types.ModuleType("modules.sd_samplers_common")
    common.InterruptedException = type("InterruptedException", (BaseException,), {})
    common.apply_refiner = lambda cfg, sigma=None: False
    common.store_latent = lambda x: None
    shared = types.ModuleType("modules.shared")
    shared.state = types.SimpleNamespace(interrupted=False, skipped=False, sampling_step=0, sampling_steps=20)
    cb = types.ModuleType("modules.script_callbacks")

    class _Params:
        def __init__(self, *a, **kw):
            names = {8: ["x", "image_cond", "sigma", "sampling_step", "total_sampling_steps", "text_cond", "text_uncond", "denoiser"],
                     4: ["x", "sampling_step", "total_sampling_steps", "inner_model"], 3: ["x", "sampling_step", "total_sampling_steps"]}[len(a)]
            for n, v in zip(names, a):
                setattr(self, n, v)
    cb.CFGDenoiserParams = cb.CFGDenoisedParams = cb.AfterCFGCallbackParams = _Params
    cb.cfg_denoiser_callback = cb.cfg_denoised_callback = cb.cfg_after_cfg_callback = lambda params: None
    for n, m in (("prompt_parser", pp), ("sd_samplers_common", common), ("shared", shared), ("script_callbacks", cb)):
        sys.modules["modules." + n] = m
        setattr(mods, n, m)
    out = {}
    for k, (name, sc) in enumerate(CFG_SCENARIOS):
        base = dict(skip_early_cond=0.0, s_min_uncond_all=False, pad_cond_uncond=False, pad_cond_uncond_v0=False,
                    batch_cond_uncond=True, live_preview_content="Prompt")
        base.update(sc.get("opts", {}))
        shared.opts = types.SimpleNamespace(**base)
        inp = cfg_scenario_inputs(k, sc)
        shared.sd_model = types.SimpleNamespace(cond_stage_key="edit" if sc.get("edit") else "txt",
                                                model=types.SimpleNamespace(conditioning_key="crossattn-adm" if sc.get("adm") else "hybrid"),
                                                cond_stage_model_empty_prompt=inp["empty"])
        ref = load_by_path("ref_cfg_denoiser", "modules/sd_samplers_cfg_denoiser.py")       # re-imported: binds this opts object

        class D(ref.CFGDenoiser):
            @property
            def inner_model(self):
                return lambda x_in, sigma_in, cond: cfg_inner_model(x_in, sigma_in, cond["c_crossattn"][0],
                                                                    cond["c_adm"] if "c_adm" in cond else cond["c_concat"][0])

        d = D(types.SimpleNamespace(last_latent=None, sampler_extra_args={}))
        d.p = types.SimpleNamespace(extra_generation_params={}, scripts=None)
        d.step, d.total_steps = sc.get("step", 0), sc.get("total_steps", 20)
        d.image_cfg_scale = sc.get("image_cfg_scale")
        d.cond_scale_miltiplier = sc.get("cond_scale_miltiplier", 1.0)
        d.need_last_noise_uncond = sc.get("need_last_noise_uncond", False)
        d.mask_before_denoising = sc.get("mask_before", False)
        if sc.get("mask") or sc.get("edit"):
            d.init_latent = inp["init_latent"]
        if sc.get("mask"):
            d.mask, d.nmask = inp["mask"], 1.0 - inp["mask"]
        res = d.forward(inp["x"].clone(), inp["sigma"], inp["uncond"].clone(), (inp["conds_list"], inp["cond"].clone()), 7.0,
                        sc.get("s_min_uncond", 0.0), inp["image_cond"])
        out[f"{name}_denoised"] = res.numpy()
        out[f"{name}_last_latent"] = d.sampler.last_latent.numpy()
        out[f"{name}_flags"] = np.array([d.padded_cond_uncond, d.padded_cond_uncond_v0, d.step,
                                         "NGMS" in d.p.extra_generation_params, "Skip Early CFG" in d.p.extra_generation_params], dtype=np.int64)
        if d.need_last_noise_uncond:
            out[f"{name}_last_noise_uncond"] = d.last_noise_uncond.numpy()
    np.savez_compressed(os.path.join(OUT, "cfg_denoiser.npz"), **out)
    print("cfg_denoiser.npz")


class FakeFirstStage:
    """Deterministic stand-in for the VAE encoder in the image-conditioning fixture: a fixed strided conv to 8 "moment"
    channels.  Exposes both the reference's sd_model methods and the oracle's model.vae methods."""
    scale_factor = 0.5

    def __init__(self):
        self.w = seeded((8, 3, 2, 2), 6100, 0.4)
        self.vae = self
        self.model = types.SimpleNamespace(conditioning_key="hybrid")
        self.first_stage_model = types.SimpleNamespace(to=lambda *a, **k: None)
        self.is_sdxl_inpaint = False
        self.dtype = torch.float32

    def encode_moments(self, x):
        return torch.nn.functional.conv2d(x, self.w, stride=2)

    def encode_first_stage(self, x):                      # the reference gets a distribution object; .mode() = the mean
        m = self.encode_moments(x)
        return types.SimpleNamespace(moments=m, mode=lambda: torch.chunk(m, 2, dim=1)[0])

    def get_first_stage_encoding(self, enc):
        return torch.chunk(enc.moments, 2, dim=1)[0] * self.scale_factor

    def encode_first_stage_mean(self, x):
        return torch.chunk(self.encode_moments(x), 2, dim=1)[0] * self.scale_factor


def gen_image_conditioning():
    """Exec, from modules/processing.py's own text, txt2img_image_conditioning (:100-133) and the methods
    inpainting_image_conditioning (:332-374) / edit_image_conditioning (:321-324), plus images_tensor_to_samples from
    modules/sd_samplers_common.py (:83-113), over FakeFirstStage; nothing is copied into this repository."""
    import re
    import textwrap
    fs = FakeFirstStage()
    shared = types.SimpleNamespace(sd_model=fs, device="cpu", opts=types.SimpleNamespace(inpainting_mask_weight=1.0, sd_vae_encode_method="Full"))
    devices = types.SimpleNamespace(device="cpu", dtype=torch.float32, dtype_vae=torch.float32)
    ns = {"torch": torch, "np": np, "shared": shared, "opts": shared.opts, "devices": devices, "approximation_indexes": {"Full": 0}}
    src = open(os.path.join(REF, "modules/sd_samplers_common.py")).read()
    a = src.index("def images_tensor_to_samples")
    exec(src[a:src.index("def store_latent")], ns)
    src = open(os.path.join(REF, "modules/processing.py")).read()
    a = src.index("def txt2img_image_conditioning(sd_model")
    exec(src[a:src.index("@dataclass(repr=False)")], ns)
    a = src.index("    def edit_image_conditioning(self")
    b = src.index("    def unclip_image_conditioning(self")
    exec(textwrap.dedent(src[a:b]), ns)
    a = src.index("    def inpainting_image_conditioning(self")
    b = src.index("    def img2img_image_conditioning(self")
    exec(textwrap.dedent(src[a:b]), ns)
    self_ = types.SimpleNamespace(sd_model=fs)
    out = {}
    x = torch.zeros(3, 4, 6, 8)
    out["txt2img"] = ns["txt2img_image_conditioning"](fs, x, 16, 12).numpy()
    img = seeded((2, 3, 12, 16), 6101).clamp(-1, 1)
    from PIL import Image
    mask_u8 = ((seeded((12, 16), 6102) * 0.5 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).numpy()
    mask = Image.fromarray(mask_u8, mode="L")            # the webui hands a PIL mask over; tensors skip the rounding (:336-347)
    out["mask_u8"] = mask_u8
    lat = torch.zeros(2, 4, 6, 8)
    out["inpaint_round"] = ns["inpainting_image_conditioning"](self_, img, lat, image_mask=mask, round_image_mask=True).numpy()
    out["inpaint_soft"] = ns["inpainting_image_conditioning"](self_, img, lat, image_mask=mask, round_image_mask=False).numpy()
    out["inpaint_nomask"] = ns["inpainting_image_conditioning"](self_, img, lat).numpy()
    self_.inpainting_mask_weight = 0.35
    out["inpaint_weight"] = ns["inpainting_image_conditioning"](self_, img, lat, image_mask=mask, round_image_mask=True).numpy()
    out["edit"] = ns["edit_image_conditioning"](self_, img).numpy()
    np.savez_compressed(os.path.join(OUT, "image_conditioning.npz"), **out)
    print("image_conditioning.npz")


RESIZE_CASES = [(48, 32, "Lanczos"), (56, 40, "Lanczos"), (48, 32, "Nearest"), (50, 36, "Nearest"), (48, 32, "None"), (48, 32, None),
                (20, 12, "Lanczos"), (96, 64, "Nearest"), (24, 16, "Lanczos")]


def gen_resize_image():
    """Exec, from their own text, images.resize_image (modules/images.py:252-291) and the Upscaler driver loop + the None /
    Lanczos / Nearest scalers (modules/upscaler.py:10-154) and run RESIZE_CASES on a seeded 24x16 RGB image (mode 0 = what the
    non-latent hires fix calls, modules/processing.py:1411)."""
    from PIL import Image
    import abc
    shared = types.SimpleNamespace(opts=types.SimpleNamespace(ESRGAN_tile=192, ESRGAN_tile_overlap=8, upscaler_for_img2img=None),
                                   device="cpu", cmd_opts=types.SimpleNamespace(no_half=True), models_path="/nonexistent",
                                   state=types.SimpleNamespace(interrupted=False), sd_upscalers=[])
    modules_ns = types.SimpleNamespace(shared=shared)
    ns = {"Image": Image, "PIL": __import__("PIL"), "os": os, "abstractmethod": abc.abstractmethod, "modules": modules_ns, "shared": shared,
          "modelloader": None}
    src = open(os.path.join(REF, "modules/upscaler.py")).read()
    exec(src[src.index("LANCZOS = "):], ns)
    shared.sd_upscalers = [*ns["UpscalerNone"]().scalers, *ns["UpscalerLanczos"]().scalers, *ns["UpscalerNearest"]().scalers]
    src = open(os.path.join(REF, "modules/images.py")).read()
    a = src.index("def resize_image(")
    ns["opts"] = shared.opts
    exec(src[a:src.index("\nif not shared.cmd_opts.unix_filenames_sanitization", a)], ns)
    g = np.random.RandomState(77)
    base = g.randint(0, 256, size=(16, 24, 3)).astype(np.uint8)
    out = {"base": base}
    for k, (w, h, name) in enumerate(RESIZE_CASES):
        out[f"r{k}"] = np.array(ns["resize_image"](0, Image.fromarray(base), w, h, upscaler_name=name))
        assert out[f"r{k}"].shape == (h, w, 3)
    np.savez_compressed(os.path.join(OUT, "resize_image.npz"), **out)
    print("resize_image.npz")


FRONTEND_RESIZE_CASES = [(1, 48, 32, None), (1, 32, 48, None), (1, 40, 40, "Lanczos"), (2, 48, 20, None), (2, 20, 48, None), (2, 36, 24, None),
                         (2, 50, 36, "Nearest"), (1, 12, 30, "None")]
FRONTEND_CROP_CASES = [((10, 20, 40, 30), 64, 64, 96, 80), ((0, 0, 90, 10), 64, 64, 96, 80), ((50, 5, 60, 75), 128, 64, 96, 80),
                       ((80, 60, 96, 80), 64, 96, 96, 80), ((3, 3, 5, 6), 512, 512, 96, 80)]


def frontend_inputs():
    """Seeded 96x80 RGB image, a soft two-blob L mask, an RGBA mask whose alpha is the mask, and an all-black mask."""
    from PIL import Image
    g = np.random.RandomState(4321)
    img = Image.fromarray(g.randint(0, 256, size=(80, 96, 3)).astype(np.uint8))
    m = np.zeros((80, 96), np.uint8)
    m[20:45, 30:70] = 255
    m[55:60, 5:15] = 200
    m[30:35, 40:50] = 90
    mask = Image.fromarray(m)
    rgba = np.zeros((80, 96, 4), np.uint8)
    rgba[..., :3] = 17
    rgba[..., 3] = m
    return img, mask, Image.fromarray(rgba, "RGBA"), Image.fromarray(np.zeros((80, 96), np.uint8))


def gen_img2img_frontend():
    """The PIL side of img2img / inpainting: images.resize_image modes 1 / 2 (modules/images.py:293-326), modules/masking.py loaded as a
    module, and create_binary_mask / uncrop / apply_overlay exec'd from modules/processing.py's text (:70-98)."""
    from PIL import Image
    import abc
    shared = types.SimpleNamespace(opts=types.SimpleNamespace(ESRGAN_tile=192, ESRGAN_tile_overlap=8, upscaler_for_img2img=None),
                                   device="cpu", cmd_opts=types.SimpleNamespace(no_half=True), models_path="/nonexistent",
                                   state=types.SimpleNamespace(interrupted=False), sd_upscalers=[])
    ns = {"Image": Image, "PIL": __import__("PIL"), "os": os, "abstractmethod": abc.abstractmethod, "modules": types.SimpleNamespace(shared=shared),
          "shared": shared, "modelloader": None}
    src = open(os.path.join(REF, "modules/upscaler.py")).read()
    exec(src[src.index("LANCZOS = "):], ns)
    shared.sd_upscalers = [*ns["UpscalerNone"]().scalers, *ns["UpscalerLanczos"]().scalers, *ns["UpscalerNearest"]().scalers]
    src = open(os.path.join(REF, "modules/images.py")).read()
    a = src.index("def resize_image(")
    ns["opts"] = shared.opts
    exec(src[a:src.index("\nif not shared.cmd_opts.unix_filenames_sanitization", a)], ns)
    resize_image = ns["resize_image"]
    masking = load_by_path("ref_masking", "modules/masking.py")
    img, mask, rgba, black = frontend_inputs()
    out = {}
    for k, (mode, w, h, name) in enumerate(FRONTEND_RESIZE_CASES):
        out[f"resize{k}"] = np.array(resize_image(mode, img, w, h, upscaler_name=name))
    out["resize_mask_m2"] = np.array(resize_image(2, mask, 64, 64))             # an L image comes back as RGB (the canvas is RGB)
    for k, pad in enumerate((0, 4, 32)):
        out[f"crop_v2_{k}"] = np.array(masking.get_crop_region_v2(mask, pad))
        out[f"crop_{k}"] = np.array(masking.get_crop_region(mask, pad))
        out[f"crop_black_{k}"] = np.array(masking.get_crop_region(black, pad))
    assert masking.get_crop_region_v2(black, 3) is None
    for k, (box, pw, ph, iw, ih) in enumerate(FRONTEND_CROP_CASES):
        out[f"expand{k}"] = np.array(masking.expand_crop_region(box, pw, ph, iw, ih))
    out["fill"] = np.array(masking.fill(img, mask))
    psrc = open(os.path.join(REF, "modules/processing.py")).read()
    pns = {"Image": Image, "images": types.SimpleNamespace(resize_image=resize_image)}
    a = psrc.index("def uncrop(")
    exec(psrc[a:psrc.index("def txt2img_image_conditioning(", a)], pns)
    for tag, m, rnd in (("rgba_round", rgba, True), ("rgba_soft", rgba, False), ("l", mask, True), ("rgb", img, True)):
        out[f"binary_{tag}"] = np.array(pns["create_binary_mask"](m, round=rnd))
    overlay = Image.new('RGBa', (img.width, img.height))
    from PIL import ImageOps
    overlay.paste(img.convert("RGBA").convert("RGBa"), mask=ImageOps.invert(mask.convert('L')))
    overlay = overlay.convert('RGBA')
    gen = Image.fromarray(np.random.RandomState(99).randint(0, 256, size=(80, 96, 3)).astype(np.uint8))
    a_img, a_orig = pns["apply_overlay"](gen, None, overlay)
    out["overlay_full"], out["overlay_full_orig"] = np.array(a_img), np.array(a_orig)
    small = Image.fromarray(np.random.RandomState(98).randint(0, 256, size=(64, 64, 3)).astype(np.uint8))
    b_img, b_orig = pns["apply_overlay"](small, (30, 20, 40, 25), overlay)
    out["overlay_paste"], out["overlay_paste_orig"] = np.array(b_img), np.array(b_orig)
    np.savez_compressed(os.path.join(OUT, "img2img_frontend.npz"), **out)
    print("img2img_frontend.npz", len(out))


def refiner_cases():
    """(step, total_steps, sigma or None, sigma-space?, switch_at, by_steps, has_refiner, already_refiner, enable_hr, is_hr_pass, hires_pass_opt)"""
    cases = []
    for step, sigma in ((0, 14.6), (3, 6.1), (7, 1.9), (12, 0.71), (17, 0.2), (19, 0.03)):
        for switch_at in (0.5, 0.8, None):
            cases.append((step, 20, sigma, True, switch_at, False, True, False, False, False, "second pass"))
            cases.append((step, 20, sigma, True, switch_at, True, True, False, False, False, "second pass"))
    for t in (981.0, 601.0, 421.0, 201.0, 1.0):
        cases.append((5, 20, t, False, 0.6, False, True, False, False, False, "second pass"))
    cases.append((15, 20, 0.4, True, 0.5, False, False, False, False, False, "second pass"))     # no refiner given
    cases.append((15, 20, 0.4, True, 0.5, False, True, True, False, False, "second pass"))       # already on the refiner
    for opt in ("first pass", "second pass", "both passes"):
        for is_hr in (False, True):
            cases.append((15, 20, 0.4, True, 0.5, False, True, False, True, is_hr, opt))
    cases.append((15, 20, None, True, 0.5, False, True, False, False, False, "second pass"))     # sigma None -> by steps
    return cases


def gen_refiner():
    """Exec apply_refiner from modules/sd_samplers_common.py (:158-202) and record its decision for refiner_cases()."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import kdiffusion as okd
    src = open(os.path.join(REF, "modules/sd_samplers_common.py")).read()
    a = src.index("def apply_refiner(")
    ns = {"torch": torch}
    opts = types.SimpleNamespace()
    base_info, ref_info = types.SimpleNamespace(short_title="base"), types.SimpleNamespace(short_title="refiner")
    shared = types.SimpleNamespace(sd_model=types.SimpleNamespace(sd_checkpoint_info=base_info))
    calls = []
    ns.update(opts=opts, shared=shared, devices=types.SimpleNamespace(torch_gc=lambda: None),
              sd_models=types.SimpleNamespace(SkipWritingToConfig=lambda: __import__("contextlib").nullcontext(),
                                              reload_model_weights=lambda info=None: calls.append(info)))
    exec(src[a:src.index("class TorchHijack")], ns)
    sigmas = okd.CompVisDenoiser(None, okd.make_alphas_cumprod()).sigmas
    out = []
    for (step, total, sigma, sigma_space, switch_at, by_steps, has_ref, already, enable_hr, is_hr, hopt) in refiner_cases():
        opts.refiner_switch_by_sample_steps, opts.hires_fix_refiner_pass = by_steps, hopt
        shared.sd_model.sd_checkpoint_info = ref_info if already else base_info
        p = types.SimpleNamespace(extra_generation_params={}, refiner_switch_at=switch_at, refiner_checkpoint_info=ref_info if has_ref else None,
                                  enable_hr=enable_hr, is_hr_pass=is_hr, setup_conds=lambda: None)
        inner = types.SimpleNamespace(sigmas=sigmas) if sigma_space else types.SimpleNamespace()
        d = types.SimpleNamespace(step=step, total_steps=total, p=p, inner_model=inner, update_inner_model=lambda: None)
        out.append(bool(ns["apply_refiner"](d, None if sigma is None else torch.full((2,), float(sigma)))))
    np.savez_compressed(os.path.join(OUT, "refiner.npz"), decisions=np.array(out))
    print("refiner.npz", sum(out), "of", len(out))


def lyco_cases():
    """name -> (module kind, sd_module spec, weight-dict builder).  Shapes are small; seeds derive from the case index."""
    lin, conv3, conv1 = ("linear", 24, 16), ("conv", 24, 16, 3), ("conv", 24, 16, 1)       # (out=24, in=16)

    def t(shape, seed, scale=0.3):
        return seeded(shape, seed, scale)

    cases = {
        "lora_linear": ("lora", lin, lambda k: {"lora_up.weight": t((24, 4), k), "lora_down.weight": t((4, 16), k + 1), "alpha": torch.tensor(2.0)}),
        "lora_linear_scale": ("lora", lin, lambda k: {"lora_up.weight": t((24, 4), k), "lora_down.weight": t((4, 16), k + 1), "scale": torch.tensor(0.7)}),
        "lora_conv3": ("lora", conv3, lambda k: {"lora_up.weight": t((24, 4, 1, 1), k), "lora_down.weight": t((4, 16, 3, 3), k + 1), "alpha": torch.tensor(4.0)}),
        "lora_conv_cp": ("lora", conv3, lambda k: {"lora_up.weight": t((24, 5, 1, 1), k), "lora_down.weight": t((4, 16, 1, 1), k + 1),
                                                   "lora_mid.weight": t((5, 4, 3, 3), k + 2), "alpha": torch.tensor(2.0)}),
        "lora_dora": ("lora", lin, lambda k: {"lora_up.weight": t((24, 4), k), "lora_down.weight": t((4, 16), k + 1), "alpha": torch.tensor(2.0),
                                              "dora_scale": t((1, 16), k + 2, 0.2).abs() + 1.0}),
        "lora_conv_dora": ("lora", conv3, lambda k: {"lora_up.weight": t((24, 4, 1, 1), k), "lora_down.weight": t((4, 16, 3, 3), k + 1),
                                                     "alpha": torch.tensor(4.0), "dora_scale": t((1, 16, 1, 1), k + 2, 0.2).abs() + 1.0}),
        "lora_dyn": ("lora", lin, lambda k: {"lora_up.weight": t((24, 8), k), "lora_down.weight": t((8, 16), k + 1), "alpha": torch.tensor(8.0)}),
        "hada_linear": ("hada", lin, lambda k: {"hada_w1_a": t((24, 4), k), "hada_w1_b": t((4, 16), k + 1), "hada_w2_a": t((24, 4), k + 2),
                                                "hada_w2_b": t((4, 16), k + 3), "alpha": torch.tensor(2.0)}),
        "hada_conv_t": ("hada", conv3, lambda k: {"hada_w1_a": t((4, 24), k), "hada_w1_b": t((4, 16), k + 1), "hada_t1": t((4, 4, 3, 3), k + 2),
                                                  "hada_w2_a": t((4, 24), k + 3), "hada_w2_b": t((4, 16), k + 4), "hada_t2": t((4, 4, 3, 3), k + 5),
                                                  "alpha": torch.tensor(4.0)}),
        "lokr_full": ("lokr", lin, lambda k: {"lokr_w1": t((4, 2), k), "lokr_w2": t((6, 8), k + 1), "alpha": torch.tensor(3.0)}),
        "lokr_lowrank": ("lokr", lin, lambda k: {"lokr_w1_a": t((4, 2), k), "lokr_w1_b": t((2, 2), k + 1), "lokr_w2_a": t((6, 3), k + 2),
                                                 "lokr_w2_b": t((3, 8), k + 3), "alpha": torch.tensor(1.5)}),
        "lokr_conv": ("lokr", conv3, lambda k: {"lokr_w1": t((4, 2), k), "lokr_w2": t((6, 8, 3, 3), k + 1), "alpha": torch.tensor(3.0)}),
        "lokr_conv_t2": ("lokr", conv3, lambda k: {"lokr_w1": t((4, 2), k), "lokr_w2_a": t((3, 6), k + 1), "lokr_w2_b": t((3, 8), k + 2),
                                                   "lokr_t2": t((3, 3, 3, 3), k + 3), "alpha": torch.tensor(3.0)}),
        "glora": ("glora", lin, lambda k: {"a1.weight": t((4, 16), k), "a2.weight": t((16, 4), k + 1), "b1.weight": t((4, 16), k + 2),
                                           "b2.weight": t((24, 4), k + 3), "alpha": torch.tensor(2.0)}),
        "ia3_out": ("ia3", lin, lambda k: {"weight": t((24,), k), "on_input": torch.tensor(False)}),
        "ia3_in": ("ia3", lin, lambda k: {"weight": t((16,), k), "on_input": torch.tensor(True)}),
        "full": ("full", conv1, lambda k: {"diff": t((24, 16, 1, 1), k)}),
        "full_bias": ("full", lin, lambda k: {"diff": t((24, 16), k), "diff_b": t((24,), k + 1)}),
        "norm": ("norm", ("groupnorm", 24), lambda k: {"w_norm": t((24,), k), "b_norm": t((24,), k + 1)}),
    }
    return cases


def lyco_oft_cases():
    """OFT family (network_oft.py), kept apart from lyco_cases(): the oracle restates them, the engine host does not yet."""
    lin, conv3 = ("linear", 24, 16), ("conv", 24, 16, 3)
    t = lambda shape, seed, scale=0.2: seeded(shape, seed, scale)
    return {
        "oft_kohya": ("oft", lin, lambda k: {"oft_blocks": t((4, 6, 6), k)}),                                   # 4 blocks of 6: no constraint
        "oft_kohya_constrained": ("oft", lin, lambda k: {"oft_blocks": t((4, 6, 6), k, 0.6), "alpha": torch.tensor(0.01)}),
        "oft_conv": ("oft", conv3, lambda k: {"oft_blocks": t((3, 8, 8), k)}),
        "oft_old_diag": ("oft", lin, lambda k: {"oft_diag": t((6, 4, 4), k) + torch.eye(4)}),                    # ready rotation blocks R
        "boft": ("oft", lin, lambda k: {"oft_blocks": t((2, 6, 4, 4), k)}),                                     # 2 butterfly factors
        "boft_rescale": ("oft", conv3, lambda k: {"oft_blocks": t((2, 6, 4, 4), k), "rescale": t((24, 1), k + 1, 0.1) + 1.0}),
    }


def lyco_bias_cases():
    """Modules carrying the extra dense "bias" entry of network.py:154, 196-199 (added to updown before scale / DoRA / multiplier).
    Kept apart (and indexed after the other two tables) so that the seeds of the older cases do not move."""
    lin, conv3 = ("linear", 24, 16), ("conv", 24, 16, 3)
    t = lambda shape, seed, scale=0.3: seeded(shape, seed, scale)
    return {
        "lora_linear_bias": ("lora", lin, lambda k: {"lora_up.weight": t((24, 4), k), "lora_down.weight": t((4, 16), k + 1), "alpha": torch.tensor(2.0),
                                                     "bias": t((24, 16), k + 2, 0.05)}),
        "lora_conv3_bias": ("lora", conv3, lambda k: {"lora_up.weight": t((24, 4, 1, 1), k), "lora_down.weight": t((4, 16, 3, 3), k + 1),
                                                      "alpha": torch.tensor(4.0), "bias": t((24, 16, 3, 3), k + 2, 0.05)}),
        "lora_dora_bias": ("lora", lin, lambda k: {"lora_up.weight": t((24, 4), k), "lora_down.weight": t((4, 16), k + 1), "alpha": torch.tensor(2.0),
                                                   "dora_scale": t((1, 16), k + 2, 0.2).abs() + 1.0, "bias": t((24, 16), k + 3, 0.05)}),
        "hada_linear_bias": ("hada", lin, lambda k: {"hada_w1_a": t((24, 4), k), "hada_w1_b": t((4, 16), k + 1), "hada_w2_a": t((24, 4), k + 2),
                                                     "hada_w2_b": t((4, 16), k + 3), "alpha": torch.tensor(2.0), "bias": t((24 * 16,), k + 4, 0.05)}),
    }


def lyco_orig_weight(spec, k):
    if spec[0] == "linear":
        return seeded((spec[1], spec[2]), 8000 + k, 0.2)
    if spec[0] == "conv":
        return seeded((spec[1], spec[2], spec[3], spec[3]), 8000 + k, 0.2)
    return seeded((spec[1],), 8000 + k, 0.2) + 1.0


def gen_lyco():
    """Load extensions-builtin/Lora/{network,lyco_helpers,network_lora,network_hada,network_lokr,network_glora,network_ia3,
    network_full,network_norm}.py by path (webui modules stubbed) and record calc_updown(orig_weight) for lyco_cases() with
    unet_multiplier 0.8 (dyn_dim 3 for the lora_dyn case)."""
    mods = sys.modules.setdefault("modules", types.ModuleType("modules"))
    for n in ("sd_models", "cache", "errors", "hashes", "devices"):
        m = types.ModuleType("modules." + n)
        sys.modules["modules." + n] = m
        setattr(mods, n, m)
    sys.modules["modules.devices"].cpu = torch.device("cpu")
    sys.modules["modules.devices"].dtype = torch.float32
    sys.modules["modules.devices"].device = torch.device("cpu")
    shared = types.ModuleType("modules.shared")
    shared.opts = types.SimpleNamespace()
    sys.modules["modules.shared"] = shared
    mods.shared = shared
    for n in ("modules.models", "modules.models.sd3"):
        sys.modules.setdefault(n, types.ModuleType(n))
    mmdit = types.ModuleType("modules.models.sd3.mmdit")
    mmdit.QkvLinear = type("QkvLinear", (torch.nn.Linear,), {})
    sys.modules["modules.models.sd3.mmdit"] = mmdit
    sys.modules["modules.models"].sd3 = sys.modules["modules.models.sd3"]
    sys.modules["modules.models.sd3"].mmdit = mmdit
    mods.models = sys.modules["modules.models"]
    lora_dir = "extensions-builtin/Lora/"
    for name in ("lyco_helpers", "network"):
        sys.modules[name] = load_by_path(name, lora_dir + name + ".py")
    kinds = {}
    for name, cls in (("lora", "ModuleTypeLora"), ("hada", "ModuleTypeHada"), ("lokr", "ModuleTypeLokr"), ("glora", "ModuleTypeGLora"),
                      ("ia3", "ModuleTypeIa3"), ("full", "ModuleTypeFull"), ("norm", "ModuleTypeNorm")):
        kinds[name] = getattr(load_by_path("network_" + name, lora_dir + "network_" + name + ".py"), cls)()
    network = sys.modules["network"]
    kinds["oft"] = load_by_path("network_oft", lora_dir + "network_oft.py").ModuleTypeOFT()
    out = {}
    all_cases = list(lyco_cases().items()) + list(lyco_oft_cases().items()) + list(lyco_bias_cases().items())
    for k, (name, (kind, spec, build)) in enumerate(all_cases):
        if spec[0] == "linear":
            sd_module = torch.nn.Linear(spec[2], spec[1])
        elif spec[0] == "conv":
            sd_module = torch.nn.Conv2d(spec[2], spec[1], spec[3], padding=spec[3] // 2)
        else:
            sd_module = torch.nn.GroupNorm(4, spec[1])
        orig = lyco_orig_weight(spec, k)
        with torch.no_grad():
            sd_module.weight.copy_(orig)
        net = network.Network("n", None)
        net.unet_multiplier, net.te_multiplier = 0.8, 0.3
        net.dyn_dim = 3 if name == "lora_dyn" else None
        w = build(9000 + 10 * k)
        weights = network.NetworkWeights(network_key="lora_unet_x", sd_key="diffusion_model_x", w=dict(w), sd_module=sd_module)
        module = kinds[kind].create_module(net, weights)
        assert module is not None, name
        for other, mt in kinds.items():          # the type dispatch order of networks.py:26-36 never mis-assigns these cases
            if other != kind and other in ("hada", "lokr", "glora", "ia3", "full", "norm") and kind != "oft":
                assert mt.create_module(net, network.NetworkWeights("a", "b", dict(w), sd_module)) is None or kind == "lora", (name, other)
        with torch.no_grad():
            updown, ex_bias = module.calc_updown(sd_module.weight)
        out[name + "_updown"] = updown.numpy()
        if ex_bias is not None:
            out[name + "_ex_bias"] = ex_bias.numpy()
    np.savez_compressed(os.path.join(OUT, "lyco.npz"), **out)
    print("lyco.npz", len(out))


MULTICOND_PROMPTS = ["a cat", "a cat AND a dog :1.5", "castle :0.25 AND sky:-.5 AND a cat", "ANDROID and sand", "x:1. AND y : +2 AND z:abc",
                     "  spaced   AND  out :3  ", "a: b :0.7"]


def prompt_cond_schedules(dict_conds=False):
    """Seeded schedules shared by the generator and the tests: 3 images; image 0 one prompt with two schedule entries, image 1
    two AND-ed prompts (weights 1.0 / 0.6; the second has 3 entries and more tokens), image 2 one constant prompt."""
    def cond(seed, tokens=8):
        c = seeded((tokens, 6), seed, 0.5)
        return {"crossattn": c, "vector": seeded((10,), seed + 500, 0.5)} if dict_conds else c
    S = lambda end, seed, tokens=8: (end, cond(seed, tokens))
    multi = [[([S(4, 7001), S(20, 7002)], 1.0)],
             [([S(20, 7003)], 1.0), ([S(2, 7004, 16), S(9, 7005, 16), S(20, 7006, 16)], 0.6)],
             [([S(20, 7007)], 0.8)]]
    uncond = [[S(6, 7011), S(20, 7012)], [S(20, 7013)], [S(3, 7014), S(20, 7015)]]
    return multi, uncond


def gen_prompt_cond():
    """Exec, from modules/prompt_parser.py's own text (the module itself needs lark), the conditioning containers and
    get_multicond_prompt_list / reconstruct_cond_batch / stack_conds / reconstruct_multicond_batch (:136-154, 205-349)."""
    import json
    import re
    from collections import namedtuple
    src = open(os.path.join(REF, "modules/prompt_parser.py")).read()
    ns = {"re": re, "torch": torch, "namedtuple": namedtuple, "annotations": None}
    a = src.index("ScheduledPromptConditioning = namedtuple")
    b = src.index("def get_learned_conditioning(model")
    exec("from __future__ import annotations\n" + src[a:b], ns)
    a = src.index("re_AND = re.compile")
    b = src.index("def get_multicond_learned_conditioning(")
    exec("from __future__ import annotations\n" + src[a:b], ns)
    a = src.index("class DictWithShape(dict)")
    exec("from __future__ import annotations\n" + src[a:], ns)
    res_indexes, flat, idx = ns["get_multicond_prompt_list"](MULTICOND_PROMPTS)
    meta = {"res_indexes": res_indexes, "flat": list(flat), "indexes": idx}
    out = {}
    for dict_conds in (False, True):
        tag = "dict_" if dict_conds else ""
        multi, uncond = prompt_cond_schedules(dict_conds)
        SPC, CSPC, MLC = ns["ScheduledPromptConditioning"], ns["ComposableScheduledPromptConditioning"], ns["MulticondLearnedConditioning"]
        c = MLC(shape=(3,), batch=[[CSPC([SPC(e, t) for e, t in sch], w) for sch, w in img] for img in multi])
        uc = [[SPC(e, t) for e, t in sch] for sch in uncond]
        for step in (0, 2, 3, 4, 5, 9, 10, 25):
            conds_list, stacked = ns["reconstruct_multicond_batch"](c, step)
            u = ns["reconstruct_cond_batch"](uc, step)
            meta[f"{tag}conds_list_{step}"] = conds_list
            if dict_conds:
                for k in ("crossattn", "vector"):
                    out[f"{tag}c_{k}_{step}"] = stacked[k].numpy()
                    out[f"{tag}uc_{k}_{step}"] = u[k].numpy()
                assert tuple(stacked.shape) == tuple(stacked["crossattn"].shape)
            else:
                out[f"c_{step}"] = stacked.numpy()
                out[f"uc_{step}"] = u.numpy()
    np.savez_compressed(os.path.join(OUT, "prompt_cond.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "prompt_cond.json"), "w"), indent=0, sort_keys=True)
    print("prompt_cond.npz prompt_cond.json")


IMAGE_RNG_CASES = [
    # name, shape, seeds, kwargs, eta_noise_seed_delta
    ("plain", (4, 8, 8), [10, 11], {}, 0),
    ("ensd", (4, 8, 8), [10, 11], {}, 31337),
    ("subseed", (4, 8, 8), [10, 11], dict(subseeds=[20, 21], subseed_strength=0.3), 0),
    ("subseed_short_list", (4, 8, 8), [10, 11, 12], dict(subseeds=[20], subseed_strength=0.75), 0),
    ("subseed_same", (4, 8, 8), [10, 11], dict(subseeds=[10, 11], subseed_strength=0.4), 0),       # dot = 1 -> the lerp branch
    ("resize_smaller", (4, 8, 8), [10, 11], dict(seed_resize_from_h=48, seed_resize_from_w=32), 0),
    ("resize_larger", (4, 8, 8), [10, 11], dict(seed_resize_from_h=96, seed_resize_from_w=80), 0),
    ("resize_mixed", (4, 8, 10), [10], dict(seed_resize_from_h=40, seed_resize_from_w=112), 0),
    ("subseed_resize_ensd", (4, 8, 8), [10, 11], dict(subseeds=[20, 21], subseed_strength=0.5, seed_resize_from_h=48, seed_resize_from_w=96), 7),
]


def gen_image_rng():
    """Load modules/rng.py (+ the real modules/rng_philox.py) by path with randn_source "NV" and record ImageRNG.next() three
    times for IMAGE_RNG_CASES: subseed slerp (both branches), seed-resize (smaller / larger / mixed), eta_noise_seed_delta."""
    mods = sys.modules.setdefault("modules", types.ModuleType("modules"))
    devices = types.ModuleType("modules.devices")
    devices.device, devices.cpu = torch.device("cpu"), torch.device("cpu")
    shared = types.ModuleType("modules.shared")
    shared.opts = types.SimpleNamespace(randn_source="NV", eta_noise_seed_delta=0)
    shared.device = torch.device("cpu")
    philox = load_by_path("modules.rng_philox", "modules/rng_philox.py")
    for n, m in (("devices", devices), ("shared", shared), ("rng_philox", philox)):
        sys.modules["modules." + n] = m
        setattr(mods, n, m)
    ref = load_by_path("ref_rng", "modules/rng.py")
    out = {}
    for name, shape, seeds, kw, ensd in IMAGE_RNG_CASES:
        shared.opts.eta_noise_seed_delta = ensd
        r = ref.ImageRNG(shape, seeds, **kw)
        for k in range(3):
            out[f"{name}_{k}"] = r.next().numpy()
    np.savez_compressed(os.path.join(OUT, "image_rng.npz"), **out)
    print("image_rng.npz")


def unet_twin_modules():
    """Oracle modules with seeded weights, shared by the generator and the test (constructed identically in both)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import unet as ou
    torch.manual_seed(0)
    attn_self = ou.CrossAttention(64, None, 4, 16)
    attn_cross = ou.CrossAttention(64, 48, 4, 16)
    st_conv = ou.SpatialTransformer(64, 4, 16, 2, 48, use_linear=False)
    st_lin = ou.SpatialTransformer(64, 2, 32, 1, 48, use_linear=True)
    for k, m in enumerate((attn_self, attn_cross, st_conv, st_lin)):
        seeded_module_weights(m, 9500 + k)
        m.eval().requires_grad_(False)
    return attn_self, attn_cross, st_conv, st_lin


def gen_unet_twins():
    """The in-tree functions the webui patches over ldm's UNet, run on the ORACLE's module instances: timestep_embedding and
    spatial_transformer_forward (modules/sd_hijack_unet.py:56-102), attention_CrossAttention_forward (the baseline attention,
    modules/hypernetworks/hypernetwork.py:382-407, hypernetworks empty).  The fixture pins the oracle's own forwards of the
    same modules (embedding order cos|sin, NCHW <-> token reshapes around proj_in / proj_out, head split / merge, softmax)."""
    import math
    from einops import rearrange, repeat
    src = open(os.path.join(REF, "modules/sd_hijack_unet.py")).read()
    ns = {"torch": torch, "math": math, "repeat": repeat}
    exec(src[src.index("def timestep_embedding(_"):src.index("class GELUHijack")], ns)
    src = open(os.path.join(REF, "modules/hypernetworks/hypernetwork.py")).read()
    ns2 = {"torch": torch, "rearrange": rearrange, "repeat": repeat, "einsum": torch.einsum, "default": lambda a, b: a if a is not None else b,
           "shared": types.SimpleNamespace(loaded_hypernetworks=[]), "apply_hypernetworks": lambda hn, context, layer=None: (context, context)}
    a = src.index("def attention_CrossAttention_forward(")
    exec(src[a:src.index("def stack_conds(")], ns2)
    attn_self, attn_cross, st_conv, st_lin = unet_twin_modules()
    out = {}
    t = torch.tensor([999.0, 500.25, 37.5, 0.0])
    out["temb_320"] = ns["timestep_embedding"](None, t, 320).numpy()
    out["temb_65"] = ns["timestep_embedding"](None, t, 65).numpy()
    x = seeded((2, 40, 64), 9600)
    ctx = seeded((2, 77, 48), 9601)
    out["attn_self"] = ns2["attention_CrossAttention_forward"](attn_self, x).numpy()
    out["attn_cross"] = ns2["attention_CrossAttention_forward"](attn_cross, x, context=ctx).numpy()
    img = seeded((2, 64, 6, 5), 9602)
    out["st_conv"] = ns["spatial_transformer_forward"](None, st_conv, img, context=[ctx, ctx]).numpy()
    out["st_lin"] = ns["spatial_transformer_forward"](None, st_lin, img, context=ctx).numpy()
    np.savez_compressed(os.path.join(OUT, "unet_twins.npz"), **out)
    print("unet_twins.npz")


def gen_euler_twin():
    """modules/models/sd3/sd3_impls.py:145-163 holds an in-tree copy of k-diffusion's to_d / sample_euler ("Algorithm 2 (Euler
    steps) from Karras et al."): run it on an analytic denoiser over a Karras and a model-uniform schedule — the only in-tree
    pin for the k-diffusion sampler family (the others are restated from the published code)."""
    import warnings
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import kdiffusion as okd
    stub = types.ModuleType("modules.models.sd3.mmdit")
    stub.MMDiT = object
    for name in ("modules", "modules.models", "modules.models.sd3"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["modules.models.sd3.mmdit"] = stub
    m = load_by_path("ref_sd3_impls_euler", "modules/models/sd3/sd3_impls.py")
    den = okd.CompVisDenoiser(None, okd.make_alphas_cumprod())
    smin, smax = den.sigmas[0].item(), den.sigmas[-1].item()

    def model(x, sigma, **kw):
        s = sigma[:, None, None, None]
        return x / (1 + s * s) + torch.tanh(0.5 * x) * (s * s / (1 + s * s)) * 0.3

    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for ci, sigmas in enumerate((okd.get_sigmas_karras(10, smin, smax), den.get_sigmas(7))):
            x0 = seeded((2, 4, 8, 8), 9700 + ci) * sigmas[0]
            out[f"c{ci}_sigmas"] = sigmas.numpy()
            out[f"c{ci}_out"] = m.sample_euler(model, x0.clone(), sigmas).numpy()
    np.savez_compressed(os.path.join(OUT, "euler_twin.npz"), **out)
    print("euler_twin.npz")


def gen_zsnr():
    """Exec rescale_zero_terminal_snr_abar from modules/sd_models.py (:628-644) on the SD schedule, in fp32 and after the fp16
    round trip of opts.use_downcasted_alpha_bar (:659-662)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import kdiffusion as okd
    src = open(os.path.join(REF, "modules/sd_models.py")).read()
    ns = {"torch": torch}
    exec(src[src.index("def rescale_zero_terminal_snr_abar("):src.index("def apply_alpha_schedule_override(")], ns)
    ac = okd.make_alphas_cumprod()
    out = {"fp32": ns["rescale_zero_terminal_snr_abar"](ac.clone()).numpy(),
           "downcast": ns["rescale_zero_terminal_snr_abar"](ac.clone().half()).float().numpy()}
    np.savez_compressed(os.path.join(OUT, "zsnr.npz"), **out)
    print("zsnr.npz")


HN_CASES = {            # name -> (dim, layer_structure, activation_func, add_layer_norm, activate_output, dropout_structure)
    "lin_121": (64, [1, 2, 1], "linear", False, False, None),
    "relu_121": (64, [1, 2, 1], "relu", False, False, None),
    "swish_ln_1221_ao": (64, [1, 2, 2, 1], "swish", True, True, None),
    "elu_drop_1221": (128, [1, 2, 2, 1], "elu", False, False, [0, 0.3, 0.3, 0]),
    "tanh_131_ao": (64, [1, 3, 1], "tanh", False, True, None),
    "sigmoid_121": (128, [1, 2, 1], "sigmoid", False, False, None),
    "leakyrelu_121": (64, [1, 2, 1], "leakyrelu", False, False, None),
    "mish_ln_121": (64, [1, 2, 1], "mish", True, False, None),
}


def gen_hypernetwork():
    """HypernetworkModule / apply_hypernetworks of modules/hypernetworks/hypernetwork.py, exec'd from the file's own text (the module
    imports half of the webui at import time): module outputs at multiplier 0.7 for HN_CASES on seeded weights, and two chained
    networks over a context (K and V paths)."""
    import inspect as _inspect
    src = open(os.path.join(REF, "modules/hypernetworks/hypernetwork.py")).read()
    a, b = src.index("class HypernetworkModule"), src.index("#param layer_structure")
    c, d = src.index("def apply_single_hypernetwork"), src.index("def attention_CrossAttention_forward")
    from torch.nn.init import normal_, xavier_normal_, xavier_uniform_, kaiming_normal_, kaiming_uniform_, zeros_
    devices = types.SimpleNamespace(torch_npu_set_device=lambda: None, device=torch.device("cpu"), cond_cast_unet=lambda x: x,
                                    cond_cast_float=lambda x: x.float())
    ns = dict(torch=torch, inspect=_inspect, normal_=normal_, xavier_normal_=xavier_normal_, xavier_uniform_=xavier_uniform_,
              kaiming_normal_=kaiming_normal_, kaiming_uniform_=kaiming_uniform_, zeros_=zeros_, devices=devices)
    exec(src[a:b] + "\n" + src[c:d], ns)
    out = {}
    mods = {}
    with torch.no_grad():
        for k, (name, (dim, ls, act, ln, ao, ds)) in enumerate(HN_CASES.items()):
            m = ns["HypernetworkModule"](dim, None, ls, act, "Normal", ln, ao, dropout_structure=ds)
            m.eval()
            seeded_module_weights(m, 6000 + k)
            m.multiplier = 0.7
            x = seeded((2, 10, dim), 6100 + k)
            out[name] = m(x).numpy()
            mods[name] = m
        # two networks chained over a width-64 context (K module, V module of each)
        hn_a = types.SimpleNamespace(layers={64: (mods["relu_121"], mods["lin_121"])})
        hn_b = types.SimpleNamespace(layers={64: (mods["tanh_131_ao"], mods["swish_ln_1221_ao"]), 128: (mods["elu_drop_1221"], mods["sigmoid_121"])})
        ctx = seeded((2, 7, 64), 6200)
        ck, cv = ns["apply_hypernetworks"]([hn_a, hn_b], ctx)
        out["chain_k"], out["chain_v"] = ck.numpy(), cv.numpy()
    np.savez_compressed(os.path.join(OUT, "hypernetwork.npz"), **out)
    print("hypernetwork.npz")


SAMPLER_NAME_CASES = [("DPM++ 2M Karras", None), ("Euler a", None), ("Euler a SGMUniform", None), ("DPM++ SDE Karras", "Exponential"),
                      ("nonexistent", "Karras"), (None, None), ("DPM++ 2M", "karras"), ("DPM++ 2M SDE Heun Exponential", None), ("LMS Karras", None),
                      ("DDIM", "DDIM"), ("Euler a", "Align Your Steps"), ("Heun", "bogus"), ("DPM++ 3M SDE Exponential", "Karras"),
                      ("Euler Polyexponential", None), ("DPM2 a Karras", None), ("UniPC", "SGM Uniform"), ("Euler a Uniform Karras", None),
                      ("DPM++ 2M SDE Karras", None), ("DPM adaptive", "Automatic"), ("Restart", "Simple"), ("LCM", None), ("PLMS Beta", "Normal")]


def gen_sampler_names():
    """Execute get_sampler_and_scheduler (modules/sd_samplers.py:105-126: pre-1.9 combined names -> sampler row + scheduler label) over
    the reference's OWN tables: sd_schedulers.schedulers from the file (loaded as in gen_schedulers) and the sampler rows' (name, options)
    parsed from the table literals of sd_samplers_kdiffusion.py:11-27, sd_samplers_timesteps.py:12-17 and sd_samplers_lcm.py:100-104 (the
    modules themselves import k_diffusion).  The function is exec'd from the file's text; nothing is copied into this repository."""
    import ast
    import functools
    import json
    import re
    gen_schedulers()                                           # leaves the k_diffusion / modules.shared stand-ins in sys.modules
    sched = load_by_path("ref_sd_schedulers_names", "modules/sd_schedulers.py")
    Row = types.SimpleNamespace
    rows = []
    kd_src = open(os.path.join(REF, "modules/sd_samplers_kdiffusion.py")).read()
    table = kd_src[kd_src.index("samplers_k_diffusion = ["):kd_src.index("]\n", kd_src.index("samplers_k_diffusion = ["))]
    for m in re.finditer(r"\(\s*'([^']+)',\s*[^,\[]+,\s*\[[^\]]*\],\s*(\{[^}]*\})\s*\)", table):
        rows.append(Row(name=m.group(1), options=ast.literal_eval(m.group(2))))
    ts_src = open(os.path.join(REF, "modules/sd_samplers_timesteps.py")).read()
    for m in re.finditer(r"\('([^']+)',\s*sd_samplers_timesteps_impl\.\w+,\s*\[[^\]]*\],\s*(\{[^}]*\})\)", ts_src):
        rows.append(Row(name=m.group(1), options=ast.literal_eval(m.group(2))))
    lcm_src = open(os.path.join(REF, "modules/sd_samplers_lcm.py")).read()
    for m in re.finditer(r"\('([^']+)',\s*sample_lcm,\s*\[[^\]]*\],\s*(\{[^}]*\})\)", lcm_src):
        rows.append(Row(name=m.group(1), options=ast.literal_eval(m.group(2))))
    assert len(rows) == 20 and rows[0].name == "DPM++ 2M", [r.name for r in rows]
    src = open(os.path.join(REF, "modules/sd_samplers.py")).read()
    fn = re.search(r"def get_sampler_and_scheduler\(.*?\n    return sampler\.name, found_scheduler\.label\n", src, re.S).group(0)
    ns = {"functools": functools, "samplers": rows, "all_samplers_map": {r.name: r for r in rows}, "sd_schedulers": sched}
    exec(fn, ns)
    out = [[list(c), bool(conv), list(ns["get_sampler_and_scheduler"](c[0], c[1], convert_automatic=conv))]
           for c in SAMPLER_NAME_CASES for conv in (True, False)]
    json.dump({"rows": [[r.name, r.options] for r in rows], "cases": out}, open(os.path.join(OUT, "sampler_names.json"), "w"), indent=0)
    print("sampler_names.json", len(out))


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_philox()
    gen_subquad()
    gen_vae()
    gen_vae_512()
    gen_ddim()
    gen_schedulers()
    gen_lora_names()
    gen_clip()
    gen_restart()
    gen_unipc()
    gen_lcm()
    gen_cfg_denoiser()
    gen_image_conditioning()
    gen_resize_image()
    gen_img2img_frontend()
    gen_refiner()
    gen_lyco()
    gen_prompt_cond()
    gen_image_rng()
    gen_unet_twins()
    gen_euler_twin()
    gen_zsnr()
    gen_hypernetwork()
    gen_sampler_names()
