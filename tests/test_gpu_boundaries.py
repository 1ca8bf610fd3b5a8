"""GPU tests of the drop-in boundaries: the SdUnet adapter (B1), the SdOptimization attention
forward inside an unmodified torch CrossAttention module (B2), the VAE decode hook (B4) and the sampler registry (B3)."""
import importlib
import types

import numpy as np
import pytest
import torch

from helpers import rel_l2, seeded

pytestmark = pytest.mark.gpu


def sub(name):
    return importlib.import_module("stable-diffusion-webui_amd." + name)


@pytest.fixture(scope="module")
def dev():
    sub("_lib").require_device()
    return torch.device("cuda", 0)


def test_sd_unet_adapter_forward_contract(dev):
    """Mi355xUnet.forward(x, timesteps, context) as called from modules/sd_unet.py:87-91: fp16 in, fp16 eps out, same device."""
    schema, sd_unet = sub("schema"), sub("sd_unet")
    from oracle import unet as ou
    cfg = schema.tiny_unet()
    sd = schema.synthetic_state_dict(cfg, None, dtype=torch.float16)
    opt = sd_unet.Mi355xUnetOption("tiny", lambda: sd, cfg)
    assert opt.label.startswith("[MI355X]") and opt.model_name == "tiny"
    unet = opt.create_unet()
    unet.activate()
    try:
        x = seeded((4, 4, 16, 16), 1).to(dev).half()
        t = torch.tensor([801.0, 801.0, 33.0, 33.0], device=dev).half()
        ctx = seeded((4, 77, 64), 2).to(dev).half()
        out = unet.forward(x, t, ctx)
        assert out.dtype == torch.float16 and out.device == x.device and out.shape == x.shape
        ref = ou.build_unet(ou.tiny_config(), sd)(x.float().cpu(), t.float().cpu(), ctx.float().cpu())
        assert rel_l2(out.float().cpu(), ref) < 8e-3
        out2 = unet.forward(x, t, ctx)                     # same context tensor: cached projections, same bits
        assert torch.equal(out, out2)
        # a NEW context that lands on the freed tensor's address (what the webui's per-step torch.cat does, and what the next
        # job's prompt does): the cache is validated by content, so the projections are rebuilt
        ctx_b = seeded((4, 77, 64), 3).to(dev).half()
        del ctx
        reused = torch.empty_like(ctx_b)
        reused.copy_(ctx_b)
        out3 = unet.forward(x, t, reused)
        assert not torch.equal(out3, out)
        ref3 = ou.build_unet(ou.tiny_config(), sd)(x.float().cpu(), t.float().cpu(), ctx_b.float().cpu())
        assert rel_l2(out3.float().cpu(), ref3) < 8e-3
        assert torch.equal(unet.forward(x, t, ctx_b), out3)      # equal content at another address: cache hit, same bits
        with pytest.raises(NotImplementedError):
            unet.forward(x, t, reused, control=[x])
    finally:
        unet.deactivate()


def test_sd_optimization_attention_inside_torch_module(dev):
    """mi355x_attention_forward bound to a torch CrossAttention module (the oracle's class has the ldm attribute names
    to_q / to_k / to_v / to_out / heads / scale that modules/sd_hijack_optimizations.py:221-281 relies on)."""
    from oracle import unet as ou
    opt_mod = sub("sd_hijack_optimizations")
    opt = opt_mod.SdOptimizationMi355x()
    assert opt.name == "mi355x" and opt.is_available() and opt.priority > 100
    x, ctx = seeded((2, 256, 320), 1), seeded((2, 77, 768), 2)
    g = torch.Generator().manual_seed(0)
    for context_dim, context in ((None, None), (768, ctx)):          # attn1 (self) and attn2 (cross) of a transformer block
        attn = ou.CrossAttention(320, context_dim, heads=8, dim_head=40)
        for p in attn.parameters():
            p.data = torch.randn(p.shape, generator=g) * (p.shape[-1] ** -0.5 if p.dim() > 1 else 0.02)
        with torch.no_grad():
            ref = attn(x, context)
            gattn = ou.CrossAttention(320, context_dim, heads=8, dim_head=40)
            gattn.load_state_dict(attn.state_dict())
            gattn = gattn.to(dev).half()
            gattn.forward = types.MethodType(opt_mod.mi355x_attention_forward, gattn)
            got = gattn(x.to(dev).half(), context=None if context is None else context.to(dev).half())
        assert got.dtype == torch.float16
        assert rel_l2(got.float().cpu(), ref) < 4e-3


def test_vae_decode_hook_replaces_first_stage_decode(dev):
    schema, hook = sub("schema"), sub("sd_vae_hook")
    from oracle import vae as ov
    cfg = schema.tiny_vae()
    sd = schema.synthetic_state_dict(None, cfg, dtype=torch.float16)
    okl = ov.build_vae(ov.tiny_vae_config(), sd)

    class FakeSdModel:
        is_sdxl = False
        first_stage_model = okl
    m = FakeSdModel()
    # the hook builds the engine with the default SD1.5 VAE config unless the model says otherwise: use the tiny config here
    eng = sub("engine").Engine(0)
    cfg1 = schema.tiny_vae(scale_factor=1.0)
    eng.load_vae(cfg1, {schema.VAE_PREFIX + k: v for k, v in okl.state_dict().items()}, decoder_only=True)
    z = seeded((2, 4, 16, 16), 3)
    with torch.no_grad():
        ref = okl.decode(z)                                # caller already divided by scale_factor
    got = eng.vae_decode(z.to(dev))
    assert rel_l2(got.cpu(), ref) < 5e-3
    assert callable(hook.install) and callable(hook.uninstall)


def test_sampler_registry_drives_engine(dev):
    ss, schema = sub("sd_samplers"), sub("schema")
    model = sub("sd_models").SdModel(schema.synthetic_state_dict(schema.tiny_unet(), None), schema.tiny_unet(), None, device=0,
                                     load_vae=False)
    s = ss.create_sampler("k_euler_a", model)
    assert isinstance(s, ss.KDiffusionSampler) and s.config.name == "Euler a"
    assert isinstance(ss.create_sampler("DDIM", model), ss.CompVisSampler)
    calls = []

    class P:
        steps, cfg_scale, eta, scheduler, is_hr_pass = 3, 4.0, None, None, False
        sampler_noise_scheduler_override = None
        rng = sub("rng").ImageRNG((4, 16, 16), [5, 6], device=dev)
    p = P()
    x = p.rng.next()
    g = torch.Generator().manual_seed(1)
    c, uc = torch.randn(2, 77, 64, generator=g).to(dev), torch.randn(2, 77, 64, generator=g).to(dev)
    s.callback_state = lambda d: calls.append(d["i"])
    out = s.sample(p, x, c, uc)
    assert calls == [0, 1, 2] and out.shape == x.shape and torch.isfinite(out).all()
    # interruption returns the last latent like Sampler.launch_sampling (modules/sd_samplers_common.py:265-281)
    s2 = ss.create_sampler("Euler", model)
    s2.stop_at = 0
    p.rng = sub("rng").ImageRNG((4, 16, 16), [5, 6], device=dev)
    out2 = s2.sample(p, p.rng.next(), c, uc)
    assert out2 is s2.last_latent
