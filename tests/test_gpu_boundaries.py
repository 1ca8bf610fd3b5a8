"""GPU tests of the drop-in boundaries: the SdUnet adapter (B1), the SdOptimization attention
forward inside an unmodified torch CrossAttention module (B2), the VAE decode hook (B4) and the sampler registry (B3)."""
import importlib
import os
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
        with pytest.raises(NotImplementedError):              # extra inputs the engine does not know are refused, loudly
            unet.forward(x, t, reused, some_other_extra_input=[x])
        with pytest.raises(sub("_lib").SdmiError, match="control"):      # ControlNet residuals are the engine's since round 6: a malformed
            unet.forward(x, t, reused, control=[x])                      # list is an error of the call, not a silent run
        assert torch.equal(unet.forward(x, t, reused), out3)             # ... and leaves nothing behind
    finally:
        unet.deactivate()


def test_sampler_row_fans_out_over_two_engines_with_the_real_sampler(dev):
    """webui_bridge.sample_over_devices with REAL engines and the real Euler-a sampler: the box has one GPU, so the option names it twice
    — the second worker gets its own engine UNet replica (an engine serves one caller at a time), both sample concurrently in their
    threads (one after the other under the host-emulated tier), and the concatenated latents are the single-call latents bit for bit
    at equal per-call batch size (4 rows -> 2 + 2 against two 2-row calls) and to fp16 rounding against the 4-row call."""
    import types
    schema, sd_unet, bridge, amd, rng = sub("schema"), sub("sd_unet"), sub("webui_bridge"), sub("sd_samplers"), sub("rng")
    cfg = schema.tiny_unet()
    sd = schema.synthetic_state_dict(cfg, None, dtype=torch.float16)
    unet = sd_unet.Mi355xUnetOption("tiny", lambda: sd, cfg).create_unet()
    unet.activate()
    webui_model = types.SimpleNamespace(alphas_cumprod=schema.make_alphas_cumprod(), model=types.SimpleNamespace(conditioning_key="crossattn"),
                                        parameterization="eps", is_sdxl=False, cond_stage_key="txt")
    view = bridge.EngineModelView(webui_model, unet)
    sub("shared").sd_model = view
    row = amd.all_samplers_map["Euler a"]
    seeds = [500, 501, 502, 503]
    g = torch.Generator().manual_seed(5)
    cond, uncond = torch.randn(4, 77, 64, generator=g).to(dev), torch.randn(4, 77, 64, generator=g).to(dev)

    def job(lo, hi):
        class P:
            steps, cfg_scale, eta, scheduler, is_hr_pass, batch_size, iteration = 2, 6.0, None, None, False, hi - lo, 0
            sampler_noise_scheduler_override, extra_generation_params = None, {}
        p = P()
        p.seeds = seeds[lo:hi]
        p.rng = rng.ImageRNG((4, 16, 16), seeds[lo:hi], device=dev)
        return p, p.rng.next()
    monkey_serial = os.environ.get("SDMI_HOSTEMU") == "1"
    prev = bridge.serial_device_workers
    bridge.serial_device_workers = monkey_serial
    try:
        def single(lo, hi):
            p, x = job(lo, hi)
            s = row.constructor(view)
            s.config = row
            return s.sample(p, x, cond[lo:hi], uncond[lo:hi]).cpu()
        whole = single(0, 4)
        halves = torch.cat([single(0, 2), single(2, 4)])
        p, x = job(0, 4)
        view._row_config = row
        fanned = bridge.sample_over_devices(row.constructor, view, "sample", p, (x, cond, uncond), {}, [0, 0]).cpu()
        reps = [k for k in bridge._unet_replicas if k[0] == id(unet)]
        assert len(reps) == 1 and bridge._unet_replicas[reps[0]].engine.handle != unet.engine.handle
        assert torch.equal(fanned, halves)
        assert rel_l2(fanned, whole) < 1e-2                   # the 4-row dispatch: other tiles (chaotic tiny model)
    finally:
        bridge.serial_device_workers = prev
        for k in [k for k in bridge._unet_replicas if k[0] == id(unet)]:
            bridge._unet_replicas.pop(k).deactivate()
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


def test_vae_decode_hook_install_uninstall_and_reinstall(dev):
    """Boundary B4 executed: sd_vae_hook.install() on a stand-in sd_model whose first_stage_model is a torch AutoencoderKL (the
    oracle's class has ldm's state-dict layout) — decode_first_stage -> first_stage_model.decode(z / scale) now runs the engine;
    install() again (the webui re-fires on_model_loaded on the SAME object after a checkpoint reload, modules/sd_models.py:994, or a
    VAE switch, modules/sd_vae.py:280) closes the previous engine, picks up the new weights, and uninstall() restores torch's decode."""
    schema, hook, engine_mod = sub("schema"), sub("sd_vae_hook"), sub("engine")
    from oracle import vae as ov
    cfg = ov.sd15_vae_config()                             # the hook assumes the real SD VAE geometry (ch 128, 4 levels)
    okl = ov.AutoencoderKL(cfg).eval().requires_grad_(False)
    from helpers import seeded_module_weights
    seeded_module_weights(okl, 31)
    original_decode = okl.decode

    class FakeSdModel:
        is_sdxl = False
        first_stage_model = okl
    m = FakeSdModel()
    z = seeded((1, 4, 8, 8), 3)
    with torch.no_grad():
        ref = original_decode(z)                           # the caller already divided by scale_factor
    eng1 = hook.install(m)
    assert okl.decode is not original_decode and okl._torch_decode == original_decode
    got = okl.decode(z.to(dev))
    assert got.dtype == torch.float32 and rel_l2(got.cpu(), ref) < 2.5e-3
    got16 = okl.decode(z.to(dev).half())                   # dtype_vae = fp16 callers get fp16 back
    assert got16.dtype == torch.float16
    # in-place weight change + second on_model_loaded: new engine, old one closed, original decode still remembered
    seeded_module_weights(okl, 32)
    with torch.no_grad():
        ref2 = original_decode(z)
    eng2 = hook.install(m)
    assert eng1.handle is None and eng2 is not eng1 and okl._torch_decode == original_decode
    got2 = okl.decode(z.to(dev))
    assert rel_l2(got2.cpu(), ref2) < 2.5e-3 and rel_l2(got2.cpu(), ref) > 1e-2
    hook.uninstall(m)
    assert okl.decode == original_decode and eng2.handle is None and not hasattr(okl, "_mi355x_engine")


def test_checkpoint_file_to_engine_and_external_vae(dev, tmp_path):
    """Row a15: a synthetic SD1.x-schema checkpoint written as .safetensors (with the OLD CLIP key layout the reference fixes up,
    modules/sd_models.py:243-281) -> read_state_dict -> load_model -> same UNet / VAE bits as the in-memory dict; then an external
    VAE file (modules/sd_vae.py:194-235) replaces the first stage and None restores the checkpoint's own."""
    import safetensors.torch
    schema, sd_models = sub("schema"), sub("sd_models")
    ucfg, vcfg = schema.tiny_unet(), schema.tiny_vae()
    sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16)
    disk = dict(sd)
    disk["cond_stage_model.transformer.embeddings.position_ids"] = torch.arange(77)[None]
    path = tmp_path / "tiny.safetensors"
    safetensors.torch.save_file({k: v.contiguous() for k, v in disk.items()}, str(path))
    a = sd_models.load_model(checkpoint_file=str(path), unet_cfg=ucfg, vae_cfg=vcfg, device=0)
    b = sd_models.SdModel(sd, ucfg, vcfg, device=0)
    assert "cond_stage_model.transformer.text_model.embeddings.position_ids" in a._checkpoint
    x, t, ctx = seeded((2, 4, 16, 16), 1).to(dev), torch.tensor([500.0, 10.0]).to(dev), seeded((2, 77, 64), 2).to(dev)
    assert torch.equal(a.engine.unet_forward(x, t, ctx), b.engine.unet_forward(x, t, ctx))
    z = seeded((2, 4, 16, 16), 5).to(dev)
    own = a.decode_first_stage(z)
    assert torch.equal(own, b.decode_first_stage(z))
    # external VAE with other weights
    sd2 = schema.synthetic_state_dict(None, vcfg, dtype=torch.float16, seed=0xABCD)
    ext = {k[len(schema.VAE_PREFIX):]: v.contiguous() for k, v in sd2.items() if k.startswith(schema.VAE_PREFIX)}
    ext["model_ema.decay"] = torch.zeros(1)
    vpath = tmp_path / "other.vae.safetensors"
    safetensors.torch.save_file(ext, str(vpath))
    a.load_vae(str(vpath), "from test")
    swapped = a.decode_first_stage(z)
    c = sd_models.SdModel({**sd, **sd2}, ucfg, vcfg, device=0)
    assert torch.equal(swapped, c.decode_first_stage(z)) and not torch.equal(swapped, own)
    a.load_vae(None)
    assert torch.equal(a.decode_first_stage(z), own) and a.loaded_vae_file is None
    for m in (a, b, c):
        m.engine.close()


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
        sampler_noise_scheduler_override, extra_generation_params = None, {}
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


def _bench_ranks(n, extra_env, args, timeout=600):
    """Launch bench.py as n ranks the way torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment);
    returns rank 0's JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r if "SDMI_DIST_BACKEND" not in extra_env else 0))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)] + args, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode(errors="replace")[-2000:] for o in outs]
    line = [ln for ln in outs[0][0].decode().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


BENCH_TINY = ["--model", "tiny", "--size", "128", "--sampler-steps", "3", "--batch", "3", "--steps", "1", "--warmup", "1",
              "--no-roofline", "--no-cpu-baseline", "--verify-shards"]


def test_bench_two_ranks_share_one_gpu_gathered_images_equal_rank_local_replays(dev):
    """The N > 1 path of bench.py end to end on ONE device: two processes (gloo for the collectives, both engines on cuda:0) — weights
    generated on rank 0 and broadcast, the global job sharded by process_images_sharded, uint8 images gathered on rank 0, barrier + max
    over ranks around the timed region — and --verify-shards: every rank's slice replayed on rank 0 equals the gathered images bit for
    bit.  What a 2-GPU box adds to this is RCCL instead of gloo (next test)."""
    out = _bench_ranks(2, {"SDMI_DIST_BACKEND": "gloo"}, BENCH_TINY)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 6 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["shard_check"] == "ok", out["config"]["shard_check"]
    assert out["config"]["weights_broadcast_ms"] > 0 and out["value"] > 0 and out["scaling"] == "weak"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_bench_two_gpus_rccl_gathered_images_equal_rank_local_replays(dev):
    """Same over RCCL, one rank per GPU (runs only where two devices are visible: the builder's boxes have one)."""
    out = _bench_ranks(2, {}, BENCH_TINY)
    assert out["n_gpus"] == 2 and out["config"]["shard_check"] == "ok" and out["config"]["weights_broadcast_ms"] > 0


def test_engine_sampler_progress_interrupt_live_preview_and_mask_blend_hooks(dev, monkeypatch):
    """What the reference's sampler loop does besides arithmetic (modules/sd_samplers_cfg_denoiser.py:157-158, 176-185, 295-304;
    modules/sd_samplers_common.py:256-281), on the engine samplers with the real kernels:
      * every step stores an x0 prediction through ``shared.store_latent`` — "Prompt" / "Negative prompt" / "Combined" previews obey
        combined = negative + cfg * (prompt - negative) (x0 is affine in the model output) for the sigma AND the timestep family,
        and the choice never changes the samples;
      * ``state.interrupted`` raised mid-run ends the job at the next denoiser call, returning the LAST "Prompt" x0 prediction;
      * a script with ``on_mask_blend`` sees every per-step blend (current, blended, denoiser, sigma) and the final one, and what it
        writes to ``blended_latent`` is what the sampler continues from."""
    schema, processing, shared, ss = sub("schema"), sub("processing"), sub("shared"), sub("sd_samplers")
    ucfg, vcfg = schema.tiny_unet(), schema.tiny_vae()
    sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16)
    model = sub("sd_models").SdModel(sd, ucfg, vcfg, device=0)
    g = torch.Generator().manual_seed(11)
    cond, uncond = torch.randn(2, 77, 64, generator=g), torch.randn(2, 77, 64, generator=g)

    def run(name, content, hook=None, steps=6):
        stored = []

        def store(decoded):
            stored.append(decoded.clone())
            shared.state.current_latent = decoded
            if hook is not None:
                hook(len(stored))
        monkeypatch.setattr(shared, "store_latent", store)
        monkeypatch.setattr(shared.opts, "live_preview_content", content)
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=model, c=cond, uc=uncond, seed=1000, batch_size=2, steps=steps,
                                                        cfg_scale=7.0, width=128, height=128, sampler_name=name)
        res = processing.process_images(p)
        return res.latents.clone(), stored

    try:
        for name in ("Euler a", "DDIM"):
            lat_c, combined = run(name, "Combined")
            lat_p, prompt = run(name, "Prompt")
            lat_n, negative = run(name, "Negative prompt")
            assert torch.equal(lat_c, lat_p) and torch.equal(lat_c, lat_n)
            assert len(combined) == len(prompt) == len(negative) == 6 and shared.state.sampling_steps == 6 and shared.state.sampling_step == 5
            for c, pr, ng in zip(combined, prompt, negative):
                assert rel_l2(c.cpu(), (ng + 7.0 * (pr - ng)).cpu()) < 2e-5, name
                assert not torch.equal(pr, ng)
            # interrupt after the third stored preview: three UNet evaluations happened, the fourth denoiser call raises
            def press_interrupt(n):
                if n == 3:
                    shared.state.interrupted = True
            lat_i, seen = run(name, "Prompt", hook=press_interrupt)
            shared.state.interrupted = False
            assert len(seen) == 3 and torch.equal(lat_i, seen[-1]) and torch.equal(seen[-1], prompt[2])
            assert shared.state.sampling_step == 2
    finally:
        shared.state.interrupted = False

    # ---- Script.on_mask_blend (soft inpainting's hook): img2img with a latent mask
    img = torch.rand((2, 3, 32, 32), generator=torch.Generator().manual_seed(11))
    mask = torch.zeros(2, 4, 16, 16)
    mask[:, :, 4:12, 2:9] = 1.0
    monkeypatch.setattr(shared, "store_latent", lambda d: None)

    def inpaint(name, runner):
        p = processing.StableDiffusionProcessingImg2Img(sd_model=model, c=cond, uc=uncond, seed=3000, batch_size=2, steps=5, cfg_scale=7.0,
                                                        width=128, height=128, sampler_name=name, init_images=img,
                                                        denoising_strength=0.6, latent_mask=mask)
        p.scripts = runner
        return processing.process_images(p).latents.clone()

    class Runner:
        def __init__(self, rewrite=None):
            self.seen, self.rewrite = [], rewrite

        def on_mask_blend(self, p, mba):
            self.seen.append((mba.is_final_blend, mba.denoiser, None if mba.sigma is None else tuple(mba.sigma.shape),
                              tuple(mba.current_latent.shape), tuple(mba.blended_latent.shape)))
            assert torch.equal(mba.blended_latent, mba.current_latent * mba.nmask + mba.init_latent * mba.mask) or \
                rel_l2(mba.blended_latent.cpu(), (mba.current_latent * mba.nmask + mba.init_latent * mba.mask).cpu()) < 1e-6
            if self.rewrite is not None:
                mba.blended_latent = self.rewrite(mba)

    for name in ("Euler a", "DDIM"):                          # blend after / before denoising
        fused = inpaint(name, None)
        watch = Runner()
        seen = inpaint(name, watch)
        assert rel_l2(seen.cpu(), fused.cpu()) < 1e-6, name   # same blends, one kernel later
        steps = [s for s in watch.seen if not s[0]]
        # Euler a: t_enc + 1 sigma intervals; DDIM: t_enc = 3 timesteps (sd_samplers_timesteps.py:101), of which ddim's loop runs
        # len(timesteps) - 1 (sd_samplers_timesteps_impl.py:22)
        n_eval = 4 if name == "Euler a" else 2
        assert len(steps) == n_eval and len(watch.seen) == n_eval + 1 and watch.seen[-1][0] and watch.seen[-1][1] is None
        assert all(isinstance(s[1], ss.CFGDenoiser) and s[2] == (2,) and s[3] == s[4] == (2, 4, 16, 16) for s in steps)
        unblended = inpaint(name, Runner(rewrite=lambda mba: mba.current_latent if not mba.is_final_blend else mba.blended_latent))
        assert rel_l2(unblended.cpu(), fused.cpu()) > 1e-3, name      # the script's word counts: no per-step blending happened


def test_sd_optimization_attnblock_forward_inside_torch_vae_module(dev):
    """mi355x_attnblock_forward bound to a torch AttnBlock (the oracle's class has ldm's attribute names norm / q / k / v / proj_out,
    modules/sd_hijack_optimizations.py:554-610): d = 512 single head through sdmi_attention_wide, at a ragged token count (N = 30*34 =
    1020, not a multiple of 64) and with a batch of 2 (one image's scores in the workspace at a time); fp16 and fp32 module dtypes."""
    from oracle import vae as ov
    from helpers import seeded_module_weights
    opt_mod = sub("sd_hijack_optimizations")
    blk = ov.AttnBlock(512).eval().requires_grad_(False)
    seeded_module_weights(blk, 5)
    x = seeded((2, 512, 30, 34), 7)
    with torch.no_grad():
        ref = blk(x)
        # fp16 module: GroupNorm, the q / k / v / proj_out 1x1 convs and the residual add run in torch fp16, and `got - x` is taken
        # from the fp16-rounded sum (one rounding of |x| against a branch a third its size): measured 4.3e-3 on the MI355X
        errs = {}
        for dt, tol in ((torch.float16, 6e-3), (torch.float32, 1e-3)):
            g = ov.AttnBlock(512).eval().requires_grad_(False)
            g.load_state_dict(blk.state_dict())
            g = g.to(dev, dt)
            g.forward = types.MethodType(opt_mod.mi355x_attnblock_forward, g)
            got = g(x.to(dev, dt))
            assert got.dtype == dt and got.shape == ref.shape
            err = rel_l2((got.float().cpu() - x), (ref - x))         # the attention branch alone (the residual x would mask it)
            print(f"[attnblock {dt}] branch rel-L2 {err:.3e}")
            errs[dt] = (err, tol)
        assert all(e < t for e, t in errs.values()), errs
    # the ops-level entry against plain softmax(q k^T) v, cross shapes (M != N)
    q, k, v = seeded((1, 200, 512), 1).half(), seeded((1, 333, 512), 2).half(), seeded((1, 333, 512), 3).half()
    want = torch.softmax(q.float() @ k.float().transpose(1, 2) * 512 ** -0.5, -1) @ v.float()
    got = sub("ops").attention(q.to(dev), k.to(dev), v.to(dev), heads=1)
    assert rel_l2(got.float().cpu(), want) < 1e-3
