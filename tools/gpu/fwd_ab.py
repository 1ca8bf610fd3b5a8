#!/usr/bin/env python3
"""Torch-free, same-process A/B of runtime knobs on the UNet forward of the headline workload (SD1.5, 64x64 latent, 16-row CFG batch).

    python tools/gpu/fwd_ab.py base conv_korder=0 "gn_cat=1" ... [--reps 3] [--fwd 10] [--profile] [--model sd15|sdxl|tiny] [--hw 64]
    python tools/gpu/fwd_ab.py base ... --what vae --rows 8          (the job's batched VAE decode instead of the UNet forward)

The C1 job is 20 such forwards + one VAE decode (the forwards are ~95 % of it), so a knob that moves the forward moves the job; what the
whole-job sweep (tools/gpu/knob_sweep.py) adds is the sampler and the decode, at the price of `import torch` (1-2 minutes on a fresh
GPU box — more than this whole script).  Nothing here imports torch: device memory comes from tools/gpu/hipmem.py (ctypes over
libamdhip64), weights are seeded numpy values in the schema's shapes with the schema's variances (not the bench's torch-seeded
values: outputs of this tool are compared with each other, never with the oracle).

Per setting: ms per forward (min / median over `--reps` rounds of `--fwd` timed forwards after one warm-up, settings interleaved),
and the output's relative L2 distance and exact-equality flag against the first setting.  `--profile` adds the HIP-event kernel-class
table of one forward per setting.  Settings use knob_sweep.py's syntax, plus `gemm_override=<M,N,K,taps,kind:cfg:split;...>` as the last
item of a setting (per-shape tile / split choices: what tools/gpu/shape_tune.py explores inside the job); `cfg_pairs` / `uniform_t` are engine options here
(default 1 / 1: what the samplers ask for).  (Does not import oracle/.)
"""
import argparse
import ctypes as C
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

hipmem = None                   # tools/gpu/hipmem.py, imported by main(): tools/cpu/fwd_parity.py imports this module for its weights only

PKG = "stable-diffusion-webui_amd"
FUSE_ROWS_DEFAULT = 2           # the engine's default for option "fuse_rows" (engine.h)
ENGINE_OPTS = ("ln_fold", "streams", "arena_reuse", "cfg_pairs", "uniform_t", "gn_cat", "fuse_rows", "residual_fp32")
DEFAULTS = {"gemm_cfg": -1, "gemm_shortk_cfg": -1, "gemm_shortk_maxk": 448, "gemm_geglu_cfg": -1, "vt_mode": 1, "attn_kvt": 0, "attn_occ": 15, "attn_tau": 8, "attn_fold_min_m": 1024,
            "tile_order": -1, "conv_korder": -1, "small_linear_lds": 1, "gemm_split": 0, "gemm_pipe": -1, "gemm_lin": 1, "gn_fuse": 1, "gn_small": 1, "ep_wide": 1,
            "gemm_dbgflags": 0, "ln_fold": 0, "streams": 1, "arena_reuse": 0, "cfg_pairs": 1, "uniform_t": 1, "gn_cat": 0, "fuse_rows": FUSE_ROWS_DEFAULT, "residual_fp32": 0}
assert all(k in DEFAULTS for k in ENGINE_OPTS if k in ("gn_cat", "cfg_pairs", "uniform_t", "ln_fold", "streams", "arena_reuse"))     # every option a setting may switch is reset by the next one


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


def classify(name):
    import re
    if name.startswith("gemm_mfma"):
        if "conv3x3" in name:
            return "conv3x3"
        if "geglu" in name:
            return "1x1_geglu"
        if "_tr" in name:
            return "1x1_vt"
        if re.search(r" x\d+$", name):
            return "1x1_batched"
        return "1x1"
    for p in ("attention_mfma_self", "attention_mfma_cross", "groupnorm", "layernorm", "splitk", "small_linear", "rowchain_ff"):
        if name.startswith(p):
            return p
    return "other"


_scaled = {}


def synthetic_weight(pool, key, shape, kind):
    """fp16 values in `shape` with the schema's variance for this kind of tensor (schema.synthetic_state_dict), cut at a key-dependent
    offset from one seeded pool: generating (and converting) 860 M independent normals would cost more than the measurement.  The
    fp16 pool is kept per distinct scale (a few dozen fan-ins), so a tensor is one slice copy."""
    n = int(np.prod(shape))
    if kind == "w":
        mul, add = float(np.prod(shape[1:])) ** -0.5, 0.0
    elif kind == "e":
        mul, add = 0.5, 0.0
    elif kind == "g":
        mul, add = 0.02, 1.0
    else:
        mul, add = 0.02, 0.0
    p16 = _scaled.get((mul, add))
    if p16 is None:
        tmp = _scaled.setdefault("tmp", np.empty_like(pool))       # (one scratch buffer: fresh 64 MB temporaries cost a page fault each)
        np.multiply(pool, np.float32(mul), out=tmp)
        if add:
            tmp += np.float32(add)
        p16 = _scaled[(mul, add)] = tmp.astype(np.float16)
    if n >= p16.size:
        return np.ascontiguousarray(np.resize(p16, n).reshape(shape))
    off = hash_u32(key) % (p16.size - n)
    return np.ascontiguousarray(p16[off:off + n].reshape(shape))


def hash_u32(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--fwd", type=int, default=10)
    ap.add_argument("--model", default="sd15", choices=("sd15", "sdxl", "tiny"))
    ap.add_argument("--rows", type=int, default=16, help="UNet batch rows (the CFG batch: 2 x images)")
    ap.add_argument("--hw", type=int, default=64, help="latent height = width")
    ap.add_argument("--what", default="unet", choices=("unet", "vae"), help="unet: one UNet forward of --rows rows; vae: one batched VAE decode of --rows latents")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fwd_ab.json"))
    ap.add_argument("--dump", default=None, help="write inputs and every setting's output to this .npz (tools/cpu/fwd_parity.py compares them with "
                                                 "the fp32 CPU oracle on the same weights; use --rows 2: the oracle forward is CPU work)")
    return ap


def build(args):
    """Engine + weights + inputs for ``args`` (what / model / rows / hw): a namespace with forward(), apply(setting), profile_once() and
    the buffers — shared by main() below and tools/gpu/shape_screen.py."""
    global hipmem
    import hipmem
    t_start = time.time()
    _lib = sub("_lib")
    _lib.require_device()
    lib = _lib.lib
    schema = sub("schema")
    hipmem.set_device(0)
    handle = lib.sdmi_engine_create(0)
    if not handle:
        raise SystemExit("sdmi_engine_create failed: " + _lib.last_error())
    rng = np.random.default_rng(0x5D15)
    pool = rng.standard_normal(1 << 22, dtype=np.float32)                 # (larger tensors tile it)
    nparam = 0
    B, hw, L = args.rows, args.hw, 77
    x = t = ctx = y = None
    if args.what == "vae":
        # the batched VAE decode of the job's end (decode_latent_batch): B latents -> B x 3 x 8h x 8w fp32 images
        vcfg = {"sd15": schema.sd15_vae, "sdxl": schema.sdxl_vae, "tiny": schema.tiny_vae}[args.model]()
        vc = _lib.VAEConfigC()                                # the ctypes mirror of engine._vae_cfg_c
        vc.ch, vc.num_levels = vcfg.ch, len(vcfg.ch_mult)
        for i, m in enumerate(vcfg.ch_mult):
            vc.ch_mult[i] = m
        vc.num_res_blocks, vc.in_channels, vc.out_ch, vc.z_channels = vcfg.num_res_blocks, vcfg.in_channels, vcfg.out_ch, vcfg.z_channels
        vc.scale_factor = vcfg.scale_factor
        _lib.check(lib.sdmi_vae_configure(handle, C.byref(vc)), "vae_configure")
        for key, shape, kind in schema.vae_schema(vcfg):
            w = synthetic_weight(pool, key, tuple(shape), kind)
            nparam += w.size
            shp = (C.c_int64 * w.ndim)(*w.shape)
            _lib.check(lib.sdmi_vae_load_tensor(handle, key.encode(), C.c_void_p(w.ctypes.data), _lib.F16, w.ndim, shp, 0), f"vae_load_tensor({key})")
        _lib.check(lib.sdmi_vae_finalize(handle), "vae_finalize")
        t_loaded = time.time()
        f = 2 ** (len(vcfg.ch_mult) - 1)
        x = rng.standard_normal((B, vcfg.z_channels, hw, hw), dtype=np.float32)
        dx = hipmem.DevBuf.from_numpy(x)
        out_shape = (B, vcfg.out_ch, hw * f, hw * f)
        dout = hipmem.DevBuf(int(np.prod(out_shape)) * 4)

        def forward():
            _lib.check(lib.sdmi_vae_decode(handle, dx.ptr, _lib.F32, dout.ptr, B, hw, hw, None), "vae_decode")
    else:
        cfg = {"sd15": schema.sd15_unet, "sdxl": schema.sdxl_unet, "tiny": schema.tiny_unet}[args.model]()
        c = _lib.UNetConfigC()                                # the ctypes mirror of engine._unet_cfg_c (engine.py imports torch)
        c.in_channels, c.out_channels, c.model_channels = cfg.in_channels, cfg.out_channels, cfg.model_channels
        c.num_levels = len(cfg.channel_mult)
        ds = 1
        for i, m in enumerate(cfg.channel_mult):
            c.channel_mult[i] = m
            c.attn_level[i] = 1 if ds in cfg.attention_resolutions else 0
            c.transformer_depth[i] = cfg.depth_at(i)
            ds *= 2
        c.num_res_blocks, c.num_heads, c.num_head_channels = cfg.num_res_blocks, cfg.num_heads, cfg.num_head_channels
        c.context_dim, c.adm_in_channels = cfg.context_dim, cfg.adm_in_channels or 0
        _lib.check(lib.sdmi_unet_configure(handle, C.byref(c)), "unet_configure")
        for key, shape, kind in schema.unet_schema(cfg):
            w = synthetic_weight(pool, key, tuple(shape), kind)
            nparam += w.size
            shp = (C.c_int64 * w.ndim)(*w.shape)
            _lib.check(lib.sdmi_unet_load_tensor(handle, key.encode(), C.c_void_p(w.ctypes.data), _lib.F16, w.ndim, shp, 0), f"load_tensor({key})")
        _lib.check(lib.sdmi_unet_finalize(handle), "unet_finalize")
        t_loaded = time.time()
        half = B // 2
        lat = rng.standard_normal((half, cfg.in_channels, hw, hw), dtype=np.float32)
        x = np.concatenate([lat, lat], 0).astype(np.float32)                  # the CFG denoiser's [cond | uncond] batch: same latent twice
        t = np.full((B,), 601.0, dtype=np.float32)
        ctx = rng.standard_normal((B, L, cfg.context_dim), dtype=np.float32)
        y = rng.standard_normal((B, cfg.adm_in_channels), dtype=np.float32) if cfg.adm_in_channels else None
        dx, dt, dctx = hipmem.DevBuf.from_numpy(x), hipmem.DevBuf.from_numpy(t), hipmem.DevBuf.from_numpy(ctx)
        dy = hipmem.DevBuf.from_numpy(y) if y is not None else None
        out_shape = (B, cfg.out_channels, hw, hw)
        dout = hipmem.DevBuf(int(np.prod(out_shape)) * 4)
        _lib.check(lib.sdmi_unet_set_context(handle, dctx.ptr, _lib.F32, B, L, None), "set_context")

        def forward():
            _lib.check(lib.sdmi_unet_forward(handle, dx.ptr, dt.ptr, None, dy.ptr if dy else None, dout.ptr, _lib.F32, B, hw, hw, L, None), "unet_forward")

    def apply(setting):
        vals = dict(DEFAULTS)
        override = ""
        if "gemm_override=" in setting:                       # string knob, always LAST in a setting (its value holds , : ;):
            setting, override = setting.split("gemm_override=", 1)      #   "conv_korder=0,gemm_override=4096,1280,1280,1,0:9:1;..."
            setting = setting.rstrip(",") or "base"
        if lib.sdmi_debug_set_str(b"gemm_override", override.encode()) and override:
            _lib.check(1, "gemm_override")
        if setting != "base":
            for kv in setting.split(","):
                k, v = kv.split("=")
                vals[k] = int(v)
        for k, v in vals.items():
            if k in ENGINE_OPTS:
                rc = lib.sdmi_engine_set_option(handle, k.encode(), int(v))
            else:
                rc = lib.sdmi_debug_set(k.encode(), int(v))
            if rc and v == DEFAULTS.get(k):                   # an older library (SDMI_LIB two-builds A/B) does not know this knob
                continue
            _lib.check(rc, k)
        # cached K / V^T of the text context depend on nothing a knob changes — except "fuse_rows", whose per-image matrices live beside them
        if args.what == "unet":
            _lib.check(lib.sdmi_unet_set_context(handle, dctx.ptr, _lib.F32, B, L, None), "set_context")


    def profile_once():
        """HIP-event kernel list of ONE forward under the current setting (one untimed forward first)."""
        forward()
        hipmem.sync()
        _lib.check(lib.sdmi_profile_begin(), "profile_begin")
        forward()
        hipmem.sync()
        buf = C.create_string_buffer(1 << 21)
        _lib.check(lib.sdmi_profile_end(buf, len(buf)), "profile_end")
        return json.loads(buf.value.decode())["kernels"]

    import types
    return types.SimpleNamespace(lib=lib, _lib=_lib, handle=handle, forward=forward, apply=apply, profile_once=profile_once, dout=dout,
                                 out_shape=out_shape, x=x, t=t, ctx=ctx, y=y, nparam=nparam, t_start=t_start, t_loaded=t_loaded, B=B, hw=hw)


def main(argv=None):
    args = parser().parse_args(argv)
    env = build(args)
    lib, _lib, handle, forward, apply, dout, out_shape = env.lib, env._lib, env.handle, env.forward, env.apply, env.dout, env.out_shape
    x, t, ctx, y, nparam, t_start, t_loaded, B, hw = env.x, env.t, env.ctx, env.y, env.nparam, env.t_start, env.t_loaded, env.B, env.hw
    e0, e1 = hipmem.Event(), hipmem.Event()
    times = {s: [] for s in args.settings}
    outs, failed = {}, {}
    res = {"model": args.model, "what": args.what, "rows": B, "latent": hw, "params": nparam, "load_s": round(t_loaded - t_start, 1), "settings": {}}

    def save():
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)

    for rep in range(args.reps):
        for s in args.settings:
            if s in failed:
                continue
            try:                                              # a setting the library refuses is reported and skipped; the others still run
                apply(s)
                forward()
                hipmem.sync()
                if s not in outs:
                    outs[s] = dout.to_numpy(np.float32, out_shape).copy()
                e0.record()
                for _ in range(args.fwd):
                    forward()
                e1.record()
                times[s].append(e1.ms_since(e0) / args.fwd)
                if rep == 0:                                  # (kept on disk as they come: a later setting may take the process down)
                    res["settings"][s] = {"ms_first_round": round(times[s][0], 3), "finite": bool(np.isfinite(outs[s]).all())}
                    save()
            except (_lib.SdmiError, RuntimeError) as ex:
                failed[s] = str(ex)
                res["settings"][s] = {"error": str(ex)}
                print(f"{s:48s} FAILED: {ex}", flush=True)
                save()
    base = next((s for s in args.settings if s not in failed), None)
    prev = None
    for s in args.settings:
        if s in failed:
            continue
        a, b = outs[s].astype(np.float64), outs[base].astype(np.float64)
        rel = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        rel_prev = None if prev is None else float(np.linalg.norm(a - outs[prev]) / max(np.linalg.norm(outs[prev].astype(np.float64)), 1e-30))
        prev = s
        finite = bool(np.isfinite(outs[s]).all())
        r = {"ms_min": round(min(times[s]), 3), "ms_median": round(statistics.median(times[s]), 3), "all": [round(v, 3) for v in times[s]],
             "rel_l2_vs_first": rel, "rel_l2_vs_previous_setting": rel_prev, "identical_to_first": bool(np.array_equal(outs[s], outs[base])), "finite": finite,
             "out_rms": float(np.sqrt(np.mean(a * a)))}
        res["settings"][s] = r
        print(f"{s:48s} min {r['ms_min']:8.3f} ms  median {r['ms_median']:8.3f} ms  rel-L2 vs {base}: {rel:.3e}" + (f" vs previous: {rel_prev:.3e}" if rel_prev is not None else "") +
              f"{'  (bit-identical)' if r['identical_to_first'] else ''}{'' if finite else '  NON-FINITE OUTPUT'}", flush=True)
    save()
    if args.dump:
        ok = [s for s in args.settings if s not in failed]
        z0 = np.zeros((0,), np.float32)
        np.savez_compressed(args.dump, x=x, t=(t if t is not None else z0), ctx=(ctx if ctx is not None else z0), y=(y if y is not None else z0),
                            what=np.array(args.what), settings=np.array(ok),
                            model=np.array(args.model), **{f"out_{i}": outs[s] for i, s in enumerate(ok)})
        print(f"dumped inputs + {len(ok)} outputs -> {args.dump}", flush=True)
    if args.profile:
        for s in args.settings:
            if s in failed:
                continue
            apply(s)
            kernels = env.profile_once()
            groups = {}
            for k in kernels:
                g = groups.setdefault(classify(k["name"]), [0.0, 0.0, 0])
                g[0] += k["ms"]; g[1] += k["flops"]; g[2] += k["launches"]
            res["settings"][s]["profile_ms"] = {k: round(v[0], 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0])}
            res["settings"][s]["profile_launches"] = {k: v[2] for k, v in groups.items()}
            res["settings"][s]["kernels"] = sorted(kernels, key=lambda k: -k["ms"])          # per launch shape: name, launches, ms, flops, bytes
            print(f"--- {s}: {sum(v[0] for v in groups.values()):.2f} ms of kernels in {sum(v[2] for v in groups.values())} launches")
            for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0]):
                print(f"    {k:24s} {v[0]:8.3f} ms  {v[2]:4d} launches" + (f"  {v[1] / v[0] / 1e9:7.1f} TFLOP/s" if v[1] > 0 and v[0] > 0 else ""))
            save()
    res["wall_s"] = round(time.time() - t_start, 1)
    save()
    print(f"load {res['load_s']} s, total {res['wall_s']} s -> {args.out}", flush=True)
    lib.sdmi_engine_destroy(handle)


if __name__ == "__main__":
    main()
