#!/usr/bin/env python3
"""Same-box, same-process A/B of runtime tuning knobs (sdmi_debug_set) on the whole C1 job.

    python tools/gpu/knob_sweep.py "base" "attn_occ=5" "gemm_shortk_cfg=9" "gemm_shortk_cfg=9,gemm_shortk_maxk=1300" ...

The model is built once; every setting runs `--reps` rounds of (1 warm-up + `--jobs` timed jobs), settings interleaved round by
round so that clock / thermal drift hits all of them alike.  Prints ms per job (min and median over rounds) and, with
--profile, the per-kernel-class HIP-event breakdown of one job per setting.  (Does not import oracle/.)
"""
import argparse
import ctypes
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
KNOBS = ("tile_order", "conv_korder", "small_linear_lds", "gemm_cfg", "gemm_shortk_cfg", "gemm_shortk_maxk", "gemm_geglu_cfg", "vt_mode", "attn_kvt", "attn_occ", "attn_tau", "attn_fold_min_m", "gemm_split", "gemm_pipe", "gemm_lin", "gn_fuse", "gn_small", "ep_wide", "gemm_dbgflags",
         )
ENGINE_OPTS = ("ln_fold", "streams", "arena_reuse")
DEFAULTS = {"gemm_cfg": -1, "gemm_shortk_cfg": -1, "gemm_shortk_maxk": 448, "gemm_geglu_cfg": -1, "vt_mode": 1, "attn_kvt": 0, "attn_occ": 15, "attn_tau": 8, "attn_fold_min_m": 1024, "tile_order": -1, "conv_korder": -1, "small_linear_lds": 1, "cfg_pairs": 1,
            "gemm_split": 0, "gemm_pipe": -1, "gemm_lin": 1, "gn_fuse": 1, "gn_small": 1, "ep_wide": 1, "gemm_dbgflags": 0, "ln_fold": 0, "streams": 1, "arena_reuse": 0}


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
    for p in ("attention_mfma_self", "attention_mfma_cross", "groupnorm", "layernorm", "splitk", "small_linear"):
        if name.startswith(p):
            return p
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--jobs", type=int, default=2)
    ap.add_argument("--sampler-steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "knob_sweep.json"))
    args = ap.parse_args()
    lib = sub("_lib")
    lib.require_device()
    schema, sd_models, processing = sub("schema"), sub("sd_models"), sub("processing")
    ucfg, vcfg = schema.sd15_unet(), schema.sd15_vae()
    sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16)
    model = sd_models.SdModel(sd, ucfg, vcfg, device=0, vae_decoder_only=True)
    del sd
    g = torch.Generator().manual_seed(50_000)
    c, uc = torch.randn(args.batch, 77, 768, generator=g).cuda(), torch.randn(args.batch, 77, 768, generator=g).cuda()

    def job():
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=model, c=c, uc=uc, seed=1000, batch_size=args.batch, n_iter=1,
                                                        steps=args.sampler_steps, cfg_scale=7.0, width=512, height=512,
                                                        sampler_name="Euler a", keep_latents=False)
        return processing.process_images(p)

    def apply(setting):
        vals = dict(DEFAULTS)
        if setting != "base":
            for kv in setting.split(","):
                k, v = kv.split("=")
                vals[k] = int(v)
        for k, v in vals.items():
            if k == "cfg_pairs":                               # read by Engine.unet_forward per call (the CFG denoiser asks for it)
                sub("engine").CFG_PAIRS = bool(v)
                continue
            if k in ENGINE_OPTS:
                try:
                    model.engine.set_option(k, v)
                except Exception:
                    if v != DEFAULTS.get(k):
                        raise
            else:
                rc = lib.lib.sdmi_debug_set(k.encode(), int(v))
                if rc and v == DEFAULTS.get(k):               # an older library (two-builds A/B through SDMI_LIB) does not know this knob
                    continue
                lib.check(rc, k)

    times = {s: [] for s in args.settings}
    first_img = {}
    for rep in range(args.reps):
        for s in args.settings:
            apply(s)
            res = job()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(args.jobs):
                res = job()
            torch.cuda.synchronize()
            times[s].append((time.time() - t0) / args.jobs * 1e3)
            if s not in first_img:
                first_img[s] = res.images[0].astype("int32")
    base = args.settings[0]
    out = {}
    for s in args.settings:
        d = float(abs(first_img[s] - first_img[base]).mean())
        out[s] = {"ms_min": round(min(times[s]), 2), "ms_median": round(statistics.median(times[s]), 2), "all": [round(t, 2) for t in times[s]],
                  "u8_mean_abs_diff_vs_first_setting": round(d, 4)}
        print(f"{s:60s} min {out[s]['ms_min']:8.2f} ms  median {out[s]['ms_median']:8.2f} ms  img/s {args.batch / out[s]['ms_min'] * 1e3:6.2f}  "
              f"(u8 diff vs {base}: {d:.4f})", flush=True)
    if args.profile:
        for s in args.settings:
            apply(s)
            job()
            lib.check(lib.lib.sdmi_profile_begin(), "profile_begin")
            job()
            torch.cuda.synchronize()
            buf = ctypes.create_string_buffer(1 << 21)
            lib.check(lib.lib.sdmi_profile_end(buf, len(buf)), "profile_end")
            kernels = json.loads(buf.value.decode())["kernels"]
            groups = {}
            for k in kernels:
                a = groups.setdefault(classify(k["name"]), [0.0, 0.0, 0])
                a[0] += k["ms"]; a[1] += k["flops"]; a[2] += k["launches"]
            tot = sum(a[0] for a in groups.values())
            out[s]["profile_ms"] = {g_: round(a[0], 2) for g_, a in sorted(groups.items(), key=lambda t: -t[1][0])}
            print(f"--- {s}: {tot:.1f} ms of kernels")
            for g_, a in sorted(groups.items(), key=lambda t: -t[1][0]):
                print(f"    {g_:24s} {a[0]:8.2f} ms  n={a[2]:5d}  {a[1] / a[0] / 1e9 if a[0] else 0:8.1f} TFLOP/s")
            out[s]["kernels"] = sorted(kernels, key=lambda k: -k["ms"])[:40]
            out[s]["conv3x3"] = sorted((k for k in kernels if classify(k["name"]) == "conv3x3"), key=lambda k: -k["ms"])
            for k in kernels:
                if k["name"].startswith("small_linear"):
                    print(f"    {k['name']:40s} n={k['launches']:4d} {k['ms'] / k['launches'] * 1e3:8.1f} us / launch")
    if args.profile and len(args.settings) > 1:              # per-shape 3x3 conv times, first setting against the others
        import re
        shape = lambda n: re.search(r"M\d+ N\d+ K\d+", n).group(0)
        base_k = {}
        for k in out[base]["conv3x3"]:
            base_k[shape(k["name"])] = base_k.get(shape(k["name"]), 0.0) + k["ms"]
        for s in args.settings[1:]:
            print(f"--- 3x3 convs per shape: {base} -> {s}")
            cur = {}
            for k in out[s]["conv3x3"]:
                e = cur.setdefault(shape(k["name"]), [0.0, k["name"].split(" ")[0], k["launches"]])
                e[0] += k["ms"]
            for sh, (ms, nm, n) in sorted(cur.items(), key=lambda t: -t[1][0]):
                b = base_k.get(sh, 0.0)
                print(f"    {sh:28s} n={n:4d} {b:8.2f} -> {ms:8.2f} ms  ({(ms / b - 1) * 100 if b else 0:+6.1f} %)  {nm}")
    apply("base")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
