#!/usr/bin/env python3
"""In-engine autotuner of the GEMM tile configuration / split-K factor per exact shape.

Every candidate of every target shape is timed INSIDE the whole C1 job (per-launch HIP events, sdmi_profile_*), i.e. with its
operands in the cache state the engine really leaves them in; candidate k of all shapes is applied together in "round" k through
sdmi_debug_set_str("gemm_override", ...), so a few dozen one-job rounds cover the whole search.  The winners are verified by a
same-process A/B of the whole job (interleaved repetitions) and written as the table csrc/gemm_tuned_shapes.inc + a report.

    python tools/gpu/shape_tune.py --top 28 --emit        (does not import oracle/)
"""
import argparse
import ctypes
import importlib
import json
import os
import re
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
CFG_BN = {0: 128, 2: 64, 3: 128, 4: 256, 5: 320, 6: 128, 7: 64, 8: 320, 9: 160}
CFG_NAME = {0: "128x128", 2: "64x64", 3: "128x128k32", 4: "256x256", 5: "256x320", 6: "256x128", 7: "128x64", 8: "128x320", 9: "128x160"}


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


def parse(name):
    """kernel profile name -> (M, N, K, taps, kind) or None (batched launches and non-GEMM kernels are skipped)."""
    if not name.startswith("gemm_mfma") or re.search(r" x\d+$", name):
        return None
    m = re.search(r" M(\d+) N(\d+) K(\d+)", name)
    if not m:
        return None
    head = name.split(" ")[0]
    taps = 9 if "conv3x3" in head else 1
    kind = 1 if "_geglu" in head else 2 if "_tr" in head else 0
    return (int(m.group(1)), int(m.group(2)), int(m.group(3)), taps, kind)


def candidates(key):
    M, N, K, taps, kind = key
    out = []
    for cfg, bn in CFG_BN.items():
        if N % bn:
            continue
        if kind == 1 and cfg in (5, 6, 8, 9):
            continue
        splits = [1]
        if kind == 0 and K >= 64 * 16 and ((M + 127) // 128) * ((N + 127) // 128) < 512:      # a split-K workspace exists for these
            splits += [s for s in (2, 3, 4, 6, 8) if K // 64 // s >= 4]
        out += [(cfg, s) for s in splits]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=28)
    ap.add_argument("--emit", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sampler-steps", type=int, default=20)
    ap.add_argument("--model", default="sd15", choices=["sd15", "sdxl"])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--hires", action="store_true", help="txt2img + hires-fix x2 (the c4a job) instead of plain txt2img")
    ap.add_argument("--tag", default="r03_shape_tuning")
    args = ap.parse_args()
    lib = sub("_lib")
    lib.require_device()
    schema, sd_models, processing = sub("schema"), sub("sd_models"), sub("processing")
    ucfg, vcfg = (schema.sdxl_unet(), schema.sdxl_vae()) if args.model == "sdxl" else (schema.sd15_unet(), schema.sd15_vae())
    sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16)
    model = sd_models.SdModel(sd, ucfg, vcfg, device=0, vae_decoder_only=True)
    del sd
    g = torch.Generator().manual_seed(50_000)
    B = args.batch
    c, uc = torch.randn(B, 77, ucfg.context_dim, generator=g).cuda(), torch.randn(B, 77, ucfg.context_dim, generator=g).cuda()
    y = uy = None
    if ucfg.adm_in_channels:
        y, uy = torch.randn(B, ucfg.adm_in_channels, generator=g).cuda(), torch.randn(B, ucfg.adm_in_channels, generator=g).cuda()

    def job():
        kw = dict(enable_hr=True, hr_scale=2.0, hr_upscaler="Latent", denoising_strength=0.75) if args.hires else {}
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=model, c=c, uc=uc, seed=1000, batch_size=B, n_iter=1, steps=args.sampler_steps,
                                                        cfg_scale=7.0, width=args.size, height=args.size, sampler_name="Euler a", keep_latents=False, **kw)
        if y is not None:
            p.y, p.uy = y, uy
        return processing.process_images(p)

    def set_override(table):
        spec = ";".join(f"{k[0]},{k[1]},{k[2]},{k[3]},{k[4]}:{cfg}:{split}" for k, (cfg, split) in table.items())
        lib.check(lib.lib.sdmi_debug_set_str(b"gemm_override", spec.encode()), "gemm_override")

    def profiled():
        job()
        lib.check(lib.lib.sdmi_profile_begin(), "profile_begin")
        job()
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 21)
        lib.check(lib.lib.sdmi_profile_end(buf, len(buf)), "profile_end")
        per = {}
        for k in json.loads(buf.value.decode())["kernels"]:
            key = parse(k["name"])
            if key is None:
                continue
            a = per.setdefault(key, [0.0, 0, k["name"].split(" ")[0]])
            a[0] += k["ms"]; a[1] += k["launches"]
            if "splitk" in k["name"]:
                a[2] = k["name"].split(" ")[0]
        return per

    set_override({})
    base = profiled()
    targets = sorted(base, key=lambda k: -base[k][0])[:args.top]
    cands = {k: candidates(k) for k in targets}
    rounds = max(len(v) for v in cands.values())
    print(f"{len(targets)} target shapes, {rounds} rounds", flush=True)
    results = {k: {} for k in targets}
    for r in range(rounds):
        table = {k: cands[k][r] for k in targets if r < len(cands[k])}
        set_override(table)
        try:
            per = profiled()
        except Exception as e:                                  # a candidate the library refuses: skip the round
            print("round", r, "failed:", e, flush=True)
            continue
        for k, choice in table.items():
            if k in per:
                results[k][choice] = (per[k][0], per[k][2])
    # split-K launches are followed by a reduce kernel that the per-shape time does not include: charge 12 us per launch
    best = {}
    report = [f"# In-engine per-shape GEMM tuning (tools/gpu/shape_tune.py): {args.model} {args.size}x{args.size} batch {args.batch}{' + hires x2' if args.hires else ''}, {args.sampler_steps} steps\n",
              "ms = time of all launches of the shape in one profiled job (HIP events); split-K candidates are charged 12 us per launch for the reduce kernel.\n",
              "| shape (M, N, K, taps, kind) | launches | default | ms | best | ms | gain ms | all candidates (cfg/split: ms) |", "|---|---:|---|---:|---|---:|---:|---|"]
    total_gain = 0.0
    for k in targets:
        n = base[k][1]
        scored = {ch: ms + (0.012 * n if ch[1] > 1 else 0.0) for ch, (ms, _) in results[k].items()}
        if not scored:
            continue
        b = min(scored, key=scored.get)
        dflt = base[k][0] + (0.012 * n if "splitk" in base[k][2] else 0.0)
        gain = dflt - scored[b]
        if gain > 0.02 * dflt and gain > 0.05:
            best[k] = b
            total_gain += gain
        report.append(f"| {k} | {n} | {base[k][2]} | {dflt:.2f} | {CFG_NAME[b[0]]}/{b[1]} | {scored[b]:.2f} | {gain:.2f} | " +
                      ", ".join(f"{CFG_NAME[ch[0]]}/{ch[1]}: {v:.2f}" for ch, v in sorted(scored.items(), key=lambda t: t[1])[:6]) + " |")
    print("\n".join(report), flush=True)
    print(f"predicted gain {total_gain:.1f} ms per job over {len(best)} shapes", flush=True)
    # ---- verification: whole-job A/B, interleaved
    times = {"default": [], "tuned": []}
    for _ in range(args.reps):
        for name, table in (("default", {}), ("tuned", best)):
            set_override(table)
            job()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(2):
                job()
            torch.cuda.synchronize()
            times[name].append((time.time() - t0) / 2 * 1e3)
    set_override({})
    d, t = min(times["default"]), min(times["tuned"])
    report.append(f"\nWhole-job A/B (min of {args.reps} interleaved rounds of 2 jobs): default {d:.2f} ms, tuned {t:.2f} ms ({(d - t) / d * 100:+.2f} %).")
    print(report[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"{args.tag}.md"), "w").write("\n".join(report) + "\n")
    inc = ["// generated by tools/gpu/shape_tune.py --emit (measured inside the C1 job on an MI355X; profiles/r03_shape_tuning.md)"]
    for k, (cfg, split) in best.items():
        inc.append(f"    {{{{{k[0]}, {k[1]}, {k[2]}, {k[3]}, {k[4]}}}, {cfg}, {split}}},")
    open(os.path.join(ROOT, "gpurun_out", f"gemm_tuned_shapes_{args.tag}.inc"), "w").write("\n".join(inc) + "\n")
    json.dump({"times": times, "best": {str(k): v for k, v in best.items()}}, open(os.path.join(ROOT, "gpurun_out", f"{args.tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
