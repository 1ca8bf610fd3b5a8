#!/usr/bin/env python3
"""bench.py — images/sec of the txt2img hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of synthetic input = one full txt2img job of the workload
(SD1.5, 512x512, 20-step Euler-a, batch 8 per GPU, cfg 7): Philox noise -> 20 x (CFG batch build, UNet on 16 latents,
CFG combine, Euler-ancestral update) -> batched VAE decode -> uint8 HWC, everything resident in HBM (synthetic fp16
weights in the SD1.5 state-dict schema, seeded N(0,1) conditioning; the text encoder is outside the path).
Weak scaling: every rank runs the same per-GPU batch on its own images (seeds 1000 + global index), no per-step
communication; weights are generated on rank 0 and broadcast over RCCL before the timed region.

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     dominant kernel (implicit-GEMM MFMA conv/linear family) achieved TFLOP/s vs the 2.5 PFLOP/s dense fp16 MFMA
                  peak, measured with per-launch HIP events on the launch stream in a separate profiled pass
  "cpu_baseline": the fp32 CPU oracle (restated reference path) timed on this host on a bounded sample.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
UNET_TFLOP_PER_SAMPLE = 0.8033     # SD1.5 UNet forward @64x64 latent, per sample
VAE_TFLOP_PER_IMAGE = 2.5145       # VAE decode @512^2


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed jobs (one job = batch of images through the whole path)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per job")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--sampler-steps", type=int, default=20)
    ap.add_argument("--sampler", default="Euler a")
    ap.add_argument("--model", default="sd15", choices=["sd15", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-threads", type=int, default=0)
    return ap.parse_args()


def make_job(args, model, rank, world):
    processing = sub("processing")
    ctx_dim = model.unet_cfg.context_dim
    n = args.batch
    lo = rank * n                                                 # global image index of this rank's first image
    conds, unconds = [], []
    for i in range(lo, lo + n):
        g = torch.Generator().manual_seed(50_000 + i)             # "synthetic prompt" i
        conds.append(torch.randn(77, ctx_dim, generator=g))
        unconds.append(torch.randn(77, ctx_dim, generator=g))
    c, uc = torch.stack(conds).cuda(), torch.stack(unconds).cuda()

    def run_once():
        p = processing.StableDiffusionProcessingTxt2Img(
            sd_model=model, c=c, uc=uc, seed=1000 + lo, batch_size=n, n_iter=1, steps=args.sampler_steps, cfg_scale=7.0,
            width=args.size, height=args.size, sampler_name=args.sampler, keep_latents=False)
        return processing.process_images(p)
    return run_once


def roofline_block(args, run_once):
    """Profiled pass (separate from the timed region): HIP events around every launch on its stream."""
    lib = sub("_lib")
    lib.check(lib.lib.sdmi_profile_begin(), "profile_begin")
    run_once()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 20)
    lib.check(lib.lib.sdmi_profile_end(buf, len(buf)), "profile_end")
    kernels = json.loads(buf.value.decode())["kernels"]
    fam = [k for k in kernels if k["name"].startswith("gemm_mfma")]
    if not fam:
        return None, kernels
    tot_ms = sum(k["ms"] for k in fam)
    tot_fl = sum(k["flops"] for k in fam)
    launches = sum(k["launches"] for k in fam)
    dom = max(fam, key=lambda k: k["ms"])
    achieved = tot_fl / (tot_ms * 1e-3) / 1e12
    pmc = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path)).get("gemm_mfma_bytes_per_launch")
        except Exception:
            pmc = None
    block = {
        "bound": "mfma", "kernel": "gemm_mfma_kernel (implicit-GEMM conv3x3 / 1x1 / linear, all tile configs)",
        "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
        "launches_per_job": launches, "avg_launch_ms": round(tot_ms / max(launches, 1), 5),
        "algorithmic_tflop_per_job": round(tot_fl / 1e12, 3),
        "dominant_variant": {"name": dom["name"], "launches": dom["launches"], "avg_ms": round(dom["ms"] / dom["launches"], 5),
                             "tflops": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 2)},
        "traffic": pmc,
    }
    return block, kernels


def cpu_baseline(args, sd=None):
    """The restated reference path (fp32 CPU oracle = the CI configuration --use-cpu all --no-half) on this host, bounded
    sample: ONE CFG pair of UNet forwards (batch 2 = one image's cond + uncond) + ONE VAE decode, extrapolated to the
    workload (per image: sampler_steps x pair + decode).  Threads are capped: the oracle's small fp32 GEMMs stop scaling
    (and oversubscribe) far below the 128-256 hardware threads of the GPU box's host."""
    from oracle import pipeline as opipe, unet as ou, vae as ov
    schema = sub("schema")
    threads = args.cpu_baseline_threads or min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    if args.model == "tiny":
        ucfg, vcfg, ou_cfg, ov_cfg = schema.tiny_unet(), schema.tiny_vae(), ou.tiny_config(), ov.tiny_vae_config()
    else:
        ucfg, vcfg, ou_cfg, ov_cfg = schema.sd15_unet(), schema.sd15_vae(), ou.sd15_config(), ov.sd15_vae_config()
    if sd is None:
        sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16)
    sd = {k: v.cpu() for k, v in sd.items()}
    om = opipe.OracleModel(sd, ou_cfg, ov_cfg)
    hw = args.size // 8
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, hw, hw, generator=g)
    ctx = torch.randn(2, 77, ucfg.context_dim, generator=g)
    t = torch.tensor([500.0, 500.0])
    with torch.no_grad():
        t0 = time.time(); om.apply_model(x, t, ctx); t_pair = time.time() - t0
        t0 = time.time(); om.vae.decode_first_stage(x[:1]); t_dec = time.time() - t0
    per_image = args.sampler_steps * t_pair + t_dec
    return {"value": round(1.0 / per_image, 6), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 CFG pair of UNet forwards (batch 2) = {t_pair:.2f}s + 1 VAE decode = {t_dec:.2f}s at "
                      f"{args.size}x{args.size} (single cold run each), extrapolated to {args.sampler_steps} steps + decode per "
                      f"image; fp32 torch CPU, {threads} threads"}


def main():
    args = parse()
    par = sub("parallel")
    rank, local_rank, world = par.init_distributed()
    lib = sub("_lib")
    lib.require_device()
    torch.cuda.set_device(local_rank)
    schema, sd_models = sub("schema"), sub("sd_models")

    # ---- weights: synthetic checkpoint on rank 0, broadcast over RCCL (scatter + all-gather), packed per rank ----------
    if args.model == "tiny":
        ucfg, vcfg = schema.tiny_unet(), schema.tiny_vae()
    else:
        ucfg, vcfg = schema.sd15_unet(), schema.sd15_vae()
    t0 = time.time()
    sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16) if rank == 0 else None
    t_gen = time.time() - t0
    t_bcast = 0.0
    if world > 1:
        torch.cuda.synchronize(); par.barrier(); t0 = time.time()
        sd = par.broadcast_state_dict(sd, src=0, device=torch.device("cuda", local_rank))
        torch.cuda.synchronize(); t_bcast = time.time() - t0
    model = sd_models.SdModel(sd, ucfg, vcfg, device=local_rank, vae_decoder_only=True)
    keep_sd = sd if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    del sd
    run_once = make_job(args, model, rank, world)

    for _ in range(args.warmup):
        run_once()
    torch.cuda.synchronize(); par.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        run_once()
    torch.cuda.synchronize(); par.barrier()
    elapsed = par.max_over_ranks(time.time() - t0, device=torch.device("cuda", local_rank))

    roof, kernels = (None, None)
    if rank == 0 and not args.no_roofline:
        roof, kernels = roofline_block(args, run_once)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, keep_sd)
    par.barrier()
    if rank != 0:
        return
    images = args.batch * world * args.steps
    value = images / elapsed
    tflop_per_image = args.sampler_steps * 2 * UNET_TFLOP_PER_SAMPLE + VAE_TFLOP_PER_IMAGE
    out = {
        "metric": "images/sec SD1.5 512x512 20-step Euler-a, batch 8 per GPU",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{'SD1.5' if args.model == 'sd15' else 'tiny'} txt2img {args.size}x{args.size}, "
                               f"{args.sampler_steps}-step {args.sampler}, batch {args.batch} per GPU, cfg 7.0, fp16 weights/activations, "
                               f"fp32 accumulate + fp32 sampler state, Philox (NV) noise, VAE decode to uint8 included",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "weights": "synthetic N(0,1/fan_in) in the SD1.5 state-dict schema (seed 0x5D15)",
                   "weights_broadcast_ms": round(t_bcast * 1e3, 1), "weights_generate_s": round(t_gen, 1),
                   "algorithmic_tflop_per_image": round(tflop_per_image, 3),
                   "whole_job_mfma_frac": round(value / world * tflop_per_image / MFMA_PEAK_TFLOPS, 4)},
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    if kernels is not None:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w") as f:
            json.dump(kernels, f, indent=1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
