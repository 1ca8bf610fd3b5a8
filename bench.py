#!/usr/bin/env python3
"""bench.py — images/sec of the txt2img / img2img hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4a|c4b]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Launched WITHOUT a torchrun environment and with --gpus N > 1, bench.py spawns the N ranks itself (one process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would) and `n_gpus` in the line is the world size RCCL
reports — never the flag.

A "step" = one pass of the hot path over one batch of synthetic input = one whole job of the workload; with N ranks the job is the
global batch (batch per GPU x N images) sharded by `parallel.process_images_sharded` (contiguous split of the image index range,
global seeds 1000 + index, no per-step communication, one RCCL gather of the uint8 images to rank 0 per job).  Workloads
(BASELINE.json `configs`):
    c1   SD1.5 txt2img 512x512, 20-step Euler a, batch 8 per GPU                      <- the metric (default)
    c2   SD1.5 txt2img 512x512, 50-step DPM++ 2M Karras, batch 8 per GPU (64 over 8 GPUs)
    c3   SDXL-base txt2img 1024x1024, 30-step Euler a, batch 4 per GPU
    c4a  SD1.5 txt2img 512x512 + hires-fix x2 (Latent upscaler, denoise 0.75, 20 + 20 UNet evaluations), decode at 1024x1024, batch 8
    c4b  SD1.5 img2img 512x512 (VAE encode, denoise 0.75 -> 16 evaluations, decode), batch 8
Everything is resident in HBM when the timed region starts (synthetic fp16 weights in the checkpoint's state-dict schema, seeded
N(0,1) conditioning; the text encoder is outside the path).  Weights are generated on rank 0 and broadcast over RCCL.

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     the implicit-GEMM MFMA kernel family's achieved TFLOP/s vs the 2.5 PFLOP/s dense fp16 peak — per-launch HIP events
                  on the launch stream in a separate profiled pass of the same job; "traffic" = HBM bytes per launch of that family
                  from two rocprofv3 PMC passes (FETCH_SIZE, then WRITE_SIZE; kernel trace only) of one job of THIS workload, taken by
                  this run in child processes on the headline workload wherever rocprofv3 exists (--pmc-traffic / --no-pmc-traffic;
                  null when the passes were skipped or failed — a figure from another box is never reported)
  "cpu_baseline": the fp32 CPU oracle (restated reference path) timed on this host on a bounded sample.
`config.images_per_s_every_row_computed` re-times a few jobs with the CFG denoiser's common-subexpression option off (`cfg_pairs`: the
layers in front of the first cross-attention are the same function value for the cond and the uncond row of an image and are computed
once — every row's output is still produced; docs/DESIGN_experiments.md A.1), `config.dropin_images_per_s` the same job through the B1 / B4 boundaries.
"""
import argparse
import ctypes
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "stable-diffusion-webui_amd"
MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md

# SURVEY.md section 8(d) / BASELINE.md section 2: algorithmic TFLOP per image
CONFIGS = {
    "c1": dict(model="sd15", size=512, sampler="Euler a", sampler_steps=20, batch=8, tflop_per_image=34.646,
               metric="images/sec SD1.5 512x512 20-step Euler-a, batch 8 per GPU"),
    "c2": dict(model="sd15", size=512, sampler="DPM++ 2M", scheduler="karras", sampler_steps=50, batch=8, tflop_per_image=82.84,
               metric="images/sec SD1.5 512x512 50-step DPM++ 2M Karras, batch 8 per GPU (64 over 8 GPUs)"),
    "c3": dict(model="sdxl", size=1024, sampler="Euler a", sampler_steps=30, batch=4, tflop_per_image=416.1,
               metric="images/sec SDXL-base 1024x1024 30-step Euler-a, batch 4 per GPU"),
    "c4a": dict(model="sd15", size=512, sampler="Euler a", sampler_steps=20, batch=8, hires=True, tflop_per_image=229.56,
                metric="images/sec SD1.5 txt2img 512x512 + hires-fix x2 (Latent, denoise 0.75) -> 1024x1024, batch 8 per GPU"),
    "c4b": dict(model="sd15", size=512, sampler="Euler a", sampler_steps=20, batch=8, img2img=True, tflop_per_image=29.34,
                metric="images/sec SD1.5 img2img 512x512 denoise 0.75 (encode + 16 evaluations + decode), batch 8 per GPU"),
}


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--pmc-traffic", action="store_true", default=None,
                    help="take the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of one job of this workload before the roofline block, so "
                         "that roofline.traffic is measured in this run (two child processes of one job each, ~25 s per pass on the C1 job; "
                         "rank 0, N = 1).  Default: on for the headline workload (c1) when rocprofv3 is on PATH, off for the other configs")
    ap.add_argument("--no-pmc-traffic", dest="pmc_traffic", action="store_false", help="skip the PMC passes: roofline.traffic is null")
    ap.add_argument("--pmc-timeout", type=float, default=240.0, help="wall limit of one PMC pass (seconds); a pass that exceeds it leaves traffic null")
    ap.add_argument("--verify-shards", action="store_true",
                    help="N > 1: after the timed region, replay every rank's slice on rank 0 and compare with the gathered images (config.shard_check)")
    ap.add_argument("--steps", type=int, default=3, help="timed jobs (one job = the global batch through the whole path)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c1", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per job (default: the config's)")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--sampler-steps", type=int, default=0)
    ap.add_argument("--sampler", default="")
    ap.add_argument("--model", default="", choices=["", "sd15", "sdxl", "tiny"])
    ap.add_argument("--no-dropin", action="store_true", help="skip timing the webui drop-in path (config.dropin_images_per_s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-threads", type=int, default=0)
    ap.add_argument("--cpu-baseline-budget", type=float, default=185.0, help="seconds of CPU timing the baseline may spend (it adapts its repetitions)")
    ap.add_argument("--cpu-baseline-timeout", type=float, default=330.0, help="hard wall limit of the baseline child process")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) run the CPU baseline alone and print its JSON")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    args.model = args.model or cfg["model"]
    args.size = args.size or cfg["size"]
    args.sampler = args.sampler or cfg["sampler"]
    args.sampler_steps = args.sampler_steps or cfg["sampler_steps"]
    args.batch = args.batch or cfg["batch"]
    args.scheduler = cfg.get("scheduler")
    args.hires, args.img2img = bool(cfg.get("hires")), bool(cfg.get("img2img"))
    args.named = (args.model == cfg["model"] and args.size == cfg["size"] and args.sampler == cfg["sampler"]
                  and args.sampler_steps == cfg["sampler_steps"] and args.batch == cfg["batch"])
    args.cfg = cfg
    return args


# ----------------------------------------------------------------------------------------------------------------------------
# self-spawn: python bench.py --gpus N without torchrun
# ----------------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} requested but {have} GPU(s) visible; refusing to report n_gpus={n}", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    sys.exit(rc)


# ----------------------------------------------------------------------------------------------------------------------------
def model_configs(args):
    schema = sub("schema")
    if args.model == "tiny":
        return schema.tiny_unet(), schema.tiny_vae()
    if args.model == "sdxl":
        return schema.sdxl_unet(), schema.sdxl_vae()
    return schema.sd15_unet(), schema.sd15_vae()


def make_job(args, model, rank, world, local_only=False, replay=None):
    """The GLOBAL job (batch per GPU x world images) as one processing object; every rank builds the same one and
    process_images_sharded takes its slice."""
    import torch
    processing, par = sub("processing"), sub("parallel")
    ctx_dim = model.unet_cfg.context_dim
    n_total = args.batch * world
    g = [torch.Generator().manual_seed(50_000 + i) for i in range(n_total)]      # "synthetic prompt" i
    c = torch.stack([torch.randn(77, ctx_dim, generator=g[i]) for i in range(n_total)])
    uc = torch.stack([torch.randn(77, ctx_dim, generator=g[i]) for i in range(n_total)])
    y = uy = None
    if model.is_sdxl:
        adm = model.unet_cfg.adm_in_channels
        y = torch.stack([torch.randn(adm, generator=g[i]) for i in range(n_total)])
        uy = torch.stack([torch.randn(adm, generator=g[i]) for i in range(n_total)])
    lo, hi = par.shard_range(n_total, world, rank)
    dev = model.device
    c_dev, uc_dev = c.to(dev), uc.to(dev)                       # resident in HBM before the timed region
    if y is not None:
        y, uy = y.to(dev), uy.to(dev)
    kw = dict(sd_model=model, seed=1000, batch_size=args.batch, n_iter=world, steps=args.sampler_steps, cfg_scale=7.0,
              width=args.size, height=args.size, sampler_name=args.sampler, scheduler=args.scheduler, keep_latents=False)
    init = None
    if args.img2img:
        init = torch.stack([torch.rand(3, args.size, args.size, generator=g[i]) for i in range(n_total)]).to(dev)

    def run_once():
        if args.img2img:
            p = processing.StableDiffusionProcessingImg2Img(c=c_dev, uc=uc_dev, init_images=init, denoising_strength=0.75, **kw)
        elif args.hires:
            p = processing.StableDiffusionProcessingTxt2Img(c=c_dev, uc=uc_dev, enable_hr=True, hr_scale=2.0, hr_upscaler="Latent",
                                                            denoising_strength=0.75, **kw)
        else:
            p = processing.StableDiffusionProcessingTxt2Img(c=c_dev, uc=uc_dev, **kw)
        if y is not None:
            p.y, p.uy = y, uy
        if replay is not None:                                 # one rank's slice of the SAME global job, replayed in this process
            return par.process_images_sharded(p, world=world, rank=replay)
        return par.process_images_sharded(p, world=1, rank=0) if local_only else par.process_images_sharded(p)
    return run_once, (lo, hi)


def dropin_path(args, model, jobs=3, auto_promises=False):
    """What a webui user gets WITHOUT the engine's own process_images / sampler mirror: the reference's Python loop around the plugin
    boundaries.  A torch-side stand-in of that loop — per step `torch.cat` of cond | uncond and of x (modules/sd_samplers_cfg_denoiser.py:
    236-246), x * c_in, fp16 `Mi355xUnet.forward` through the SdUnet boundary B1 with the context validated on the device
    (modules/sd_unet.py:86-93), denoised = x - eps * sigma, the CFG combine and k-diffusion's sample_euler_ancestral update as
    torch elementwise ops on the fp32 state, then `first_stage_model.decode` as the B4 hook binds it (engine decode of z / scale_factor) and
    the clamp / x255 / uint8 conversion in torch (modules/processing.py:1004-1035).  Same UNet, same VAE kernels; what differs from `value`
    is everything between them.  Returns images/s over `jobs` jobs after one warm-up job."""
    import torch
    sd_unet, smp, rng_mod = sub("sd_unet"), sub("sd_samplers"), sub("rng")
    dev = model.device
    unet = sd_unet.Mi355xUnet(lambda: None, unet_cfg=model.unet_cfg, device_index=dev.index)
    unet.engine = model.engine                                 # the engine already holds this checkpoint: no second copy of the weights
    wrap = smp.CompVisDenoiser(model)
    sigmas = wrap.get_sigmas(args.sampler_steps).to(dev)
    B, hw = args.batch, args.size // 8
    g = torch.Generator().manual_seed(50_000)
    c = torch.randn(B, 77, model.unet_cfg.context_dim, generator=g).to(dev).half()
    uc = torch.randn(B, 77, model.unet_cfg.context_dim, generator=g).to(dev).half()
    cfg_scale = 7.0

    def job():
        rng = rng_mod.ImageRNG((4, hw, hw), [1000 + i for i in range(B)], device=dev)
        x = rng.next() * sigmas[0]
        for i in range(len(sigmas) - 1):
            sigma, sigma_next = sigmas[i], sigmas[i + 1]
            x_in = torch.cat([x, x])
            cond_in = torch.cat([c, uc])
            sigma_in = sigma.expand(2 * B)
            c_in = 1.0 / (sigma_in ** 2 + 1.0) ** 0.5
            t_in = wrap.sigma_to_t(sigma_in.cpu()).to(dev)
            eps = unet.forward((x_in * c_in[:, None, None, None]).half(), t_in.half(), cond_in)
            den_in = x_in - eps.float() * sigma_in[:, None, None, None]
            den_c, den_u = den_in[:B], den_in[B:]
            denoised = den_u + (den_c - den_u) * cfg_scale
            # sample_euler_ancestral (k-diffusion sampling.py): get_ancestral_step with eta = 1
            s_up = torch.minimum(sigma_next, (sigma_next ** 2 * (sigma ** 2 - sigma_next ** 2) / sigma ** 2) ** 0.5)
            s_down = (sigma_next ** 2 - s_up ** 2) ** 0.5
            d = (x - denoised) / sigma
            x = x + d * (s_down - sigma)
            if float(sigma_next) > 0:
                x = x + rng.next() * s_up
        img = model.engine.vae_decode(x / model.scale_factor * model.scale_factor)    # decode_first_stage: z / scale_factor inside the engine config
        u8 = (255.0 * torch.clamp((img + 1.0) / 2.0, 0.0, 1.0)).permute(0, 2, 3, 1).to(torch.uint8).cpu()
        return u8
    # auto_promises: Mi355xUnet.forward lets the engine derive the [x | x] / one-timestep facts from the data of each call
    # (opts.mi355x_auto_cfg_pairs: a synchronising device -> host compare per evaluation, then the shared CFG prefix)
    shared_mod = sub("shared")
    prev = getattr(shared_mod.opts, "mi355x_auto_cfg_pairs", True)
    shared_mod.opts.mi355x_auto_cfg_pairs = bool(auto_promises)
    try:
        job()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(jobs):
            job()
        torch.cuda.synchronize()
        return B * jobs / (time.time() - t0)
    finally:
        shared_mod.opts.mi355x_auto_cfg_pairs = prev


def pmc_traffic(args):
    """HBM bytes per launch of the gemm_mfma family from rocprofv3 PMC passes of this same workload ON THIS BOX: either taken by this very
    run (--pmc-traffic: measure_pmc_traffic below) or by tools/gpu/profile.sh earlier in the same box session (gpurun_out/ is scratch and
    never travels, so a file there was written here).  (None, None) otherwise — a figure from another box is not reported."""
    path = os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
    if os.path.exists(path):
        try:
            d = json.load(open(path))
            if d.get("workload") == f"{args.config}:{args.sampler_steps}":
                return d.get("gemm_mfma_bytes_per_launch"), d.get("source", "PMC passes of this workload on this box (gpurun_out/pmc_traffic.json)")
        except Exception:
            pass
    return None, None


def pmc_child_args(argv):
    """The command line of the one-job child a PMC pass profiles: this run's workload flags, without the flags that shape the timed
    region or start further children."""
    drop_with_value = ("--steps", "--warmup", "--gpus", "--cpu-baseline-budget", "--cpu-baseline-timeout", "--pmc-timeout")
    drop_flags = ("--pmc-traffic", "--no-pmc-traffic", "--verify-shards", "--no-cpu-baseline", "--no-roofline", "--no-dropin")
    child, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in drop_with_value:
            skip = True
            continue
        if a in drop_flags or any(a.startswith(k + "=") for k in drop_with_value):
            continue
        child.append(a)
    return child + ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--no-dropin", "--no-pmc-traffic"]


def measure_pmc_traffic(args):
    """--pmc-traffic: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE — they do not fit one pass; kernel trace only, as the guide's
    HBM section prescribes) over ONE job of this workload in child processes, summed over the gemm_mfma launches: bytes per launch =
    (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / launches (gfx950 tallies 128-byte read requests at 64 B: FETCH_SIZE is doubled).  Writes
    gpurun_out/pmc_traffic.json, which pmc_traffic() then reports.  A flag the driver's command does not need: without it `traffic` is
    null unless tools/gpu/profile.sh ran on this box."""
    import collections
    import glob
    import shutil
    import sqlite3
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    child = pmc_child_args(sys.argv[1:])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env["TMPDIR"] = "/tmp"
    sums = {}
    for counter, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        d = os.path.join(out_dir, f"prof_pmc_{tag}")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)] + child
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
        try:
            _, err = proc.communicate(timeout=args.pmc_timeout)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, 9)                   # the profiler's own session: rocprofv3 and the python child it started
            except Exception:
                proc.kill()
            proc.wait()
            print(f"bench.py: PMC pass {counter} stopped at the {args.pmc_timeout:.0f} s wall limit", file=sys.stderr)
            return None
        dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
        if proc.returncode != 0 or not dbs:
            print(f"bench.py: PMC pass {counter} failed (rc {proc.returncode}): {err.decode(errors='replace')[-400:]}", file=sys.stderr)
            return None
        acc = collections.defaultdict(lambda: [0, 0.0])
        con = sqlite3.connect(dbs[0])
        for name, val in con.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
            a = acc[name]
            a[0] += 1; a[1] += val
        sums[tag] = acc
    calls = kb = 0.0
    for name, (c, fv) in sums["fetch"].items():
        if any(pfx in name for pfx in FAMILY_PREFIXES):
            wc, wv = sums["write"].get(name, [1, 0.0])
            calls += c
            kb += 2.0 * fv + wv * c / max(wc, 1)
    if not calls:
        return None
    res = {"workload": f"{args.config}:{args.sampler_steps}", "gemm_mfma_bytes_per_launch": round(kb * 1024 / calls), "gemm_mfma_launches": int(calls),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over one job of this workload, taken by this bench.py run (--pmc-traffic)",
           "note": "2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, per launch over the gemm_mfma family"}
    json.dump(res, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
    return res


FAMILY_PREFIXES = ("gemm_mfma", "splitk", "rowchain_")


def roofline_block(args, run_once):
    """Profiled pass (separate from the timed region): HIP events around every launch on its stream."""
    import torch
    lib = sub("_lib")
    lib.check(lib.lib.sdmi_profile_begin(), "profile_begin")
    run_once()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 21)
    lib.check(lib.lib.sdmi_profile_end(buf, len(buf)), "profile_end")
    kernels = json.loads(buf.value.decode())["kernels"]
    # the MFMA GEMM family: the implicit-GEMM kernels, the split-K reduce passes their deep-K launches need (time only: no flops of their
    # own), and the fused feed-forward / cross-attention chains of the 320-wide level (csrc/rowchain.hip: the same GEMMs in one launch)
    fam = [k for k in kernels if k["name"].startswith(FAMILY_PREFIXES)]
    if not fam:
        return None, kernels
    tot_ms = sum(k["ms"] for k in fam)
    tot_fl = sum(k["flops"] for k in fam)
    launches = sum(k["launches"] for k in fam)
    all_ms = sum(k["ms"] for k in kernels)
    dom = max(fam, key=lambda k: k["ms"])
    achieved = tot_fl / (tot_ms * 1e-3) / 1e12
    # the job's largest kernels whatever their family (launch names carry the shape: grouped by the kernel form in front of it), so that
    # attention — outside the GEMM family — shows up where it belongs
    groups = {}
    for k in kernels:
        g = groups.setdefault(k["name"].split(" ")[0], {"ms": 0.0, "flops": 0.0, "launches": 0})
        g["ms"] += k["ms"]; g["flops"] += k.get("flops", 0.0); g["launches"] += k["launches"]
    top = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:3]
    top_kernels = [{"name": n, "launches": g["launches"], "ms_per_job": round(g["ms"], 2), "share_of_kernel_time": round(g["ms"] / all_ms, 4),
                    "tflops": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 1) if g["flops"] else None,
                    "mfma_frac": round(g["flops"] / (g["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if g["flops"] else None} for n, g in top]
    block = {
        "bound": "mfma", "kernel": "gemm_mfma_kernel (implicit-GEMM conv3x3 / 1x1 / linear, all tile configs) + splitk_reduce passes + rowchain_ff (fused feed-forward)",
        "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
        "launches_per_job": launches, "avg_launch_ms": round(tot_ms / max(launches, 1), 5),
        "algorithmic_tflop_per_job": round(tot_fl / 1e12, 3),
        "family_ms_per_job": round(tot_ms, 2), "all_kernels_ms_per_job": round(all_ms, 2),
        "profiled_pass_note": "family / all_kernels figures come from a SEPARATE pass of the job with HIP events around every launch (slower than the timed region: ms_per_step is the timed one)",
        "top_kernels_of_the_job": top_kernels,
        "dominant_variant": {"name": dom["name"], "launches": dom["launches"], "avg_ms": round(dom["ms"] / dom["launches"], 5),
                             "tflops": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 2)},
        "traffic": pmc_traffic(args)[0], "traffic_source": pmc_traffic(args)[1],
    }
    # the same family's ALGORITHMIC bytes per launch (activations read once + weights + output written once: launch_gemm's ProfScope
    # figure), so that the measured traffic reads as a ratio: well above 1 = re-reads
    alg_bytes = sum(k.get("bytes", 0) for k in fam)
    if alg_bytes and launches:
        block["traffic_algorithmic"] = round(alg_bytes / launches)
        if block["traffic"]:
            block["traffic_over_algorithmic"] = round(block["traffic"] / (alg_bytes / launches), 3)
    return block, kernels


def usable_cpus():
    """Hardware threads this process may actually use: the scheduler affinity mask, capped by a cgroup CPU quota if one is set
    (os.cpu_count() reports the host's threads even inside a quota-limited container: oversubscribing them is pathological)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(args, sd=None):
    """the restated reference path (fp32 CPU oracle = the CI configuration --use-cpu all --no-half) on this
    host, 1 warm-up + 3 timed runs of each measured piece while the time budget lasts (--cpu-baseline-budget; a piece whose first run
    is slow is measured by that run alone).  C0 (256x256, 5-step Euler a, batch 1) is timed IN FULL when the budget allows;
    the workload itself is a bounded sample — ONE CFG pair of UNet forwards (batch 2 = one image's cond + uncond) + ONE VAE decode
    at the workload's size — extrapolated to evaluations x pair + decode per image."""
    t_enter = time.time()
    import torch
    from oracle import pipeline as opipe, unet as ou, vae as ov
    schema = sub("schema")
    host = usable_cpus()
    threads = args.cpu_baseline_threads or host
    torch.set_num_threads(threads)
    if args.model == "tiny":
        ucfg, vcfg, ou_cfg, ov_cfg = schema.tiny_unet(), schema.tiny_vae(), ou.tiny_config(), ov.tiny_vae_config()
    elif args.model == "sdxl":
        ucfg, vcfg, ou_cfg, ov_cfg = schema.sdxl_unet(), schema.sdxl_vae(), ou.sdxl_base_config(), ov.VAEConfig(scale_factor=0.13025)
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
    yv = torch.randn(2, ucfg.adm_in_channels, generator=g) if ucfg.adm_in_channels else None
    t = torch.tensor([500.0, 500.0])
    deadline = time.time() + args.cpu_baseline_budget
    notes = []

    def timed(fn, runs=3):
        """median of `runs` after one warm-up — or, when the warm-up run already shows that repetitions would not fit the budget,
        that single (cold) run"""
        t0 = time.time(); fn(); first = time.time() - t0
        if time.time() + runs * first > deadline:
            notes.append("single cold run")
            return first, 1
        ts = []
        for _ in range(runs):
            t0 = time.time(); fn(); ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2], runs
    with torch.no_grad():
        t_pair, n_pair = timed(lambda: om.apply_model(x, t, ctx, yv))
        t_dec, n_dec = timed(lambda: om.vae.decode_first_stage(x[:1]))
        if threads > 32 and not args.cpu_baseline_threads and time.time() + 2.5 * t_pair < deadline:
            # the oracle's fp32 GEMMs stop scaling well below 128+ hardware threads: keep whichever thread count is faster
            torch.set_num_threads(32)
            t0 = time.time(); om.apply_model(x, t, ctx, yv); om.apply_model(x, t, ctx, yv); t_pair32 = (time.time() - t0) / 2
            if t_pair32 < t_pair:
                t_pair, threads = t_pair32, 32
                t0 = time.time(); om.vae.decode_first_stage(x[:1]); t_dec = min(t_dec, time.time() - t0)
            else:
                torch.set_num_threads(threads)
        out = {}
        if args.model == "sd15":                               # C0 in full: 256x256, 5-step Euler a, batch 1, decode included
            # ~10 UNet evaluations at a quarter of the pixels + one small decode: ~0.35 x (10 pairs) of the 512x512 cost
            if time.time() + 4 * t_pair < deadline:
                c0c, c0u = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)

                def c0():
                    lat = opipe.sample(om, c0c, c0u, [1000], 5, "euler_a", 7.0, (32, 32))
                    opipe.to_uint8_hwc(opipe.decode(om, lat))
                t_c0, n_c0 = timed(c0, runs=3)
                out["c0_full"] = {"seconds": round(t_c0, 3), "images_per_s": round(1.0 / t_c0, 5),
                                  "what": "BASELINE.json configs[0]: SD1.5 256x256 5-step Euler a batch 1, noise -> uint8, "
                                          + ("median of 3 after 1 warm-up" if n_c0 > 1 else "one cold run (time budget)")}
            else:
                out["c0_full"] = None
                notes.append("C0 skipped (time budget)")
    evals = args.sampler_steps
    if args.img2img:
        evals = 16
    per_image = evals * t_pair + t_dec
    # BASELINE.md section 3's own protocol when the budget has room for it: the CFG step at the workload's batch (one UNet forward of
    # 2 x batch rows), measured twice, instead of `batch` single-image pairs — large-batch GEMMs use the host's cores better.
    step_note = ""
    nb = args.batch
    if nb > 1 and args.model != "tiny" and time.time() + 2.2 * nb * t_pair < deadline and (time.time() - t_enter) + 2.4 * nb * t_pair < 0.85 * args.cpu_baseline_timeout:
        gb = torch.Generator().manual_seed(2)
        xb = torch.randn(2 * nb, 4, hw, hw, generator=gb)
        cb = torch.randn(2 * nb, 77, ucfg.context_dim, generator=gb)
        yb = torch.randn(2 * nb, ucfg.adm_in_channels, generator=gb) if ucfg.adm_in_channels else None
        tb = torch.full((2 * nb,), 500.0)
        with torch.no_grad():
            ts = []
            for _ in range(2):
                t0 = time.time(); om.apply_model(xb, tb, cb, yb); ts.append(time.time() - t0)
        t_step = min(ts)
        out["cfg_step_at_batch"] = {"seconds": round(t_step, 3), "rows": 2 * nb, "runs": 2,
                                    "per_image_pair_equivalent_s": round(t_step / nb, 3)}
        if t_step / nb < t_pair:
            per_image = evals * (t_step / nb) + t_dec
            step_note = f"; UNet cost taken from 2 measured CFG steps at batch {nb} ({2 * nb} rows, {t_step:.2f}s best) = {t_step / nb:.2f}s per image-evaluation"
    if args.hires:
        per_image = None
    out.update({"value": round(1.0 / per_image, 6) if per_image else None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": f"1 CFG pair of UNet forwards (batch 2) = {t_pair:.2f}s + 1 VAE decode = {t_dec:.2f}s at "
                          f"{args.size}x{args.size} ({'median of 3 after 1 warm-up' if n_pair > 1 else 'one cold run'}), extrapolated to "
                          f"{evals} evaluations + decode per image; fp32 torch CPU, {threads} threads of {host} usable "
                          f"({os.cpu_count()} host) hardware threads" + step_note + ("; " + "; ".join(sorted(set(notes))) if notes else "")})
    return out


def cpu_baseline_isolated(args):
    """The baseline in a child process under a hard wall limit, so that a pathological host (oversubscribed cores, a throttled
    container) can delay the bench line by at most --cpu-baseline-timeout seconds and never lose it."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"] + [a for a in sys.argv[1:] if a != "--cpu-baseline-only"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=args.cpu_baseline_timeout)
        for line in reversed(stdout.decode(errors="replace").splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": f"baseline child exited with code {proc.returncode} and no result"}
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, 9)                       # the child's own session / process group only
        except Exception:
            proc.kill()
        proc.wait()
        return {"value": None, "unit": "images/s", "cores": usable_cpus(), "kind": "port",
                "sample": f"CPU baseline stopped at the {args.cpu_baseline_timeout:.0f} s wall limit ({time.time() - t0:.0f} s) on this host"}


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    import torch
    par = sub("parallel")
    rank, local_rank, world = par.init_distributed()
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == world
    lib = sub("_lib")
    lib.require_device()
    torch.cuda.set_device(local_rank)
    schema, sd_models = sub("schema"), sub("sd_models")

    # ---- weights: synthetic checkpoint on rank 0, broadcast over RCCL (scatter + all-gather), packed per rank ----------
    ucfg, vcfg = model_configs(args)
    t0 = time.time()
    sd = schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16) if rank == 0 else None
    t_gen = time.time() - t0
    t_bcast = 0.0
    if world > 1:
        torch.cuda.synchronize(); par.barrier(); t0 = time.time()
        import torch.distributed as dist
        sd = par.broadcast_state_dict(sd, src=0, device=torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else "cpu")
        torch.cuda.synchronize(); t_bcast = time.time() - t0
    model = sd_models.SdModel(sd, ucfg, vcfg, device=local_rank, vae_decoder_only=not (args.img2img))
    del sd
    run_once, shard = make_job(args, model, rank, world)

    for _ in range(args.warmup):
        run_once()
    torch.cuda.synchronize(); par.barrier()
    coll0 = par.COLLECTIVES["job"]
    t0 = time.time()
    for _ in range(args.steps):
        run_once()
    torch.cuda.synchronize(); par.barrier()
    elapsed = par.max_over_ranks(time.time() - t0, device=torch.device("cuda", local_rank))
    collectives_per_job = (par.COLLECTIVES["job"] - coll0) / max(args.steps, 1)      # data-path collectives inside the timed region

    shard_check = None
    if args.verify_shards and world > 1:
        # the gathered images of one more sharded job against rank-local replays of every rank's slice on rank 0's GPU: the images are
        # functions of (weights, cond row, seed + global index) only, so they must be bit-identical (same per-call batch size)
        whole = run_once()
        if rank == 0:
            import numpy as np
            bad = []
            for r in range(world):
                res_r = make_job(args, model, 0, world, replay=r)[0]()
                lo_r, hi_r = res_r.shard
                if len(res_r.images) != hi_r - lo_r or not all(np.array_equal(a, b) for a, b in zip(res_r.images, whole.images[lo_r:hi_r])):
                    bad.append(r)
            shard_check = "ok" if not bad and len(whole.images) == args.batch * world else f"MISMATCH on the slices of ranks {bad}"
        par.barrier()
    if args.pmc_traffic is None:                           # default: measured for the headline workload wherever the profiler exists
        import shutil
        # (not when this process is itself being profiled: the children would inherit the outer tool's environment)
        profiled = any(k in os.environ for k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")) or \
            any(k.startswith(("ROCPROF_", "ROCPROFV3_")) for k in os.environ)
        args.pmc_traffic = args.config == "c1" and args.named and not profiled and shutil.which("rocprofv3") is not None
    if args.pmc_traffic and rank == 0 and world == 1 and not args.no_roofline:
        try:
            measure_pmc_traffic(args)
        except Exception as ex:                                # never lose the bench line to the profiler
            print(f"bench.py: --pmc-traffic failed: {type(ex).__name__}: {ex}", file=sys.stderr)
    roof, kernels = (None, None)
    if rank == 0 and not args.no_roofline:
        if world > 1:                                  # profile this rank's slice alone (no collective inside the profiled pass)
            run_local, _ = make_job(args, model, 0, 1, local_only=True)
            roof, kernels = roofline_block(args, run_local)
        else:
            roof, kernels = roofline_block(args, run_once)
    dropin = dropin_auto = None
    if rank == 0 and world == 1 and not args.no_dropin and not (args.hires or args.img2img) and args.sampler == "Euler a":
        try:
            dropin = round(dropin_path(args, model, jobs=min(3, max(1, args.steps))), 4)
        except Exception as ex:                                # the drop-in stand-in must never cost the bench line
            dropin = f"failed: {type(ex).__name__}: {ex}"
        try:
            dropin_auto = round(dropin_path(args, model, jobs=min(3, max(1, args.steps)), auto_promises=True), 4)
        except Exception as ex:
            dropin_auto = f"failed: {type(ex).__name__}: {ex}"
    per_row = None
    if rank == 0 and world == 1 and not args.no_dropin:
        # transparency leg: the same job with the CFG denoiser's common-subexpression options off — every UNet row computed on its own, as the
        # reference does (engine.CFG_PAIRS: the layers in front of the first cross-attention run for both halves of the [cond | uncond] batch)
        eng_mod = importlib.import_module(PKG + ".engine")
        prev_pairs = eng_mod.CFG_PAIRS
        try:
            eng_mod.CFG_PAIRS = False
            run_once()
            torch.cuda.synchronize(); t1 = time.time()
            n_jobs = min(3, max(1, args.steps))
            for _ in range(n_jobs):
                run_once()
            torch.cuda.synchronize()
            per_row = round(args.batch * n_jobs / (time.time() - t1), 4)
        except Exception as ex:
            per_row = f"failed: {type(ex).__name__}: {ex}"
        finally:
            eng_mod.CFG_PAIRS = prev_pairs
    acc_ips = None
    if rank == 0 and world == 1 and not args.no_dropin:
        # the accuracy mode (engine option "residual_fp32": every tensor that is not a matrix-core operand with ~22 bits — the
        # configuration that meets north_star's <= 1e-3 per forward) on the same job
        shared_mod = importlib.import_module(PKG + ".shared")
        try:
            shared_mod.opts.sdmi_accuracy_mode = True          # process_images switches the engine option from it on every job
            run_once()
            torch.cuda.synchronize(); t1 = time.time()
            n_jobs = min(3, max(1, args.steps))
            for _ in range(n_jobs):
                run_once()
            torch.cuda.synchronize()
            acc_ips = round(args.batch * n_jobs / (time.time() - t1), 4)
        except Exception as ex:
            acc_ips = f"failed: {type(ex).__name__}: {ex}"
        finally:
            shared_mod.opts.sdmi_accuracy_mode = False
            model.set_accuracy_mode(False)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_isolated(args)
    par.barrier()
    if rank != 0:
        return
    images = args.batch * world * args.steps
    value = images / elapsed
    tflop_per_image = args.cfg["tflop_per_image"] if args.named else None
    names = {"sd15": "SD1.5", "sdxl": "SDXL-base", "tiny": "tiny"}
    kind = "img2img (denoise 0.75)" if args.img2img else ("txt2img + hires-fix x2 (Latent, denoise 0.75)" if args.hires else "txt2img")
    out = {
        "metric": args.cfg["metric"] if args.named else f"images/sec {names[args.model]} {args.size}x{args.size} {args.sampler_steps}-step {args.sampler}, batch {args.batch} per GPU",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{args.config}: {names[args.model]} {kind} {args.size}x{args.size}, "
                               f"{args.sampler_steps}-step {args.sampler}{' Karras' if args.scheduler == 'karras' else ''}, batch {args.batch} per GPU, cfg 7.0, "
                               f"fp16 weights/activations, fp32 accumulate + fp32 sampler state, Philox (NV) noise, VAE decode to uint8 included",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "sharding": "process_images_sharded: contiguous image ranges, seeds 1000 + global index, one uint8 gather to rank 0 per job straight from the device buffer",
                   "weights": "synthetic N(0,1/fan_in) in the checkpoint's state-dict schema (seed 0x5D15)",
                   "weights_broadcast_ms": round(t_bcast * 1e3, 1), "weights_generate_s": round(t_gen, 1),
                   "weights_collectives": par.COLLECTIVES["weights"], "collectives_per_job": collectives_per_job,
                   "shard_check": shard_check,
                   "algorithmic_tflop_per_image": tflop_per_image,
                   "cfg_pairs": "on (samplers' default): both halves of the CFG batch share latent and timestep, so conv_in, the first ResBlock and "
                                "GroupNorm / proj_in / norm1 / self-attention of the first transformer block are computed once per image and copied; every "
                                "row's output is produced (docs/DESIGN_experiments.md A.1)",
                   "images_per_s_every_row_computed": per_row,
                   "accuracy_mode_images_per_s": acc_ips,
                   "accuracy_mode": "engine option residual_fp32 (--no-half / opts.sdmi_accuracy_mode): <= 1e-3 per UNet forward from the fp32 oracle "
                                    "(tests/test_gpu_c1_parity.py); NOT the configuration of `value`",
                   "dropin_images_per_s": dropin,
                   "dropin_auto_promises_images_per_s": dropin_auto,
                   "dropin_path": "torch stand-in of the reference's CFGDenoiser + Euler-a loop calling Mi355xUnet.forward per step + engine VAE decode (bench.py dropin_path); "
                                  "dropin_images_per_s: every row computed; dropin_auto_promises_images_per_s: the engine derives the [x | x] / one-timestep facts per call "
                                  "(opts.mi355x_auto_cfg_pairs, the extension's default since round 6)",
                   # whole job against the MFMA ceiling, two ways: on the REFERENCE graph's flops (what an image costs the reference: the
                   # yardstick that stays comparable across rounds), and on the flops the engine actually executes — with cfg_pairs the
                   # shared prefix runs once per image pair, so the like-for-like rate is the every-row-computed one
                   "whole_job_mfma_frac": round(value / world * tflop_per_image / MFMA_PEAK_TFLOPS, 4) if tflop_per_image else None,
                   "whole_job_mfma_frac_on_executed_flops": (round(per_row / world * tflop_per_image / MFMA_PEAK_TFLOPS, 4)
                                                            if (tflop_per_image and isinstance(per_row, (int, float))) else None)},
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    if kernels is not None:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_{args.config if args.named else 'custom'}.json"), "w") as f:
            json.dump(kernels, f, indent=1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
