#!/usr/bin/env python3
"""Board power and shader clock of the GPU while the hot path runs — the measurement behind DESIGN.md 9.2's closing sentence ("the path from
here to 0.5 is not a schedule or a tile but power per flop").  A sampler thread reads the amdgpu hwmon files (power1_average / power1_input,
freq1_input; `rocm-smi --json` when they are absent) every few milliseconds while, in the same process and one after the other:

    idle | the C1 job x 3 | level-0 self-attention loop (random / zero operands) | 3x3 conv loop at level 1 (random / zero operands)
         | 1x1 projection loop at level 0 (fabric-bound) | LayerNorm loop (bandwidth class)

Each loop runs ~1.5 s so that the power controller settles.  Per phase: mean / max power, mean / min clock, achieved TFLOP/s or TB/s.
    python tools/gpu/power_trace.py [--out gpurun_out/power_trace.json]        (does not import oracle/)"""
import argparse
import glob
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = "stable-diffusion-webui_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


class Sampler(threading.Thread):
    def __init__(self, period=0.005, smi=False):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_flag = period, [], False
        self.power_file = self.freq_file = None
        self.raw_first = None
        for hw in ([] if smi else sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))):
            for name in ("power1_average", "power1_input"):
                f = os.path.join(hw, name)
                if self.power_file is None and os.path.exists(f) and self._read(f) is not None:
                    self.power_file = f
            f = os.path.join(hw, "freq1_input")
            if self.freq_file is None and os.path.exists(f) and self._read(f) is not None:
                self.freq_file = f
            if self.power_file:
                break
        self.source = "hwmon" if self.power_file else "rocm-smi"
        if not self.power_file:
            self.period = max(period, 0.2)

    @staticmethod
    def _read(f):
        try:
            return int(open(f).read().strip())
        except (OSError, ValueError):
            return None

    def _smi(self):
        try:
            js = json.loads(subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"], capture_output=True, timeout=5).stdout)
            if self.raw_first is None:
                self.raw_first = js
            card = js[sorted(k for k in js if k.startswith("card"))[0]]
            watts = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
            mhz = next((float("".join(ch for ch in str(v) if ch.isdigit() or ch == ".")) for k, v in card.items() if k.startswith("sclk clock speed")), None)
            return watts, mhz
        except Exception:
            return None, None

    def run(self):
        while not self.stop_flag:
            t = time.time()
            if self.power_file:
                pw = self._read(self.power_file)
                fr = self._read(self.freq_file) if self.freq_file else None
                self.samples.append((t, pw / 1e6 if pw is not None else None, fr / 1e6 if fr is not None else None))
            else:
                w, mhz = self._smi()
                self.samples.append((t, w, mhz))
            time.sleep(self.period)

    def window(self, t0, t1):
        rows = [s for s in self.samples if t0 <= s[0] <= t1]
        pw = [s[1] for s in rows if s[1] is not None]
        fr = [s[2] for s in rows if s[2] is not None]
        return {"samples": len(rows), "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None,
                "sclk_mhz_mean": round(sum(fr) / len(fr), 0) if fr else None, "sclk_mhz_min": round(min(fr), 0) if fr else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "power_trace.json"))
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--smi", action="store_true", help="sample through rocm-smi (the SMU's gpu_metrics) instead of the hwmon files, which on this pool "
                    "report a slow-moving power and a clock that is not the shader clock")
    args = ap.parse_args()
    sub("_lib").require_device()
    ops, schema, sd_models, processing = sub("ops"), sub("schema"), sub("sd_models"), sub("processing")
    smp = Sampler(smi=args.smi)
    smp.start()
    out = {"source": smp.source, "power_file": smp.power_file, "freq_file": smp.freq_file, "phases": []}

    def phase(name, fn, work=None, unit=None, min_seconds=None):
        """fn() enqueues one unit of the phase; work = flops or bytes per fn()."""
        secs = args.seconds if min_seconds is None else min_seconds
        if fn is not None:
            fn(); torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        while time.time() - t0 < secs:
            if fn is None:
                time.sleep(0.05)
            else:
                for _ in range(8):
                    fn()
                torch.cuda.synchronize()
                n += 8
        t1 = time.time()
        # the first third of a phase is the controller settling: report the rest
        row = dict(name=name, launches=n, seconds=round(t1 - t0, 3), **smp.window(t0 + (t1 - t0) / 3, t1))
        if work and n:
            row["rate"] = round(work * n / (t1 - t0) / 1e12, 1)
            row["rate_unit"] = unit
        out["phases"].append(row)
        print(json.dumps(row), flush=True)

    phase("idle", None, min_seconds=1.0)

    # ---- the C1 job ---------------------------------------------------------------------------------------------------------
    ucfg, vcfg = schema.sd15_unet(), schema.sd15_vae()
    model = sd_models.SdModel(schema.synthetic_state_dict(ucfg, vcfg, dtype=torch.float16), ucfg, vcfg, device=0, vae_decoder_only=True)
    g = torch.Generator().manual_seed(50_000)
    c, uc = torch.randn(8, 77, 768, generator=g).cuda(), torch.randn(8, 77, 768, generator=g).cuda()

    def job():
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=model, c=c, uc=uc, seed=1000, batch_size=8, n_iter=1, steps=20, cfg_scale=7.0,
                                                        width=512, height=512, sampler_name="Euler a", keep_latents=False)
        return processing.process_images(p)
    job(); torch.cuda.synchronize()
    t0 = time.time()
    njobs = max(3, int(args.seconds * 2))
    for _ in range(njobs):
        job()
    torch.cuda.synchronize()
    t1 = time.time()
    row = dict(name=f"c1 job x {njobs} (SD1.5 512x512, 20-step Euler a, batch 8)", seconds=round(t1 - t0, 3), images_per_s=round(8 * njobs / (t1 - t0), 2),
               **smp.window(t0 + 0.3, t1))
    out["phases"].append(row)
    print(json.dumps(row), flush=True)
    del model
    torch.cuda.empty_cache()
    phase("idle after the jobs", None, min_seconds=1.0)

    # ---- isolated loops -----------------------------------------------------------------------------------------------------
    b, h, n, d = 16, 8, 4096, 40
    for fill in ("random", "zero"):
        mk = (lambda *s: torch.randn(*s, generator=g).half().cuda()) if fill == "random" else (lambda *s: torch.zeros(*s, dtype=torch.float16, device="cuda"))
        q, k, vt = mk(b, n, h * d), mk(b, n, h * d), mk(b, h * d, n)
        phase(f"level-0 self-attention loop (B16 H8 N4096 d40), {fill} operands", lambda: ops.attention_vt(q, k, vt, h, n), 4.0 * b * h * n * n * d, "TFLOP/s")
    for fill in ("random", "zero"):
        mk = (lambda *s: (torch.randn(*s, generator=g) * 0.5).half().cuda()) if fill == "random" else (lambda *s: torch.zeros(*s, dtype=torch.float16, device="cuda"))
        a = mk(16, 32, 32, 640)
        w = ops.pack_conv_weight(mk(640, 640, 3, 3) * 0.05)
        phase(f"3x3 conv loop, level 1 (M16384 N640 K5760), {fill} operands", lambda: ops.conv_gemm(a, w), 2.0 * 16384 * 640 * 5760, "TFLOP/s")
    a = (torch.randn(16, 64, 64, 320, generator=g) * 0.5).half().cuda()
    w = ops.pack_conv_weight((torch.randn(320, 320, generator=g) * 0.05).half().cuda())
    r = torch.randn(16, 64, 64, 320, generator=g).half().cuda()
    phase("1x1 projection loop, level 0 (M65536 N320 K320 + residual)", lambda: ops.conv_gemm(a, w, resid=r, taps=1, pad=0), 3.0 * 65536 * 320 * 2, "TB/s (activation in + residual in + out)")
    x = torch.randn(65536, 320, generator=g).half().cuda()
    gm, bt = torch.ones(320).cuda(), torch.zeros(320).cuda()
    phase("LayerNorm loop (65536 rows x 320)", lambda: ops.layernorm(x, gm, bt), 2.0 * 65536 * 320 * 2, "TB/s (in + out)")
    phase("idle at the end", None, min_seconds=1.0)

    smp.stop_flag = True
    out["raw_first_sample"] = smp.raw_first
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    txt = os.path.splitext(args.out)[0] + ".txt"
    with open(txt, "w") as f:
        f.write(f"source: {out['source']}  power: {out['power_file']}  clock: {out['freq_file']}\n")
        f.write(f"{'phase':86s} {'W mean':>8s} {'W max':>8s} {'MHz mean':>9s} {'MHz min':>8s}  rate\n")
        for ph in out["phases"]:
            rate = f"{ph['rate']} {ph['rate_unit']}" if "rate" in ph else (f"{ph['images_per_s']} images/s" if "images_per_s" in ph else "")
            f.write(f"{ph['name'][:86]:86s} {str(ph['power_w_mean']):>8s} {str(ph['power_w_max']):>8s} {str(ph['sclk_mhz_mean']):>9s} {str(ph['sclk_mhz_min']):>8s}  {rate}\n")
    print(open(txt).read())


if __name__ == "__main__":
    main()
