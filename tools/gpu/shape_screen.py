#!/usr/bin/env python3
"""Torch-free pre-screen of the GEMM tile configuration / split-K factor per exact launch shape, inside the UNet forward (or the VAE
decode) instead of inside the whole job — shape_tune.py's search at a twentieth of its price per candidate (a forward is 18 ms, a job 400,
and nothing here imports torch).

    python tools/gpu/shape_screen.py [--what unet|vae] [--top 30] [--passes 2] [--min-gain 0.03] [--model sd15] [--rows 16] [--hw 64]

Round k applies the k-th candidate of EVERY target shape together through sdmi_debug_set_str("gemm_override", ...) and reads each shape's
time from the per-launch HIP-event profile of one forward (`--passes` profiled forwards per round, minimum taken), i.e. with the operands in
the cache state the engine leaves them in.  Output: per shape the default's time, the best candidate and the predicted gain; then ONE
verification A/B of the forward with all winners applied together (interleaved, fwd_ab's timing loop).  What survives goes to
tools/gpu/shape_tune.py --emit or straight into csrc/gemm_tuned_shapes.inc after a whole-job A/B.  (Does not import oracle/.)
"""
import json
import os
import re
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fwd_ab  # noqa: E402

# tile configurations a shape may be forced to (gemm.hip GemmCfg; 10-13: the ring-buffered 4-wave forms — candidates again since round 5
# gave them a bare barrier: round 4 had measured them with a drained ring)
CFG_BN = {0: 128, 2: 64, 3: 128, 4: 256, 5: 320, 6: 128, 7: 64, 8: 320, 9: 160, 10: 160, 11: 128, 12: 64, 13: 160}
CFG_NAME = {0: "128x128", 2: "64x64", 3: "128x128k32", 4: "256x256", 5: "256x320", 6: "256x128", 7: "128x64", 8: "128x320", 9: "128x160",
            10: "128x160r4", 11: "128x128r4", 12: "128x64r3", 13: "128x160r3"}


def parse(name):
    """kernel profile name -> (M, N, K, taps, kind) or None (batched launches and non-GEMM kernels are skipped)."""
    if not name.startswith("gemm_mfma") or re.search(r" x\d+$", name):
        return None
    m = re.search(r" M(\d+) N(\d+) K(\d+)", name)
    if not m:
        return None
    head = name.split(" ")[0]
    kind = 1 if "_geglu" in head else 2 if "_tr" in head else 0
    return (int(m.group(1)), int(m.group(2)), int(m.group(3)), 9 if "conv3x3" in head else 1, kind + (4 if "_hl" in head else 0))   # + 4: (hi, lo) launch


def candidates(key):
    M, N, K, taps, kind = key
    out = []
    for cfg, bn in CFG_BN.items():
        if N % bn or (kind % 4 == 1 and cfg in (5, 6, 8, 9, 10, 13)):
            continue
        splits = [1]
        if kind % 4 == 0 and K >= 64 * 16 and ((M + 127) // 128) * ((N + 127) // 128) < 512:      # a split-K workspace exists for these
            splits += [s for s in (2, 3, 4, 6, 8) if K // 64 // s >= 4]
        out += [(cfg, s) for s in splits]
    return out


def shape_times(kernels):
    """{shape key: (us per launch, launches)} of one profiled forward; a shape launched under several kernel names (stats / plain epilogue)
    is summed."""
    acc = {}
    for k in kernels:
        key = parse(k["name"])
        if key is None:
            continue
        a = acc.setdefault(key, [0.0, 0])
        a[0] += k["ms"] * 1e3
        a[1] += k["launches"]
    return {key: (us / n, n) for key, (us, n) in acc.items() if n}


def main():
    ap = fwd_ab.parser()
    ap.add_argument("--top", type=int, default=30, help="shapes screened, by their share of the forward")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--min-gain", type=float, default=0.03, help="a winner must beat the default by this fraction of the shape's time")
    ap.add_argument("--with", dest="with_", default="", help="knobs every setting of the screen carries, e.g. residual_fp32=1 (the accuracy mode)")
    ap.set_defaults(out=os.path.join(ROOT, "gpurun_out", "shape_screen.json"))
    argv = sys.argv[1:]
    args = ap.parse_args((["base"] if not any(not a.startswith("-") for a in argv[:1]) else []) + argv)
    env = fwd_ab.build(args)
    hipmem = sys.modules["hipmem"]

    def measure():
        best = {}
        for _ in range(args.passes):
            for key, (us, n) in shape_times(env.profile_once()).items():
                if key not in best or us < best[key][0]:
                    best[key] = (us, n)
        return best

    pre = (args.with_ + ",") if args.with_ else ""
    base_setting = args.with_ or "base"
    env.apply(base_setting)
    base = measure()
    targets = sorted(base, key=lambda k: -base[k][0] * base[k][1])[:args.top]
    cands = {k: candidates(k) for k in targets}
    rounds = max(len(v) for v in cands.values())
    print(f"{len(targets)} shapes, {sum(len(v) for v in cands.values())} candidates in {rounds} rounds", flush=True)
    seen = {k: {} for k in targets}
    for r in range(rounds):
        ov = ";".join(f"{k[0]},{k[1]},{k[2]},{k[3]},{k[4]}:{cands[k][r][0]}:{cands[k][r][1]}" for k in targets if r < len(cands[k]))
        try:
            env.apply(pre + "gemm_override=" + ov)
            got = measure()
        except (env._lib.SdmiError, RuntimeError) as ex:       # a candidate the library refuses for its shape: the whole round is void
            print(f"round {r}: {ex}", flush=True)
            continue
        for k in targets:
            if r < len(cands[k]) and k in got:
                seen[k][cands[k][r]] = got[k][0]
    env.apply(base_setting)
    report, winners = [], []
    for k in targets:
        us0, n = base[k]
        best = min(seen[k].items(), key=lambda kv: kv[1]) if seen[k] else (None, us0)
        gain = (us0 - best[1]) * n
        row = {"shape": list(k), "launches": n, "default_us": round(us0, 2), "best": None if best[0] is None else [CFG_NAME[best[0][0]], best[0][1]],
               "best_us": round(best[1], 2), "gain_us_per_forward": round(gain, 1),
               "all": {f"{CFG_NAME[c]}/{s}": round(v, 2) for (c, s), v in sorted(seen[k].items(), key=lambda kv: kv[1])}}
        report.append(row)
        if best[0] is not None and us0 - best[1] > args.min_gain * us0:
            winners.append((k, best[0]))
        print(f"M{k[0]:<7d} N{k[1]:<5d} K{k[2]:<6d} taps {k[3]} kind {k[4]}  x{n:<3d} default {us0:8.1f} us   best {row['best']} {best[1]:8.1f} us   "
              f"{gain:+7.1f} us / forward", flush=True)
    res = {"what": args.what, "model": args.model, "rows": args.rows, "latent": args.hw, "shapes": report, "winners": [[list(k), list(c)] for k, c in winners]}
    if winners:
        ov = ";".join(f"{k[0]},{k[1]},{k[2]},{k[3]},{k[4]}:{c[0]}:{c[1]}" for k, c in winners)
        e0, e1 = hipmem.Event(), hipmem.Event()
        times = {"base": [], "winners": []}
        for rep in range(max(args.reps, 3)):
            for name, setting in (("base", base_setting), ("winners", pre + "gemm_override=" + ov)):
                env.apply(setting)
                env.forward()
                hipmem.sync()
                e0.record()
                for _ in range(args.fwd):
                    env.forward()
                e1.record()
                times[name].append(e1.ms_since(e0) / args.fwd)
        env.apply("base")
        res["verify_ms"] = {k: [round(v, 3) for v in t] for k, t in times.items()}
        res["override"] = ov
        predicted = sum(r["gain_us_per_forward"] for r in report if any(list(k) == r["shape"] for k, _ in winners)) / 1e3
        print(f"{len(winners)} winners, predicted {predicted:.3f} ms per forward; verification: base {statistics.median(times['base']):.3f} ms, "
              f"winners {statistics.median(times['winners']):.3f} ms\noverride: {ov}", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    env.lib.sdmi_engine_destroy(env.handle)


if __name__ == "__main__":
    main()
