#!/usr/bin/env python3
"""Turn the rocprofv3 outputs under gpurun_out/ (rocpd sqlite) into the committed summaries under profiles/.

    python tools/summarize_profiles.py r02     # reads gpurun_out/prof_stats, prof_pmc_fetch, prof_pmc_write (+ *_k0 A/B passes)

Writes profiles/<round>_kernel_stats.md   (rocprofv3 --kernel-trace --stats of `bench.py --steps 1 --warmup 1`)
       profiles/<round>_pmc_traffic.md     (separate --pmc FETCH_SIZE / WRITE_SIZE passes over the SAME workload bench.py times;
                                            FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM": gfx950 reports half the bytes of
                                            wide coalesced reads); 3x3-conv launches (the KORD = true ping-pong instantiations) are
                                            set against their algorithmic bytes from gpurun_out/bench_kernels_c1.json
       profiles/<round>_pmc_traffic.json   (per-launch HBM bytes of the implicit-GEMM family + the workload key; read by bench.py)
"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def short(n):
    n = n.replace("void sdmi::", "").replace("sdmi::", "")
    if n.startswith("_ZN4sdmi"):
        import re
        m = re.match(r"_ZN4sdmi(\d+)", n)
        k = int(m.group(1))
        n = n[len(m.group(0)):len(m.group(0)) + k]
    return n[:90]


def first_db(d):
    for f in sorted(os.listdir(d)):
        if f.endswith(".db"):
            return os.path.join(d, f)
    raise FileNotFoundError(d)


def config_stats(rnd, configs):
    """profiles/<round>_kernel_stats_<config>.md from gpurun_out/prof_stats_<config> (rocprofv3 --kernel-trace --stats of
    `bench.py --config <config> --steps 1 --warmup 0`), plus the PMC traffic of the same workload when bench.py --pmc-traffic left it."""
    what = {"c2": "SD1.5 512x512, 50-step DPM++ 2M Karras, batch 8", "c3": "SDXL-base 1024x1024, 30-step Euler a, batch 4",
            "c4a": "SD1.5 txt2img 512x512 + hires-fix x2 (20 + 20 evaluations), batch 8", "c4b": "SD1.5 img2img 512x512, denoise 0.75, batch 8"}
    for c in configs:
        d = os.path.join(G, f"prof_stats_{c}")
        if not os.path.isdir(d):
            continue
        con = sqlite3.connect(first_db(d))
        rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        tot = sum(r[2] for r in rows)
        with open(os.path.join(P, f"{rnd}_kernel_stats_{c}.md"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats — `python bench.py --config {c} --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-dropin` ({rnd})\n\n")
            f.write(f"One job of {what.get(c, c)} plus the one-time weight packing.  Durations in microseconds.\n\n")
            pt = os.path.join(G, f"pmc_traffic_{c}.json")
            if os.path.exists(pt):
                t = json.load(open(pt))
                f.write(f"HBM traffic of the implicit-GEMM family on this workload (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes taken by "
                        f"`bench.py --config {c} --pmc-traffic` on the same box; 2 x FETCH_SIZE + WRITE_SIZE): **{t['gemm_mfma_bytes_per_launch'] / 1e6:.1f} MB per launch** "
                        f"over {t['gemm_mfma_launches']} launches.\n\n")
            f.write(f"Total kernel time: {tot / 1e3:.1f} ms\n\n| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
            for n, cnt, tt, a, pc in rows[:30]:
                f.write(f"| `{short(n)}` | {cnt} | {tt:.0f} | {a:.1f} | {pc:.2f} |\n")
    print("wrote config stats", configs)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    os.makedirs(P, exist_ok=True)
    if "--configs" in sys.argv:
        return config_stats(rnd, sys.argv[sys.argv.index("--configs") + 1:])
    con = sqlite3.connect(first_db(os.path.join(G, "prof_stats")))
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    with open(os.path.join(P, f"{rnd}_kernel_stats.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats — `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline` ({rnd})\n\n")
        f.write("Two txt2img jobs (1 warm-up + 1 timed) of the C1 workload (SD1.5 512x512, 20-step Euler-a, batch 8) plus the one-time "
                "weight packing.  Durations in microseconds; source: `top_kernels` view of the rocpd database.\n\n")
        f.write(f"Total kernel time: {tot / 1e3:.1f} ms\n\n| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for n, c, t, a, pc in rows[:40]:
            f.write(f"| `{short(n)}` | {c} | {t:.0f} | {a:.1f} | {pc:.2f} |\n")

    def agg(path, counter):
        c = sqlite3.connect(path)
        d = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for name, val, dur in c.execute("select kernel_name, value, duration from counters_collection where counter_name=?", (counter,)):
            a = d[name]
            a[0] += 1; a[1] += val; a[2] += dur
        return d
    fe = agg(first_db(os.path.join(G, "prof_pmc_fetch")), "FETCH_SIZE")
    wr = agg(first_db(os.path.join(G, "prof_pmc_write")), "WRITE_SIZE")
    names = sorted(fe, key=lambda n: -fe[n][2])
    fam = {"calls": 0, "fetch_kb": 0.0, "write_kb": 0.0}
    with open(os.path.join(P, f"{rnd}_pmc_traffic.md"), "w") as f:
        f.write(f"# HBM traffic from PMC counters ({rnd})\n\n`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and a separate `--pmc WRITE_SIZE` pass over "
                "`python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline` — the whole C1 job bench.py times (20 sampler steps).  FETCH_SIZE is DOUBLED below (gfx950 tallies 128-byte requests at 64 B — "
                "MI355X_MICROARCH.md, HBM section); WRITE_SIZE is reported as counted (uncalibrated).\n\n"
                "| kernel | launches | HBM read MB / launch (corrected) | HBM write MB / launch | avg us (profiled pass) |\n|---|---:|---:|---:|---:|\n")
        for n in names[:24]:
            c, fv, fd = fe[n]
            wv = wr.get(n, [1, 0.0, 0.0])
            f.write(f"| `{short(n)}` | {c} | {2 * fv / c / 1024:.1f} | {wv[1] / max(wv[0], 1) / 1024:.1f} | {fd / c / 1e3:.1f} |\n")
            if any(pfx in n for pfx in ("gemm_mfma", "splitk", "rowchain_")):      # bench.py FAMILY_PREFIXES
                fam["calls"] += c; fam["fetch_kb"] += 2 * fv; fam["write_kb"] += wv[1] * c / max(wv[0], 1)
    per_launch = (fam["fetch_kb"] + fam["write_kb"]) * 1024 / max(fam["calls"], 1)
    # the ping-pong instantiations (3x3 convs and the large 1x1 / linear layers share them under the tap-major K order): measured
    # bytes per launch against the algorithmic bytes of the same launches (HIP-event names carry "pp")
    conv = {"calls": 0, "fetch_kb": 0.0, "write_kb": 0.0}
    for n in fe:
        if "pingpong" in n:
            c, fv, _ = fe[n]
            wv = wr.get(n, [1, 0.0, 0.0])
            conv["calls"] += c; conv["fetch_kb"] += 2 * fv; conv["write_kb"] += wv[1] * c / max(wv[0], 1)
    alg = alg_all = None
    bk = os.path.join(G, "bench_kernels_c1.json")
    if os.path.exists(bk):
        allk = [k for k in json.load(open(bk)) if k["name"].startswith(("gemm_mfma", "splitk", "rowchain_"))]
        ks = [k for k in allk if "pp" in k["name"].split(" ")[0]]
        if ks:
            alg = sum(k["bytes"] for k in ks) / max(sum(k["launches"] for k in ks), 1)
        if allk:
            alg_all = sum(k["bytes"] for k in allk) / max(sum(k["launches"] for k in allk), 1)
    out = {"round": rnd, "workload": os.environ.get("SDMI_PMC_WORKLOAD", "c1:20"),
           "gemm_mfma_bytes_per_launch": round(per_launch), "gemm_mfma_launches": fam["calls"],
           "gemm_mfma_algorithmic_bytes_per_launch": round(alg_all) if alg_all else None,
           "pingpong_bytes_per_launch": round((conv["fetch_kb"] + conv["write_kb"]) * 1024 / max(conv["calls"], 1)) if conv["calls"] else None,
           "pingpong_launches": conv["calls"], "pingpong_algorithmic_bytes_per_launch": round(alg) if alg else None,
           "note": "2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, averaged per launch over the PMC passes of the named workload; algorithmic = "
                   "activations read once + weights + output written once per launch (launch_gemm's ProfScope bytes)"}
    if out["pingpong_bytes_per_launch"] and alg:
        out["pingpong_traffic_over_algorithmic"] = round(out["pingpong_bytes_per_launch"] / alg, 3)
    if alg_all:
        out["gemm_mfma_traffic_over_algorithmic"] = round(per_launch / alg_all, 3)
    # optional A/B passes of the two K orders of the 3x3 convs (SDMI_CONV_KORDER=0 tap-major, the default; 1 channel-block-major)
    k0f, k0w = os.path.join(G, "prof_pmc_fetch_k0"), os.path.join(G, "prof_pmc_write_k0")
    if os.path.isdir(k0f) and os.path.isdir(k0w):
        f0, w0 = agg(first_db(k0f), "FETCH_SIZE"), agg(first_db(k0w), "WRITE_SIZE")
        tot = {"calls": 0, "kb": 0.0}
        for n in f0:
            if "gemm_mfma" in n:
                c, fv, _ = f0[n]
                wv = w0.get(n, [1, 0.0, 0.0])
                tot["calls"] += c; tot["kb"] += 2 * fv + wv[1] * c / max(wv[0], 1)
        out["tap_major_ab"] = {"gemm_mfma_bytes_per_launch": round(tot["kb"] * 1024 / max(tot["calls"], 1)), "launches": tot["calls"],
                               "workload": "c1 with --sampler-steps 2 (both K orders measured on this shortened job for the A/B)"}
    k1f, k1w = os.path.join(G, "prof_pmc_fetch_k1"), os.path.join(G, "prof_pmc_write_k1")
    if os.path.isdir(k1f) and os.path.isdir(k1w):
        f1, w1 = agg(first_db(k1f), "FETCH_SIZE"), agg(first_db(k1w), "WRITE_SIZE")
        tot = {"calls": 0, "kb": 0.0}
        for n in f1:
            if "gemm_mfma" in n:
                c, fv, _ = f1[n]
                wv = w1.get(n, [1, 0.0, 0.0])
                tot["calls"] += c; tot["kb"] += 2 * fv + wv[1] * c / max(wv[0], 1)
        out["channel_major_ab"] = {"gemm_mfma_bytes_per_launch": round(tot["kb"] * 1024 / max(tot["calls"], 1)), "launches": tot["calls"]}
    json.dump(out, open(os.path.join(P, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(G, "pmc_traffic.json"), "w"), indent=1)
    # MFMA utilisation / effective clock per kernel (third PMC pass, optional)
    mdir = os.path.join(G, "prof_pmc_mfma")
    if os.path.isdir(mdir):
        c = sqlite3.connect(first_db(mdir))
        d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
        for name, cn, val, dur in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            a = d[name][cn]
            a[0] += 1; a[1] += val; a[2] += dur
        rows = []
        for name, cs in d.items():
            if "GRBM_GUI_ACTIVE" not in cs or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs:
                continue
            n, gui, dur = cs["GRBM_GUI_ACTIVE"]
            mf = cs["SQ_VALU_MFMA_BUSY_CYCLES"][1]
            cyc = gui / 8.0                                 # counter is summed over the 8 XCDs
            rows.append((dur, name, n, dur / n / 1e3, cyc / max(dur, 1) * 1e0, mf / max(cyc * 1024, 1)))
        rows.sort(reverse=True)
        with open(os.path.join(P, f"{rnd}_pmc_mfma.md"), "w") as f:
            f.write(f"# MFMA utilisation and effective clock per kernel ({rnd})\n\n`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE "
                    "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES` over `python bench.py --steps 1 --warmup 0 --sampler-steps 2 "
                    "--no-cpu-baseline --no-roofline`.  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA util = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) (the counter adds 16 cycles per 16x16x32 MFMA, 32 per 32x32x16).\n\n"
                    "| kernel | launches | avg us | clock GHz | MFMA util |\n|---|---:|---:|---:|---:|\n")
            for dur, name, n, avg, ghz, util in rows[:16]:
                f.write(f"| `{short(name)}` | {n} | {avg:.1f} | {ghz:.2f} | {util:.3f} |\n")
    print("wrote", os.listdir(P))


if __name__ == "__main__":
    main()
