#!/usr/bin/env python3
"""A first-stage decode too large for any oracle, checked through a size-independent property (round 6).

With p.tiling (every padded 3x3 conv circular, modules/sd_hijack.py:311-318) the decoder is equivariant to translations by whole
periods, so the decode of a latent that repeats a 64x64 tile k x k times IS the 512x512 decode of that tile repeated k x k times:
circular convs, nearest upsampling, GroupNorm statistics (a periodic image has its tile's mean and variance) and the mid-block attention
(a softmax over k*k copies of the same keys is the softmax over one copy) all commute with the repetition.  So

    decode(tile(z, k)) == tile(decode(z), k)      up to fp16 rounding of other GEMM dispatches / summation orders

and the left side at k = 8 is a 4096x4096 decode: GroupNorm in bands of rows (2^31 elements per image at 128 channels, norm.hip
groupnorm_banded), the mid-block attention over 262144 tokens in blocks of query rows (engine.cpp run_vae_attn), 32-bit offsets
everywhere else.

    python tools/gpu/decode_big.py [--k 8] [--out gpurun_out/r06_decode_big.json]
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def sub(name):
    return importlib.import_module("stable-diffusion-webui_amd." + name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_decode_big.json"))
    args = ap.parse_args()
    import torch
    schema, engine = sub("schema"), sub("engine")
    vcfg = schema.sd15_vae()
    sd = schema.synthetic_state_dict(None, vcfg, dtype=torch.float16)
    eng = engine.Engine(0)
    eng.load_vae(vcfg, sd, decoder_only=True)
    eng.set_option("tiling", 1)
    eng.set_option("arena_reuse", 1)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(606)
    z = (torch.randn(1, 4, 64, 64, generator=g) * 0.18215 * 4.5).to(dev)
    base = eng.vae_decode(z)
    torch.cuda.synchronize()
    res = {"tile": "z [1,4,64,64] -> 512x512 with p.tiling", "runs": []}
    for k in args.k:
        free0 = torch.cuda.mem_get_info()[0]
        zz = z.repeat(1, 1, k, k).contiguous()
        row = {"k": k, "image": f"{512 * k}x{512 * k}", "latent_tokens_in_mid_block": (64 * k) ** 2}
        try:
            torch.cuda.synchronize()
            t0 = time.time()
            big = eng.vae_decode(zz)
            torch.cuda.synchronize()
            row["seconds_first_call"] = round(time.time() - t0, 3)
            t0 = time.time()
            big = eng.vae_decode(zz)
            torch.cuda.synchronize()
            row["seconds"] = round(time.time() - t0, 3)
            want = base.repeat(1, 1, k, k)
            num = float((big - want).double().pow(2).sum().sqrt())
            den = float(want.double().pow(2).sum().sqrt())
            row["rel_l2_vs_tiled_512_decode"] = num / den
            row["max_abs_diff"] = float((big - want).abs().max())
            row["finite"] = bool(torch.isfinite(big).all())
            row["hbm_used_gb"] = round((free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30, 1)
            del big, want
        except Exception as ex:                                # a size the box cannot hold is a result, not a crash
            row["error"] = f"{type(ex).__name__}: {ex}"[:400]
        res["runs"].append(row)
        print(row, flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)
        if "error" in row:
            break
    eng.close()


if __name__ == "__main__":
    main()
