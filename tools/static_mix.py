#!/usr/bin/env python3
"""Static instruction mix of the hot loops, from the gfx950 assembly hipcc produces for csrc/*.hip (no GPU needed):

    python tools/static_mix.py > profiles/r01_static_mix.md

For the flash-attention kernel the two basic blocks holding the MFMAs (S^T = K Q^T, then softmax + O^T += V^T P^T) are one
KV tile of one wave; for the ping-pong GEMM the span between the first and the last MFMA section is one K tile.  Cycle
estimates: full-rate VALU 4 cycles per wave64 instruction, transcendental (v_exp_f32) 16, MFMA 32x32x16 f16 32, 16x16x32 f16 16.
"""
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stable-diffusion-webui_amd", "csrc")


def assembly(name):
    out = os.path.join(tempfile.mkdtemp(prefix="sdmi_mix_"), name + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(CSRC, name + ".hip")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log"):
        return "trans"
    if op.startswith("ds_read"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith(("global_", "buffer_")):
        return "vmem"
    if op in ("s_waitcnt", "s_barrier", "s_setprio", "s_nop"):
        return "sync"
    return "salu" if op.startswith("s_") else "valu"


def mix(instrs):
    c = collections.Counter(classify(i) for i in instrs)
    return c, collections.Counter(i for i in instrs if classify(i) in ("valu", "trans"))


def kernel_body(text, mangled):
    m = re.search(r"^%s:[^\n]*\n(.*?)\.Lfunc_end" % re.escape(mangled), text, re.S | re.M)
    return m.group(1) if m else None


def metadata(text, mangled):
    block = text[text.index("amdhsa.kernels:"):]
    for b in block.split("  - .agpr_count:")[1:]:
        if re.search(r"\.name:\s+%s\b" % re.escape(mangled), b):
            f = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
            return f("vgpr_count"), f("sgpr_count"), f("private_segment_fixed_size")
    return None


def main():
    print("# Static instruction mix of the hot loops (gfx950 assembly of the committed sources; `tools/static_mix.py`)\n")
    attn = assembly("attention")
    print("## Flash attention: one 64-key KV tile of one wave (32 queries)\n")
    print("| kernel | VGPRs | MFMA 32x32x16 | v_exp | other VALU | ds_read | est. MFMA cycles | est. VALU + exp cycles |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for d, var in ((40, 0), (40, 4), (40, 5), (64, 0), (80, 0), (160, 0)):
        name = "_ZN4sdmi16attn_mfma_kernelILi%dELi64ELi%dEEEvNS_5AttnPE" % (d, var)
        body = kernel_body(attn, name)
        if body is None:
            continue
        blocks, cur = [], []
        for ln in body.split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", ln) or ln.startswith("; %bb."):
                blocks.append(cur)
                cur = []
            elif re.match(r"^\s+[a-z]", ln):
                cur.append(ln.split()[0])
        blocks.append(cur)
        has = [k for k, b in enumerate(blocks) if any("mfma" in x for x in b)]
        # the K loop = the block range from the first to the last MFMA; the ragged-tail masking block (a cndmask per score,
        # taken for the last tile only) is left out, a conditional O-rescale block (lazy variants) is counted as taken
        tile = [i for b in blocks[has[0]:has[-1] + 1] if sum(x.startswith("v_cndmask") for x in b) < 8 for i in b]
        c, _ = mix(tile)
        vg = metadata(attn, name)[0]
        print(f"| `attn_mfma_kernel<{d},64,{var}>` | {vg} | {c['mfma']} | {c['trans']} | {c['valu']} | {c['ds_read']} | {32 * c['mfma']} | "
              f"{4 * c['valu'] + 16 * c['trans']} |")
    print("\nMeasured (profiles/r01_kernel_stats.md): the d = 40 self-attention launch (4096 queries and keys, 128 batch-heads) takes ~600 us ="
          " ~1230 SIMD cycles per wave-tile at ~2.1 GHz with 3 waves per SIMD resident: the kernel runs at the sum of its VALU + exp and"
          " roughly half of its MFMA time — VALU / exp bound.\n")
    gemm = assembly("gemm")
    print("## Ping-pong GEMM: one 64-deep K tile of one wave\n")
    print("| kernel | VGPRs | MFMA 16x16x32 | LDS-direct loads | ds_read_b128 | VALU | SALU | est. MFMA cycles | other issue slots |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for bm, bn in ((256, 320), (256, 256), (128, 320)):
        name = "_ZN4sdmi25gemm_mfma_pingpong_kernelILi%dELi%dELb0ELb0EEEvNS_5GemmPE" % (bm, bn)
        body = kernel_body(gemm, name)
        loop = body[body.index("s_setprio 1"):body.rindex("s_setprio 0")]
        ins = [l.split()[0] for l in loop.split("\n") if re.match(r"^\s+[a-z]", l)]
        c, _ = mix(ins)
        vg = metadata(gemm, name)[0]
        print(f"| `gemm_mfma_pingpong_kernel<{bm},{bn}>` | {vg} | {c['mfma']} | {c['vmem']} | {c['ds_read']} | {c['valu']} | {c['salu']} | "
              f"{16 * c['mfma']} | {c['vmem'] + c['ds_read'] + c['valu'] + c['salu'] + c['sync']} |")
    print("\nThe GEMM loops are MFMA-issue dominated (non-MFMA instructions are < 15 % of the MFMA cycles); their measured 40-53 % MFMA"
          " utilisation (profiles/r01_pmc_mfma.md) is waiting — operands, barriers, clock — not instruction overhead.")


if __name__ == "__main__":
    main()
