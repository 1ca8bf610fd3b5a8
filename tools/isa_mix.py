"""Instruction mix of the kernels in a hipcc -S listing: python tools/isa_mix.py file.s [name substring ...]"""
import re
import sys
from collections import Counter

KEYS = ['v_mfma_f32_32x32x16_f16', 'v_mfma_f32_16x16x32_f16', 'ds_read_b128', 'ds_read_b64', 'ds_read_b32', 'ds_write_b128', 'global_load_lds_dwordx4',
        'global_load_dwordx4', 'global_load_dwordx2', 'global_store_dwordx2', 'global_store_dwordx4', 'scratch_store_dwordx4', 'scratch_load_dwordx4',
        'scratch_store_dword', 'scratch_load_dword', 's_barrier', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32', 'v_accvgpr_mov_b32', 's_waitcnt',
        'v_exp_f32', 'v_rcp_f32', 'v_mov_b32', 's_nop', 'v_cvt_f16_f32', 'v_cvt_pk_f16_f32', 'v_pk_mul_f32', 'v_fma_f32', 'v_fmac_f32', 'v_mul_f32']


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want and not any(w in name for w in want):
            continue
        lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
        c = Counter(l.split()[0] for l in lines)
        print(name, len(lines), 'instructions')
        for k in KEYS:
            if c.get(k):
                print('    %-28s %d' % (k, c[k]))


if __name__ == '__main__':
    main()
