#!/usr/bin/env python3
"""Instruction mix of the main (MFMA-bearing) loop of a kernel in a hipcc -save-temps .s file.
usage: asm_mix.py file.s <kernel-name-substring> [--dump out.s]"""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E'))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    fn = lines[start:end + 1]
    if '--dump' in sys.argv:
        open(sys.argv[sys.argv.index('--dump') + 1], 'w').write('\n'.join(fn))
    labels = {}
    for i, l in enumerate(fn):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(fn):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            n = sum('v_mfma' in x for x in fn[a:b])
            if n > 20 and (best is None or b - a > best[1] - best[0]):
                best = (a, b)
    a, b = best
    body = [x.strip() for x in fn[a:b] if x.strip() and not x.strip().startswith((';', '.'))]
    c = Counter()
    for x in body:
        op = x.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_accvgpr'): c['accvgpr'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_')): c['vmem'] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_nop'): c['s_nop'] += 1
        elif op.startswith('s_cbranch'): c['branch'] += 1
        elif op.startswith('s_load'): c['smem'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        else: c['other'] += 1
    meta = [l.strip() for l in lines if key in l and ('vgpr_count' in l or 'agpr' in l)]
    print(key, 'loop instrs', len(body), dict(c))


main()
