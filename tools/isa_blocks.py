#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a -save-temps .s (default: newest /tmp/aether_isa_*)."""
import glob, os, re, sys
from collections import Counter
pat = sys.argv[1] if len(sys.argv) > 1 else "k_fusedILi2ELi8ELi3"
f = sys.argv[2] if len(sys.argv) > 2 else sorted(glob.glob('/tmp/aether_isa_*/*.s'), key=os.path.getmtime)[-1]
txt = open(f).read()
m = re.search(r'^(\S*' + re.escape(pat) + r'\S*):', txt, re.M)
start = m.start(); end = txt.find('s_endpgm', start)
body = txt[start:end].split('\n')
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_sqrt', 'v_rsq', 'v_sin', 'v_cos')): return 'trans'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    return 'other'
blocks = []; cur = ['entry', []]
for ln in body[1:]:
    mm = re.match(r'^(\.LBB\d+_\d+):', ln)
    if mm:
        blocks.append(cur); cur = [mm.group(1), []]
    else:
        t = ln.strip()
        if t and not t.startswith(('.', ';', '//')):
            cur[1].append(t.split()[0])
blocks.append(cur)
tot = Counter()
for lab, ops in blocks:
    c = Counter(cls(o) for o in ops)
    tot.update(c)
    if c.get('mfma', 0) >= 16:
        print(f"{lab:12s} n={len(ops):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("total", dict(tot))
big = max(blocks, key=lambda b: sum(1 for o in b[1] if o.startswith('v_mfma')))
print("top ops in the largest MFMA block:", Counter(o for o in big[1] if cls(o) != 'mfma').most_common(30))
