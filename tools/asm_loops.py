#!/usr/bin/env python3
"""Loop structure of a kernel in hipcc's assembly output (hipcc --cuda-device-only -S): every backward branch with the
instruction mix of its body (VALU, fp64 arithmetic, scratch / global / LDS accesses, DPP) -- a static stand-in for a profile
when iterating on register pressure without a GPU:  python tools/asm_loops.py kernel.s [min_len]"""
import re, sys
src = sys.argv[1]; minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ins = []; labels = {}
for l in open(src):
    l = l.rstrip()
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = len(ins); continue
    m = re.match(r'^\s+([a-z_0-9]+)\s*(.*)$', l)
    if m and not l.strip().startswith('.') and not l.strip().startswith(';'):
        ins.append((m.group(1), m.group(2)))
print("instructions", len(ins))
def mix(a, b):
    c = dict(valu=0, f64=0, scratch_ld=0, scratch_st=0, glob=0, lds=0, dpp=0, salu=0, div=0, cnd=0, acc=0, branch=0)
    for op, args in ins[a:b + 1]:
        if op.startswith('v_'): c['valu'] += 1
        if op.startswith('s_') and not op.startswith('s_waitcnt') and not op.startswith('s_nop'): c['salu'] += 1
        if re.match(r'v_(fma|mul|add|fmac|max|min)_f64', op) or op.startswith('v_pk_') : c['f64'] += 1
        if op.startswith('scratch_load'): c['scratch_ld'] += 1
        if op.startswith('scratch_store'): c['scratch_st'] += 1
        if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): c['glob'] += 1
        if op.startswith('ds_'): c['lds'] += 1
        if 'dpp' in op or 'quad_perm' in args: c['dpp'] += 1
        if op.startswith('v_div_') or op.startswith('v_rcp_f64') or op.startswith('v_rsq_f64') or op.startswith('v_sqrt_f64'): c['div'] += 1
        if op.startswith('v_cndmask'): c['cnd'] += 1
        if op.startswith('v_accvgpr'): c['acc'] += 1
        if op.startswith('s_cbranch') or op == 's_branch': c['branch'] += 1
    return c
loops = []
for k, (op, args) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        t = labels.get(args.split()[0].strip())
        if t is not None and t <= k: loops.append((t, k))
loops.sort(key=lambda x: (x[0], -x[1]))
print("whole kernel", mix(0, len(ins) - 1))
for a, b in loops:
    if b - a >= minlen:
        depth = sum(1 for (x, y) in loops if x <= a and y >= b and (x, y) != (a, b))
        print("  " * depth + "loop [%d, %d] len %d" % (a, b, b - a + 1), mix(a, b))
