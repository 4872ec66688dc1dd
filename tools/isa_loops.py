#!/usr/bin/env python3
"""Static ISA statistics of one kernel of a disassembled code object: instruction count, calls, loops (backward branches) and the
scratch (register spill) loads / stores inside each, used for DESIGN.md 4.6:
  clang-offload-bundler --unbundle --type=o --input=mjpcx.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=k.co
  llvm-objdump -d --no-show-raw-insn k.co > k.s;  python tools/isa_loops.py k.s <mangled-name-substring>"""
import re,collections,itertools,sys
src=sys.argv[1]; fn=sys.argv[2]
lines=open(src).read().split('\n')
starts=[(i,l) for i,l in enumerate(lines) if re.match(r'^[0-9a-f]+ <.*>:',l)]
for k,(i,l) in enumerate(starts):
    if fn in l:
        e=starts[k+1][0] if k+1<len(starts) else len(lines); break
body=lines[i+1:e]
ins=[]
for l in body:
    m=re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-F ]+(<.*>)?',l)
    if m: ins.append((int(m.group(3),16),m.group(1),m.group(2),m.group(4) or ''))
base=ins[0][0]
idx={a-base:k for k,(a,_,_,_) in enumerate(ins)}
print('n',len(ins),'bytes',ins[-1][0]-base)
calls=[k for k,(a,op,_,_) in enumerate(ins) if op=='s_swappc_b64']
print('calls',calls)
loops=[]
for k,(a,op,args,sym) in enumerate(ins):
    if op.startswith('s_cbranch') or op=='s_branch':
        m=re.search(r'\+0x([0-9a-f]+)>',sym)
        if m:
            tk=idx.get(int(m.group(1),16))
            if tk is not None and tk<k: loops.append((tk,k))
loops.sort(key=lambda x:(x[0],-x[1]))
sc=[1 if op.startswith('scratch_') else 0 for _,op,_,_ in ins]
cum=[0]+list(itertools.accumulate(sc))
for (a,b) in loops:
    if b-a>400: print('loop',a,b,'len',b-a,'scratch',cum[b+1]-cum[a], 'calls', [c for c in calls if a<=c<=b])
def rng(a, b):
    ld = collections.Counter(); st = collections.Counter()
    for k in range(a, b + 1):
        _, op, args, _ = ins[k]
        mo = re.search(r'offset:(\d+)', args)
        slot = mo.group(1) if mo else '0'
        if op.startswith('scratch_load'): ld[slot] += 1
        if op.startswith('scratch_store'): st[slot] += 1
    return ld, st
print('scratch_load', sum(op.startswith('scratch_load') for _, op, _, _ in ins), 'scratch_store', sum(op.startswith('scratch_store') for _, op, _, _ in ins))
# outermost loops that spill: loads / stores per trip and how many of the reloaded slots are loop-invariant
seen_end = -1
for (a, b) in loops:
    if b - a > 400 and cum[b + 1] - cum[a] > 0 and a > seen_end:
        ld, st = rng(a, b)
        print((a, b), 'loads', sum(ld.values()), 'stores', sum(st.values()), 'slots loaded', len(ld), 'of them never stored inside', len([x for x in ld if x not in st]))
        seen_end = b
