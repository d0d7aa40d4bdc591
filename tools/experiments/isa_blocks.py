"""Instruction census of ONE kernel of a hipcc -S listing, per basic block and by class (valu / mfma / lds / vmem / salu / wait / branch):
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o k.s file.hip;  python tools/experiments/isa_blocks.py k.s <mangled kernel name> [-ops] [-v]
-ops: opcode histogram of the largest block; -v: every block.  (round 4: what the vector instructions of the 14-wide-tile kernels were)"""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i,l in enumerate(lines) if l.startswith(sym + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.set ' + sym + '.uses_flat'))
body = lines[start:end]
# split by labels
blocks = []; cur = ('entry', [])
for l in body[1:]:
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.'):
        if re.match(r'^\.LBB\d+_\d+:', s):
            blocks.append(cur); cur = (s.split(':')[0], [])
        continue
    if re.match(r'^\.?LBB\d+_\d+:', s):
        blocks.append(cur); cur = (s.split(':')[0], []); continue
    cur[1].append(s.split()[0])
blocks.append(cur)
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_','buffer_','scratch_','flat_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'
tot = collections.Counter()
for name, ops in blocks:
    c = collections.Counter(cls(o) for o in ops)
    tot += c
    if len(ops) >= 20 or '-v' in sys.argv:
        print(f"{name:12s} n={len(ops):5d} ", dict(c))
print('TOTAL', dict(tot))
if '-ops' in sys.argv:
    big = max(blocks, key=lambda b: len(b[1]))
    oc = collections.Counter(big[1])
    for k,v in oc.most_common(60): print(f"  {k:32s} {v}")
