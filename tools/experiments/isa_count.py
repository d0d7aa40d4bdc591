"""Count instructions by class in one kernel of a hipcc -save-temps .s file:  isa_count.py file.s <substring of name>"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
inside = False
cnt = collections.Counter()
ops = collections.Counter()
for ln in src:
    if re.match(r"^_Z\S*:", ln):
        inside = key in ln
        continue
    if not inside:
        continue
    if ln.strip().startswith(".end_amdhsa_kernel") or ln.startswith("\t.section"):
        inside = False
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", ln)
    if not m or ln.strip().startswith((".", ";")):
        continue
    op = m.group(1)
    ops[op] += 1
    if op.startswith("v_mfma"): c = "mfma"
    elif op.startswith("v_"): c = "valu"
    elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): c = "wait/nop"
    elif op.startswith("s_load") or op.startswith("s_buffer"): c = "smem"
    elif op.startswith("s_"): c = "salu"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
    else: c = "other"
    cnt[c] += 1
print(dict(cnt))
print(ops.most_common(40))
