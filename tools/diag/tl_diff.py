"""per-launch difference of two step timelines (tools/step_timeline.py outputs): tl_diff.py OFF.txt ON.txt
launches that exist on one side only (finalize launches the other build no longer has) are listed as dropped / added"""
import re
import sys


def load(p):
    rows = []
    for line in open(p):
        m = re.match(r"\s*(\d+) (\S+)\s+g=.*?\s([\d.]+) us", line)
        if m:
            rows.append((m.group(2), float(m.group(3))))
    return rows


def key(n):
    n = re.sub(r"(conv3x3_fast_kernel<\d+,\d+,\d+),[567],", r"\1,1,", n)
    return re.sub(r"(bnrelu_bwd_pool_kernel<bf16,\w+,\w+),\w+>", r"\1>", n)


off, on = load(sys.argv[1]), load(sys.argv[2])
i = j = 0
drop = add = delta = 0.0
while i < len(off) or j < len(on):
    a = off[i] if i < len(off) else None
    b = on[j] if j < len(on) else None
    if a and b and key(a[0]) == key(b[0]):
        d = b[1] - a[1]
        delta += d
        if abs(d) > 0.6:
            print("%-52s %8.1f %8.1f %+6.1f" % (b[0][:52], a[1], b[1], d))
        i += 1
        j += 1
    elif a and (b is None or any(key(a[0]) != key(x[0]) for x in on[j:j + 1]) and not any(key(a[0]) == key(x[0]) for x in on[j:j + 3])):
        print("%-52s %8.1f %8s" % (a[0][:52], a[1], "-"))
        drop += a[1]
        i += 1
    else:
        print("%-52s %8s %8.1f" % (b[0][:52], "-", b[1]))
        add += b[1]
        j += 1
print("only in OFF %.1f us, only in ON %.1f us, common launches %+.1f us" % (drop, add, delta))
