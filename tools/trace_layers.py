"""Median kernel durations of a rocprofv3 kernel trace of tools/bench_kernels.py, keyed by (kernel, call order block).
usage: trace_layers.py <kernel_trace.csv> [name filter]"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spcl::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
# bench_kernels runs each timed lambda 23 times in a row; a lambda may be several kernels: bucket by (name, block index)
blocks = collections.OrderedDict()
count = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spcl::", "").replace("unsigned short", "bf16")
    if flt and flt not in n:
        continue
    count[n] += 1
    b = (count[n] - 1) // 23
    blocks.setdefault((n, b), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, b), v in blocks.items():
    v = sorted(v)
    print(f"{n:44s} block {b:2d} n={len(v):3d} median {v[len(v) // 2]:7.1f} min {v[0]:7.1f}")
