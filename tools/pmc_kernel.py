"""Mean per-launch value of the given counters for kernels matching a substring, from rocprofv3 --pmc output dirs.
usage: pmc_kernel.py <kernel substring> <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(f"{k:32s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
