"""Mean per launch of every counter in a rocprofv3 --pmc output directory, for kernels whose name contains a filter.

    python tools/pmc_summary.py <dir> [name-filter] [min-grid]"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "spcl::"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if flt not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + r.get("Grid_Size", "?")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")


if __name__ == "__main__":
    main()
