"""Per-launch timeline of ONE training step from a rocprofv3 kernel trace of bench.py: the launches in stream order
with their median duration over the traced steps (steps are cut at every launch of the first kernel of a step).
usage: step_timeline.py <dir or kernel_trace.csv> [first-kernel substring, default flip_batch]"""
import csv
import glob
import os
import re
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("spcl::", "").replace("unsigned short", "bf16")
    return re.sub(r"\s+", "", n)


def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        src = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    first = sys.argv[2] if len(sys.argv) > 2 else "flip_batch"
    rows = [r for r in csv.DictReader(open(src))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], None
    for r in rows:
        n = short(r["Kernel_Name"])
        if first in n and (cur is None or len(cur) > 4):
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r))
    lens = sorted(len(s) for s in steps)
    L = lens[len(lens) // 2]
    steps = [s for s in steps if len(s) == L][-20:]
    tot = 0.0
    for i in range(L):
        d = sorted((s[i][2] - s[i][1]) / 1e3 for s in steps)
        gap = sorted((s[i][1] - s[i - 1][2]) / 1e3 for s in steps) if i else [0.0]
        r = steps[-1][i][3]
        med = d[len(d) // 2]
        tot += med
        print(f"{i:3d} {steps[-1][i][0][:58]:58s} g={int(r['Grid_Size_X']):7d},{int(r['Grid_Size_Y']):3d},"
              f"{int(r['Grid_Size_Z']):3d} {med:7.1f} us  gap {gap[len(gap) // 2]:5.1f}")
    span = sorted((s[-1][2] - s[0][1]) / 1e3 for s in steps)
    print(f"launches {L}, sum of medians {tot:.1f} us, step span median {span[len(span) // 2]:.1f} us over {len(steps)} steps")


if __name__ == "__main__":
    main()
