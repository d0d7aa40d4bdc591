"""Summarise a rocprofv3 --kernel-trace CSV by (kernel symbol, grid): launches, avg/min/max us, registers, LDS.

    python tools/prof_summary.py <dir with *kernel_trace.csv> [out.csv]"""
import collections
import csv
import glob
import sys


def main():
    src = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(src)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "spcl::" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"],
               r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    nmin = min(len(v) for v in agg.values())
    total = sum(sum(v) for v in agg.values())
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_x", "grid_y", "grid_z", "wg_x", "vgpr", "agpr", "lds_bytes", "launches", "avg_us",
                        "min_us", "max_us"])
            for k, v in out:
                w.writerow(list(k) + [len(v), round(sum(v) / len(v), 2), round(min(v), 2), round(max(v), 2)])
    print(f"total spcl kernel time {total:.0f} us over ~{nmin} steps -> {total / nmin:.0f} us/step")
    for k, v in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
        name = k[0].replace("spcl::", "").replace("unsigned short", "bf16")[:58]
        print(f"{name:58s} g={k[1]:>8s},{k[2]:>3s},{k[3]:>3s} wg={k[4]:>4s} v={k[5]:>3s} lds={k[7]:>6s} n={len(v):>4d} "
              f"avg={sum(v) / len(v):7.2f} step={sum(v) / nmin:7.1f}")


if __name__ == "__main__":
    main()
