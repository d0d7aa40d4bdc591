"""HBM bytes per launch of every libspcl kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB).

    python tools/pmc_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> profiles/r01_k_pmc_hbm_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads -> doubled."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "spcl::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        out[k] = {"FETCH_SIZE": {"mean_per_launch": round(fm, 4), "launches": len(f)},
                  "WRITE_SIZE": {"mean_per_launch": round(wm, 4), "launches": len(w)},
                  "hbm_bytes_per_launch_corrected": int((2 * fm + wm) * 1024),
                  "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB); gfx950 FETCH_SIZE reads 1/2 "
                          "of wide coalesced streams -> x2 (MI355X_MICROARCH.md)"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k[:70]:70s} {v['hbm_bytes_per_launch_corrected'] / 2**20:9.1f} MiB/launch")


if __name__ == "__main__":
    main()
