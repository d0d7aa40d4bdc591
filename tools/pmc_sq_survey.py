"""Per-kernel SQ counters of the eager step (rocprofv3 --pmc passes made by tools/diag/pmc_step_sq.sh): total per step and the
ratios that say what a kernel is bound by.   python tools/pmc_sq_survey.py [--json out.json] <dir> [<dir> ...]
  lds_busy  = SQ_LDS_IDX_ACTIVE / CUs / kernel cycles (approx: SQ_BUSY_CYCLES / 32 shader engines)
  conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  valu_busy = 4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs / kernel cycles
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel cycles"""
import collections
import csv
import glob
import json
import sys

json_out = None
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    json_out = sys.argv[i + 1]
    del sys.argv[i:i + 2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "spcl::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spcl::", "")
            agg[k][(d, r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[k][(d, r["Counter_Name"])] += 1
rows = []
for k, c2 in agg.items():
    n = max(cnt[k].values())
    c = {}
    for (d, name), v in c2.items():  # a counter collected in several passes (SQ_BUSY_CYCLES): the mean of the passes
        c.setdefault(name, []).append(v)
    c = {name: sum(v) / len(v) for name, v in c.items()}
    busy = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    if busy <= 0:
        continue
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256.0 / busy
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    valu = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / 1024.0 / busy
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / busy
    rows.append((busy, k, n, lds, conf, valu, mfma))
rows.sort(reverse=True)
print(f"{'kernel':64s} {'launches':>8s} {'Mcycles':>8s} {'lds_busy':>8s} {'conflict':>8s} {'valu_busy':>9s} {'mfma_busy':>9s}")
for busy, k, n, lds, conf, valu, mfma in rows:
    print(f"{k[:64]:64s} {n:8d} {busy / 1e6:8.3f} {lds:8.2f} {conf:8.2f} {valu:9.2f} {mfma:9.2f}")
if json_out:
    json.dump({"method": "rocprofv3 --pmc, two passes (SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU | "
                         "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES) of eager steps (tools/diag/"
                         "pmc_step_sq.sh); busy = SQ_BUSY_CYCLES / 32 shader engines; lds_busy = SQ_LDS_IDX_ACTIVE / 256 CUs / "
                         "busy; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; valu_busy = 4 SQ_ACTIVE_INST_VALU / 1024 "
                         "SIMDs / busy; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / busy",
               "kernels": [{"kernel": k, "launches_counted": n, "busy_Mcycles": round(busy / 1e6, 4), "lds_busy": round(lds, 3),
                            "lds_conflict": round(conf, 3), "valu_busy": round(valu, 3), "mfma_busy": round(mfma, 3)}
                           for busy, k, n, lds, conf, valu, mfma in rows]}, open(json_out, "w"), indent=1)
