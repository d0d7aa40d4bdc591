"""The pre-train -> fine-tune -> validation Dice chain (tests/_dice_chain.py) on the HIP path in both storage types next to the
CPU oracle: per-class Dice, the loss curves and their distances.   python tools/diag/dice_chain.py [key=value ...]
  keys: size bs_pre k_pre bs_ft m_ft val_scans val_slices seed (make_data) / max_channel pre_lr ft_lr (HYPER)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import _dice_chain as DC  # noqa: E402

kw, hyper = {}, dict(DC.HYPER)
for a in sys.argv[1:]:
    k, v = a.split("=")
    v = float(v) if "." in v or "e" in v else int(v)
    (hyper if k in hyper else kw)[k] = v
data = DC.make_data(**kw)
t0 = time.time()
ref = DC.run_oracle(data, hyper)
t1 = time.time()
out = {"args": sys.argv[1:], "oracle_s": round(t1 - t0, 1), "oracle": {"dsc": ref["dsc"], "val_loss": ref["val_loss"],
       "ft_first_last": [ref["ft_curve"][0], ref["ft_curve"][-1]], "pre_curve": ref["pre_curve"]}}
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    got = DC.run_hip(data, dt, hyper)
    rel = np.abs(np.array(got["ft_curve"]) - np.array(ref["ft_curve"])) / np.array(ref["ft_curve"])
    prel = np.abs(np.array(got["pre_curve"]) - np.array(ref["pre_curve"])) / np.array(ref["pre_curve"])
    tail = [round(100 * (a["DSC_mean"] - b["DSC_mean"]), 3) for a, b in zip(got["dice_curve"][-10:], ref["dice_curve"][-10:])]
    out[name] = {"dsc": got["dsc"], "dice_tail_delta_points": tail, "best_score_delta_points": round(100 * (got["best_score"] - ref["best_score"]), 4), "val_loss": got["val_loss"], "ft_first_last": [got["ft_curve"][0], got["ft_curve"][-1]],
                 "dice_delta_points": {c: round(100 * (got["dsc"][c] - ref["dsc"][c]), 4) for c in ref["dsc"]},
                 "ft_curve_max_rel": float(rel.max()), "ft_curve_mean_rel": float(rel.mean()), "pre_curve_max_rel": float(prel.max())}
    if "--curves" in os.environ.get("DICE_CHAIN_FLAGS", ""):
        out[name]["ft_curve"] = got["ft_curve"]
print(json.dumps(out, indent=1))
