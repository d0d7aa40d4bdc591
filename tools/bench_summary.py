"""print the headline fields of a bench.py JSON line"""
import json
import sys

l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", l["value"], "ms/step", l["ms_per_step"], "replay", l.get("replay_us"))
r = l.get("roofline") or {}
print("roofline:", r.get("kernel"), "frac", r.get("frac"), "avg_us", r.get("avg_us"), "launches", r.get("launches_per_step"))
print("instrumented_step_ms", l.get("instrumented_step_ms"), "extra", l.get("extra"))
for k in l.get("kernel_breakdown", [])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("  ", k)
