"""Print the headline fields of a bench.py JSON line (file argument)."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("value", round(d["value"], 1), d["unit"], "| ms/step", round(d["ms_per_step"], 5), "| windows", d.get("timing", {}).get("window_ms"))
print("flush: avg_us", r.get("avg_us"), "frac", r.get("frac"), "| pivot frac", (r.get("pivot") or {}).get("frac"), "| objective", d.get("objective_after_run"))
for k in ("c4", "c2", "revised_engine", "sparse_engine"):
    if isinstance(d.get(k), dict) and "value" in d[k]:
        print(k, round(d[k]["value"], 1))
