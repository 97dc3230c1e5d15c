"""Print the `sparse_engine.scale` section of a bench line (bench.py: sparse_scale)."""
import json
import sys

d = json.load(open(sys.argv[1]))
s = d["sparse_engine"]["scale"]
print(s["workload"])
for k in ("lu", "lu_product_form_fallback", "tableau", "revised"):
    r = s[k]
    print(f"{k:26s} {r['value']:9.0f} it/s  {r['pivots']:6d} pivots  {r['seconds']:7.2f} s run  {r['create_and_run_seconds']:7.2f} s with create  "
          f"{r.get('kernel_layout', '')} same pivots {r.get('first_250_pivots_equal_the_lu_engines')} clocks/pivot {r.get('pivot_kernel_clocks_per_pivot')}")
print("lu / fallback", s["lu_over_fallback"], " lu / tableau", s["lu_over_tableau"])
if "cpu_baseline" in s:
    c = s["cpu_baseline"]
    print("cpu port", round(c["value"]), "it/s; LU engine over the same stretch", round(c.get("lu_engine_value_over_sample", 0)), ";", c["sample"], "; same 5000 pivots:", c["lu_engine_takes_the_same_5000_pivots"], c["objective_after_sample"], c["lu_engine_objective_after_sample"])
print("dense10k", round(d["value"]), "c4", round(d["c4"]["value"]), "25fv47 lu", round(d["sparse_engine"]["value"]))
