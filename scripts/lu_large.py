"""Probe: the LU engine (persistent Forrest-Tomlin kernel) on bases beyond 2,400 / 4,096 rows against the other engines.
usage: python scripts/lu_large.py REL_PATH FIXED(0/1) ENGINE PIVOTS [update_block] [ratio_rule] [artificial_removal]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa: E402,F401
from rust_lp_amd import engine  # noqa: E402
from lp_files import load  # noqa: E402

ENG = {"lu": engine.ENGINE_LU, "revised": engine.ENGINE_REVISED, "tableau": engine.ENGINE_TABLEAU}


def main():
    rel, fixed, kind, pivots = sys.argv[1], bool(int(sys.argv[2])), sys.argv[3], int(sys.argv[4])
    block = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    rr = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    ar = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    gf, ex, md, emd = load(rel, fixed=fixed)
    t = engine.Tableau(md, engine=ENG[kind], update_block=block, ratio_rule=rr, artificial_removal=ar)
    print(f"{rel} {kind}: m {t.nr_rows()} n {t.nr_columns()} block {t.update_block()}", flush=True)
    t0 = time.time()
    total = 0
    while total < pivots:
        done, oc = t.run(min(20000, pivots - total))
        total += done
        el = time.time() - t0
        print(f"  {total} pivots {el:.2f}s {total / max(el, 1e-9):.0f} it/s phase {t.phase} objective "
              f"{t.objective_function_value() + float(gf.fixed_cost):.10g} degenerate {t.degenerate_pivots()} "
              f"{engine.OUTCOME_NAMES.get(oc, oc)}", flush=True)
        if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE):
            break
    if kind == "lu":
        print("  ", t.lu_stats())
        try:
            cyc = t.lu_phase_cycles()
            tot = sum(cyc.values())
            print("   clocks/pivot", tot // max(total, 1), {k: round(100.0 * v / max(tot, 1), 1) for k, v in cyc.items()})
        except Exception as e:      # the product-form fallback has no phase clocks
            print("   no phase clocks:", e)


if __name__ == "__main__":
    main()
