"""Probe: the three engines on synthetic sparse LPs beyond the row range of the LDS layouts of the persistent pivot kernel.
usage: python scripts/xl_probe.py M N ENGINE PIVOTS [update_block]           random sparse LP (synthetic.sparse_lp)
       python scripts/xl_probe.py mc:V,E,K 0 ENGINE PIVOTS [update_block]    multi-commodity flow (synthetic.multicommodity_lp)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lp_amd  # noqa: E402,F401
from rust_lp_amd import MatrixData, engine, synthetic  # noqa: E402

ENG = {"lu": engine.ENGINE_LU, "revised": engine.ENGINE_REVISED, "tableau": engine.ENGINE_TABLEAU}


def main():
    kind, pivots = sys.argv[3], int(sys.argv[4])
    block = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    if sys.argv[1].startswith("mc:"):
        v, e, k = (int(w) for w in sys.argv[1][3:].split(","))
        md = MatrixData.from_sparse_dict(synthetic.multicommodity_lp(v, e, k, 7))
        m, n = f"mc {v}", f"{e},{k}"
    elif sys.argv[1].startswith("le:"):                      # all rows <=, A >= 0, c < 0: phase 2 (Dantzig pricing) from the first pivot
        m, n = (int(w) for w in sys.argv[1][3:].split(","))
        import numpy as np
        d = synthetic.sparse_lp(m, n, 7, frac_eq=0.0, frac_ge=0.0)
        d["values"] = np.abs(d["values"]); d["b"] = np.abs(d["b"]) + 1.0; d["c"] = -d["c"]
        md = MatrixData.from_sparse_dict(d)
    else:
        m, n = int(sys.argv[1]), int(sys.argv[2])
        md = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, 7))
    t0 = time.time()
    t = engine.Tableau(md, engine=ENG[kind], update_block=block)
    print(f"lp({m},{n}) {kind}: m {t.nr_rows()} n {t.nr_columns()} block {t.update_block()} create {time.time() - t0:.2f}s", flush=True)
    t0 = time.time()
    total = 0
    while total < pivots:
        done, oc = t.run(min(5000, pivots - total))
        total += done
        el = time.time() - t0
        print(f"  {total} pivots {el:.2f}s {total / max(el, 1e-9):.0f} it/s phase {t.phase} objective "
              f"{t.objective_function_value():.10g} degenerate {t.degenerate_pivots()} {engine.OUTCOME_NAMES.get(oc, oc)}", flush=True)
        if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE):
            break
    if kind == "lu":
        print("  ", t.lu_stats())
        try:
            cyc = t.lu_phase_cycles()
            tot = sum(cyc.values())
            print("   clocks/pivot", tot // max(total, 1), {k: round(100.0 * v / max(tot, 1), 1) for k, v in cyc.items()})
        except Exception as e:
            print("   no phase clocks:", e)


if __name__ == "__main__":
    main()
