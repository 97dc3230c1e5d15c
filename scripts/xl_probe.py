"""Probe: the three engines on a synthetic Netlib-shaped sparse LP beyond the persistent FT kernel's row reach.
usage: python scripts/xl_probe.py M N ENGINE PIVOTS [update_block]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lp_amd  # noqa: E402,F401
from rust_lp_amd import MatrixData, engine, synthetic  # noqa: E402

ENG = {"lu": engine.ENGINE_LU, "revised": engine.ENGINE_REVISED, "tableau": engine.ENGINE_TABLEAU}


def main():
    m, n, kind, pivots = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    block = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    md = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, 7))
    t0 = time.time()
    t = engine.Tableau(md, engine=ENG[kind], update_block=block)
    print(f"sparse_lp({m},{n}) {kind}: m {t.nr_rows()} n {t.nr_columns()} block {t.update_block()} create {time.time() - t0:.2f}s", flush=True)
    t0 = time.time()
    total = 0
    while total < pivots:
        done, oc = t.run(min(2000, pivots - total))
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
