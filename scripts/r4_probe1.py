"""Round-4 probe: SCORPION under the literal and the textbook rules on all three engines."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
import corpus
from oracle import relp_f64

md, fixed = corpus.load("SCORPION")
rec = corpus.index()["SCORPION"]
print("SCORPION", rec)
for ar in (0, 1):
    o = relp_f64.OracleF64(md, artificial_removal=ar)
    st = o.run()
    print(f"oracle artificial_removal={ar}: {st}, {len(o.trace)} pivots, objective {o.objective + fixed:.10f}, removed {o.filtered_rows()}, zero-level {o.nr_zero_level_pivots}, exchanges {o.nr_position_exchanges}")
    for kind, block, nm in ((engine.ENGINE_REVISED, 0, "revised"), (engine.ENGINE_TABLEAU, 32, "tableau"), (engine.ENGINE_LU, -1, "lu")):
        t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 15, artificial_removal=ar)
        oc = t.solve_relaxation()
        tr = t.trace()
        same = next((k for k, (a, b) in enumerate(zip(tr, o.trace)) if a != b), min(len(tr), len(o.trace)))
        print(f"  {nm}: {engine.OUTCOME_NAMES[oc]}, {t.iterations()} pivots, objective {t.objective_function_value() + fixed:.10f}, rows {t.nr_rows()} of {md.nr_rows}, "
              f"same as oracle for {same} of {len(o.trace)}, check_basis {t.check_basis()}")
        t.close()
