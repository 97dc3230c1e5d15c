"""Probe: the largest-coefficient rule (PivotRule::SteepestDescent) in BOTH phases under the safeguards, on the files no leg solves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
import corpus
idx = corpus.index()
for name in sys.argv[1:] or ["TUFF", "DEGEN3", "CYCLE", "DFL001", "PILOT87"]:
    md, fixed = corpus.load(name)
    for data, prov in (("read", md), ("scaled", md.scaled()[0])):
        for ename, kind in (("lu", engine.ENGINE_LU), ("tableau", engine.ENGINE_TABLEAU)):
            t0 = time.perf_counter()
            try:
                t = engine.Tableau(prov, config=engine.robust_config(engine=kind, phase_one_rule=engine.STEEPEST_DESCENT, phase_two_rule=engine.STEEPEST_DESCENT))
                total, oc = 0, engine.RUNNING
                while time.perf_counter() - t0 < 45:
                    done, oc = t.run(20000); total += done
                    if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE): break
                obj = t.objective_function_value() + fixed
                chk = t.check_basis() if oc == engine.OPTIMAL else None
                print(name, data, ename, engine.OUTCOME_NAMES.get(oc, oc), total, "phase", t.phase, f"{obj:.10g}", idx[name]["highs_objective"], chk, f"{time.perf_counter()-t0:.0f}s", flush=True)
                t.close()
            except engine.RelpError as e:
                print(name, data, ename, "error", str(e)[:90], flush=True)
