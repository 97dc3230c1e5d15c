"""Probe: does DFL001 converge with a larger budget under the largest-coefficient rule in both phases? (tableau engine, safeguards)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
import corpus
idx = corpus.index()
cap = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
for name in sys.argv[2:] or ["DFL001"]:
    md, fixed = corpus.load(name)
    for data, prov in (("scaled", md.scaled()[0]), ("read", md)):
        t0 = time.perf_counter()
        t = engine.Tableau(prov, config=engine.robust_config(engine=engine.ENGINE_TABLEAU, phase_one_rule=engine.STEEPEST_DESCENT, phase_two_rule=engine.STEEPEST_DESCENT))
        total, oc, marks = 0, engine.RUNNING, []
        try:
            while time.perf_counter() - t0 < cap:
                done, oc = t.run(100000); total += done
                marks.append(f"{total}:{t.phase}:{t.objective_function_value() + fixed:.8g}")
                if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE): break
            chk = t.check_basis() if oc == engine.OPTIMAL else None
            print(name, data, engine.OUTCOME_NAMES.get(oc, oc), total, idx[name]["highs_objective"], chk, t.robust_stats(), " ".join(marks), f"{time.perf_counter()-t0:.0f}s", flush=True)
        except engine.RelpError as e:
            print(name, data, "error", str(e)[:100], " ".join(marks[-6:]), flush=True)
        t.close()
