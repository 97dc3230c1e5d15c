"""Probe: Bland's rule (FirstProfitable entering column, the reference's lowest-index leaving column) on the files that cycle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
import corpus
idx = corpus.index()
for name in sys.argv[1:] or ["TUFF", "DEGEN3", "CYCLE"]:
    md, fixed = corpus.load(name)
    smd, r, s = md.scaled()
    for data, prov in (("scaled", smd), ("read", md)):
        for cfg_name, kw in (("bland-literal", dict()), ("bland-largest-pivot", dict(ratio_rule=engine.RATIO_LARGEST_PIVOT, artificial_removal=engine.ARTIFICIAL_TEXTBOOK, pivot_rescue=1, auto_reinversion=1))):
            for ename, kind in (("lu", engine.ENGINE_LU), ("tableau", engine.ENGINE_TABLEAU)):
                t0 = time.perf_counter()
                try:
                    t = engine.Tableau(prov, engine=kind, phase_one_rule=engine.FIRST_PROFITABLE, phase_two_rule=engine.FIRST_PROFITABLE, **kw)
                    total, oc = 0, engine.RUNNING
                    while total < 600000 and time.perf_counter() - t0 < 40:
                        done, oc = t.run(20000); total += done
                        if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE): break
                    obj = t.objective_function_value() + fixed if oc == engine.OPTIMAL else None
                    chk = t.check_basis() if oc == engine.OPTIMAL else None
                    t.close()
                    print(name, data, cfg_name, ename, engine.OUTCOME_NAMES.get(oc, oc), total, obj, idx[name]["highs_objective"], chk, f"{time.perf_counter()-t0:.1f}s", flush=True)
                except engine.RelpError as e:
                    print(name, data, cfg_name, ename, "error", str(e)[:80], flush=True)
