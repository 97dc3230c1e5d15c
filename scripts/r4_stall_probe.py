"""Objective by chunks of pivots under relp_robust_config: do the files that end at the pivot limit stall (no progress) or crawl?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
import corpus
names = sys.argv[1:] or ["BNL1", "MAROS", "MODSZK1"]
for name in names:
    md, fixed = corpus.load(name)
    for ename, kind in (("tableau", engine.ENGINE_TABLEAU), ("lu", engine.ENGINE_LU)):
        t = engine.Tableau(md, engine=kind, ratio_rule=engine.RATIO_LARGEST_PIVOT, artificial_removal=engine.ARTIFICIAL_TEXTBOOK, pivot_rescue=1, auto_reinversion=1)
        total, line = 0, []
        while total < 40000:
            done, oc = t.run(2000)
            total += done
            line.append(f"{total}:{t.phase}:{t.objective_function_value():.10g}")
            if oc == engine.PHASE_ONE_DONE: continue
            if oc != engine.RUNNING: break
        print(name, ename, engine.OUTCOME_NAMES.get(oc, oc), " ".join(line[:24]), flush=True)
        t.close()
