import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
import corpus
for name, kind in (("GREENBEA", engine.ENGINE_TABLEAU), ("BNL1", engine.ENGINE_TABLEAU), ("BNL1", engine.ENGINE_LU), ("BNL1", engine.ENGINE_REVISED)):
    md, fixed = corpus.load(name)
    cfg = engine.robust_config(); cfg.engine = kind
    t = engine.Tableau(md, config=cfg)
    tot = 0
    for step in range(60):
        done, oc = t.run(20000)
        tot += done
        print(name, kind, "run ->", done, engine.OUTCOME_NAMES.get(oc, oc), "phase", t.phase, "iterations", t.iterations(), t.robust_stats(), "obj", t.objective_function_value() + fixed, flush=True)
        if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE) or done == 0:
            break
    t.close()
