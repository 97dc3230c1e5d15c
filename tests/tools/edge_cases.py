"""Edge cases through every engine against the f64 oracle: 1 x 1 problems, empty columns, all-zero right-hand
sides (fully degenerate), zero costs, a single equality, bounds only.  Prints every difference."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rust_lp_amd  # noqa: F401
from rust_lp_amd import engine
from oracle import relp_f64

from edge_lps import CASES

bad = 0
for name, problem in CASES.items():
    ref = relp_f64.OracleF64(problem.ensure_csc())
    status = ref.run(10000)
    for kind, block in ((engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 4), (engine.ENGINE_TABLEAU, 4), (engine.ENGINE_LU, 4)):
        try:
            t = engine.Tableau(problem, engine=kind, update_block=block, trace_capacity=256)
            oc = engine.OUTCOME_NAMES[t.solve_relaxation(max_iters=1000)]
            ok = oc == status and t.trace() == ref.trace
            if ok and status == "optimal":
                ok = abs(t.objective_function_value() - ref.objective) <= 1e-9 * max(1.0, abs(ref.objective))
            detail = f"{oc} {t.trace()}"
            t.close()
        except Exception as e:      # noqa: BLE001
            ok, detail = False, f"exception {e}"
        if not ok:
            bad += 1
            print(f"DIFF {name!r} kind {kind} block {block}: {detail} vs oracle {status} {ref.trace}", flush=True)
    print(f"{name:45s} oracle {status:10s} {len(ref.trace)} pivots", flush=True)
print("differences:", bad)
sys.exit(1 if bad else 0)
