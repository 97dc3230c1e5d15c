"""Debug probe: 25FV47 on the LU engine with the device factorisation, resident vs downloaded schedules."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa: E402,F401
from rust_lp_amd import engine  # noqa: E402
from lp_files import load  # noqa: E402

gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
block = int(sys.argv[1]) if len(sys.argv) > 1 else -1
out = {}
rr = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for mode, fuse in (("2", "0"), ("1", "256"), ("0", "0")):
    os.environ["RELP_LU_DEVICE_FACTOR"] = mode
    os.environ["RELP_FUSE_LANES"] = fuse
    t = engine.Tableau(md, engine=engine.ENGINE_LU, update_block=block, trace_capacity=1 << 15, ratio_rule=rr)
    total = 0
    while True:
        done, oc = t.run(200)
        total += done
        ident, basic, min_b = t.check_basis()
        if ident > 1e-6 or min_b < -1e-3 or oc not in (engine.RUNNING, engine.PHASE_ONE_DONE) or total > 20000:
            print(f"mode {mode} fuse {fuse} block {block}: {total} pivots, outcome {engine.OUTCOME_NAMES.get(oc)}, |B^-1 B - I| {ident:.2e}, min b {min_b:.2e}, "
                  f"objective {t.objective_function_value():.8g}, stats {t.lu_device_factorisation_stats()}")
            break
    out[mode] = t.trace()
    t.close()
a, b = out["2"], out["1"]
c = out["0"]
print("host-factor unfused vs device-resident: identical for", next((k for k, (x, y) in enumerate(zip(c, b)) if x != y), min(len(c), len(b))))
same = next((k for k, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
print("traces identical for the first", same, "pivots of", len(a), len(b))
