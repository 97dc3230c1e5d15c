"""25FV47 on the LU engine with every refactorisation on the device (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
from lp_files import load
gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
block = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t = engine.Tableau(md, engine=engine.ENGINE_LU, update_block=block, trace_capacity=1 << 15)
t.lu_set_device_factorisation(bool(dev))
t0 = time.perf_counter()
oc = t.solve_relaxation()
dt = time.perf_counter() - t0
print(f"block {block} device {dev}: {engine.OUTCOME_NAMES[oc]}, {t.iterations()} pivots in {dt:.3f} s = {t.iterations() / dt:.0f} it/s, objective {t.objective_function_value() + float(gf.fixed_cost):.8f}, "
      f"{t.lu_device_factorisation_stats()}, {t.lu_stats()}")
