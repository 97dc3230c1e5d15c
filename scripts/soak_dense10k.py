"""Usage: soak_dense10k.py [M [N [SEED]]].  Solve the 10,000 x 10,000 bench LP (or M x N) to optimality on the tableau and the revised engine and compare
(objective, pivot count, trace prefix); report residuals of the final tableau state."""
import ctypes as C
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = int(sys.argv[2]) if len(sys.argv) > 2 else m
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20250002
lib = engine.load_library()
b = n * (1000 + (synthetic.splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 4000.0
c = -(1000 + (synthetic.splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 1000.0
ptr = C.c_void_p()
assert lib.relp_device_alloc(C.byref(ptr), m * n * 8) == 0
assert lib.relp_synth_fill_dense(ptr, m, m, n, seed, 0, None) == 0
md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=b, cost=c, upper_bound=np.full(n, np.inf))
res = {}
for name, kind in (("tableau", engine.ENGINE_TABLEAU), ("revised", engine.ENGINE_REVISED)):
    t = engine.Tableau(md, engine=kind, device_dense_ptr=ptr.value, device_dense_ld=m, poll_interval=512, trace_capacity=1 << 17)
    t0 = time.perf_counter()
    oc = t.solve_relaxation()
    dt = time.perf_counter() - t0
    tr = t.trace()
    bb = t.b()
    print(f"{name}: {engine.OUTCOME_NAMES[oc]} {len(tr)} pivots in {dt:.2f} s ({len(tr) / dt:.0f} it/s) objective {t.objective_function_value()!r} "
          f"min b {bb.min():.3e}", flush=True)
    if name == "tableau":
        d = t.relative_costs()
        basis = t.basis_indices()
        nb = np.ones(len(d), dtype=bool); nb[basis] = False
        print(f"  basic reduced costs max {np.max(np.abs(d[basis])):.3e}, nonbasic min {d[nb].min():.3e}", flush=True)
    res[name] = (oc, tr, t.objective_function_value())
    t.close()
a, r = res["tableau"], res["revised"]
first = next((i for i, (x, y) in enumerate(zip(a[1], r[1])) if x != y), None)
print("same outcome", a[0] == r[0], "first trace difference", first, "of", len(a[1]), len(r[1]), "objective rel diff",
      abs(a[2] - r[2]) / abs(r[2]))
