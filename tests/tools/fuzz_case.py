"""Re-run one case of tests/tools/fuzz_gpu.py verbosely (same random stream) and compare with HiGHS:
   python tests/tools/fuzz_case.py SEED0 CASE [SIZE_SCALE]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic
from oracle import relp_f64

seed0, target = int(sys.argv[1]), int(sys.argv[2])
SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed0)
KINDS = [(engine.ENGINE_REVISED, (0, 1, 3, 7, 64)), (engine.ENGINE_TABLEAU, (1, 2, 5, 64)), (engine.ENGINE_LU, (1, 2, 6, 64))]
for case in range(target + 1):
    seed = seed0 + case
    if rng.random() < 0.5:
        m, n = SCALE * int(rng.integers(2, 70)), SCALE * int(rng.integers(2, 90))
        make = lambda m=m, n=n, seed=seed: MatrixData.from_dense_le(*[synthetic.dense_lp(m, n, seed)[k] for k in ("A", "b", "c")])
    elif rng.random() < 0.5:
        m, n = SCALE * int(rng.integers(4, 80)), SCALE * int(rng.integers(4, 120))
        kw = dict(nnz_per_col=int(rng.integers(2, 7)), frac_eq=float(rng.uniform(0, 0.5)), frac_ge=float(rng.uniform(0, 0.4)),
                  frac_bounded=float(rng.uniform(0, 0.6)))
        make = lambda m=m, n=n, seed=seed, kw=kw: MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed, **kw))
    else:
        m, n = SCALE * int(rng.integers(6, 70)), SCALE * int(rng.integers(4, 100))
        kw = dict(nnz_per_col=int(rng.integers(2, 6)), frac_eq=float(rng.uniform(0, 0.3)), frac_range=float(rng.uniform(0, 0.3)),
                  frac_ge=float(rng.uniform(0, 0.3)), frac_bounded=float(rng.uniform(0, 0.6)),
                  frac_negative_cost=float(rng.choice([0.0, 0.0, 0.1, 0.3])), infeasible=bool(rng.random() < 0.15))
        make = lambda m=m, n=n, seed=seed, kw=kw: MatrixData.from_sparse_dict(synthetic.mixed_lp(m, n, seed, **kw))
    kind, blocks = KINDS[int(rng.integers(0, 3))]
    block = int(blocks[int(rng.integers(0, len(blocks)))])
md = make()
print("case", target, "m,n", m, n, "kind", kind, "block", block, "counts", md.nr_eq, md.nr_range, md.nr_le, md.nr_ge)
ref = relp_f64.OracleF64(md.ensure_csc() if md.col_ptr is None else md)
print("oracle", ref.run(200000), len(ref.trace), repr(ref.objective), "rows removed", ref.filtered_rows())
try:
    from scipy.optimize import linprog
    D = md.ensure_dense()
    A, ne, nr, nl = D.dense, D.nr_eq, D.nr_range, D.nr_le
    lo = ne + nr + nl
    ub_rows = [A[ne:ne + nr], -A[ne:ne + nr], A[ne + nr:lo], -A[lo:]]
    ub_rhs = [D.b[ne:ne + nr], -(D.b[ne:ne + nr] - D.ranges), D.b[ne + nr:lo], -D.b[lo:]]
    res = linprog(D.cost, A_ub=np.vstack(ub_rows), b_ub=np.concatenate(ub_rhs), A_eq=A[:ne] if ne else None,
                  b_eq=D.b[:ne] if ne else None, bounds=[(0, None if not np.isfinite(u) else u) for u in D.upper_bound], method="highs")
    print("highs", res.status, repr(res.fun))
except Exception as e:      # noqa: BLE001
    print("highs unavailable:", e)
for k, b in ((kind, block), (engine.ENGINE_REVISED, 0)):
    t = engine.Tableau(md, engine=k, update_block=b, trace_capacity=1 << 16)
    oc = engine.OUTCOME_NAMES[t.solve_relaxation(max_iters=20 * len(ref.trace) + 1000)]
    tr = t.trace()
    first = next((i for i, (a, c) in enumerate(zip(tr, ref.trace)) if a != c), None)
    extra = t.check_basis() if oc == "optimal" else None
    print("engine", k, b, oc, len(tr), repr(t.objective_function_value()), "rows", t.nr_rows(), "first diff", first, "check_basis", extra)
