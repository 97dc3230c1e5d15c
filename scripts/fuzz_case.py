"""Re-run one fuzz case verbosely: python scripts/fuzz_case.py SEED0 CASE"""
import sys
sys.path.insert(0, ".")
import numpy as np
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic
from oracle import relp_f64

seed0, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
KINDS = [(engine.ENGINE_REVISED, (0, 1, 3, 7, 64)), (engine.ENGINE_TABLEAU, (1, 2, 5, 64)), (engine.ENGINE_LU, (1, 2, 6, 64))]
for case in range(target + 1):
    seed = seed0 + case
    if rng.random() < 0.5:
        m, n = int(rng.integers(2, 70)), int(rng.integers(2, 90))
        mk = lambda: MatrixData.from_dense_le(*(lambda lp: (lp["A"], lp["b"], lp["c"]))(synthetic.dense_lp(m, n, seed)))
    else:
        m, n = int(rng.integers(4, 80)), int(rng.integers(4, 120))
        args = dict(nnz_per_col=int(rng.integers(2, 7)), frac_eq=float(rng.uniform(0, 0.5)), frac_ge=float(rng.uniform(0, 0.4)),
                    frac_bounded=float(rng.uniform(0, 0.6)))
        mk = lambda: MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed, **args))
    kind, blocks = KINDS[int(rng.integers(0, 3))]
    block = int(blocks[int(rng.integers(0, len(blocks)))])
md = mk()
print("case", target, "m,n", m, n, "kind", kind, "block", block, "counts", md.nr_eq, md.nr_range, md.nr_le, md.nr_ge)
ref = relp_f64.OracleF64(md.ensure_csc() if md.col_ptr is None else md)
print("oracle", ref.run(200000), len(ref.trace), ref.objective, "m after", ref.m)
for k, b in ((kind, block), (engine.ENGINE_REVISED, 0)):
    t = engine.Tableau(md, engine=k, update_block=b, trace_capacity=1 << 16)
    oc = engine.OUTCOME_NAMES[t.solve_relaxation()]
    tr = t.trace()
    first = next((i for i, (a, c) in enumerate(zip(tr, ref.trace)) if a != c), None)
    print("engine", k, b, oc, len(tr), t.objective_function_value(), "rows", t.nr_rows(), "first diff", first,
          None if first is None else (tr[first], ref.trace[first]))
    if first is not None:
        print("  around:", tr[max(0, first - 2):first + 2], ref.trace[max(0, first - 2):first + 2])
