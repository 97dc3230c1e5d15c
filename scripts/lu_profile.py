"""One LU-engine solve of a sparse problem, for rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic

m, n, seed = (int(v) for v in (sys.argv[1:4] or (400, 800, 77)))
md = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed))
t = engine.Tableau(md, engine=engine.ENGINE_LU)
print(engine.OUTCOME_NAMES[t.solve_relaxation()], t.iterations(), t.lu_stats())
