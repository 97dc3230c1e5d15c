"""One LU-engine solve for rocprofv3 --kernel-trace --stats: Netlib 25FV47 (config C3) by default, or
`lu_profile.py synth M N SEED` for a synthetic sparse LP."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic

if len(sys.argv) > 1 and sys.argv[1] == "synth":
    m, n, seed = (int(v) for v in sys.argv[2:5])
    md, kw = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed)), {}
else:
    from lp_files import load
    gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
    kw = dict(tol_pivot=1e-5, tol_cost=1e-7)
t = engine.Tableau(md, engine=engine.ENGINE_LU, **kw)
print(engine.OUTCOME_NAMES[t.solve_relaxation()], t.iterations(), t.lu_stats())
