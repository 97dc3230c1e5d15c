"""One LU-engine solve for rocprofv3 --kernel-trace --stats: Netlib 25FV47 (config C3) by default, or
`lu_profile.py synth M N SEED` for a synthetic sparse LP."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic

if len(sys.argv) > 1 and sys.argv[1] == "synth":
    m, n, seed = (int(v) for v in sys.argv[2:5])
    md, kw = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed)), {}
else:
    from rust_lp_amd import general_form, mps
    gf = general_form.GeneralForm.from_mps(mps.import_file(os.path.join(ROOT, "tests", "golden", "mps", "netlib", "25FV47.SIF"), True))
    md = gf.to_matrix_data(gf.derive_matrix_data_exact())
    kw = {}
t = engine.Tableau(md, engine=engine.ENGINE_LU, **kw)
print(engine.OUTCOME_NAMES[t.solve_relaxation()], t.iterations(), t.lu_stats())
