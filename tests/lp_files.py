"""Helpers for the file-driven tests: MPS/SIF fixture -> GeneralForm -> MatrixData (exact and f64).
The fixtures under tests/golden/mps/ are the data files the reference's own tests hold
(/root/reference/tests/{burkardt,cook,unicamp,netlib,miplib}/problem_files)."""
import os

from rust_lp_amd import general_form, mps
from oracle import relp_exact as ox

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mps")


def load(rel_path, fixed=False):
    """Returns (general_form, exact_tuple, f64 MatrixData, exact oracle MatrixData)."""
    m = mps.import_file(os.path.join(GOLDEN, rel_path), fixed)
    gf = general_form.GeneralForm.from_mps(m)
    ex = gf.derive_matrix_data_exact()
    cols, b, ranges, (ne, nr, nl, ng), costs, ubs = ex
    md = gf.to_matrix_data(ex)
    emd = ox.MatrixData(cols, b, ranges, ne, nr, nl, ng, costs, ubs)
    return gf, ex, md, emd


def exact_solve(gf, emd, BI=ox.BasisInverseRows, trace=None):
    out = ox.solve_relaxation(emd, BI, trace=trace)
    if out["status"] != "optimal":
        return out["status"], None, None
    provider = out["tableau"].kind.provider
    obj, sol = gf.compute_full_solution(dict(provider.reconstruct_solution(out["bfs"])))
    return "optimal", obj, sol
