"""The synthetic LPs of bench.py and of the large-scale GPU tests are what their docstrings say (CPU tier; the f64 oracle is the
checker).  `multicommodity_lp` has the shape of the reference's KEN-* / PDS-* files (tests/netlib/problem_files)."""
import numpy as np

import rust_lp_amd  # noqa: F401
from oracle import relp_f64
from rust_lp_amd import MatrixData, synthetic


def _columns(d):
    cp, ri, va = d["col_ptr"], d["row_idx"], d["values"]
    return [(ri[cp[j]:cp[j + 1]], va[cp[j]:cp[j + 1]]) for j in range(d["n"])]


def test_multicommodity_structure_and_solve():
    v, e, k = 30, 120, 3
    d = synthetic.multicommodity_lp(v, e, k, 7)
    assert d["m"] == k * (v - 1) + e and d["n"] == k * e and d["nr_eq"] == k * (v - 1) and d["nr_le"] == e
    assert np.all(d["b"] >= 0) and np.all(d["c"] >= 1)
    cols = _columns(d)
    for j, (rows, vals) in enumerate(cols):
        assert len(rows) in (2, 3) and list(rows) == sorted(rows) and set(np.abs(vals)) == {1.0}
        assert rows[-1] == d["nr_eq"] + j % e and vals[-1] == 1.0                # the arc's capacity row
        assert all(r // (v - 1) == j // e for r in rows[:-1])                    # conservation rows of its own commodity
    # deterministic, and the whole incidence block when node 0 keeps its row
    again = synthetic.multicommodity_lp(v, e, k, 7)
    assert all(np.array_equal(d[q], again[q]) for q in ("col_ptr", "row_idx", "values", "b", "c"))
    full = synthetic.multicommodity_lp(v, e, k, 7, drop_one_node=False)
    assert full["m"] == k * v + e and all(len(r) == 3 for r, _ in _columns(full))
    # feasible and bounded by construction; the rank-deficient variant goes through the artificial-removal path to the same optimum
    a = relp_f64.OracleF64(MatrixData.from_sparse_dict(d))
    assert a.run() == "optimal"
    b = relp_f64.OracleF64(MatrixData.from_sparse_dict(full))
    assert b.run() == "optimal" and len(b.filtered_rows()) == k                  # one redundant row per commodity
    assert abs(a.objective - b.objective) < 1e-9 * abs(a.objective) and a.objective > 0


def test_sparse_lp_all_le_variant_starts_in_phase_two():
    d = synthetic.sparse_lp(60, 150, 11, frac_eq=0.0, frac_ge=0.0)
    assert d["nr_eq"] == 0 and d["nr_ge"] == 0 and d["nr_le"] == 60
    d["values"] = np.abs(d["values"]); d["b"] = np.abs(d["b"]) + 1.0; d["c"] = -d["c"]
    o = relp_f64.OracleF64(MatrixData.from_sparse_dict(d))
    assert o.run() == "optimal" and o.nr_artificial == 0 and o.objective < 0
    assert all(ph == 2 for ph, _, _, _ in o.trace)
