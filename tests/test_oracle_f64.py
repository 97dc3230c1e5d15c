"""The C (f64) oracle against the exact oracle: identical pivot traces, values within 1e-9 relative.
These pin the f64 restatement that the GPU parity tests use as their checker at larger sizes."""
from fractions import Fraction as Fr

import numpy as np
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, synthetic
from oracle import relp_exact as ox
from oracle import relp_f64


def to_exact(md):
    cols = []
    for j in range(md.nr_normal):
        s, t = md.col_ptr[j], md.col_ptr[j + 1]
        cols.append([(int(md.row_idx[p]), Fr(md.values[p])) for p in range(s, t)])
    ub = [None if not np.isfinite(u) else Fr(u) for u in md.upper_bound]
    return ox.MatrixData(cols, [Fr(v) for v in md.b], [Fr(v) for v in md.ranges], md.nr_eq, md.nr_range, md.nr_le,
                         md.nr_ge, [Fr(c) for c in md.cost], ub)


@pytest.mark.parametrize("m,n,seed", [(8, 8, 1), (32, 48, 7), (64, 40, 3)])
def test_dense_f64_trace_equals_exact(m, n, seed):
    lp = synthetic.dense_lp(m, n, seed)
    ref = relp_f64.OracleF64(MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]).ensure_csc())
    assert ref.run() == "optimal"
    cols, b, c = synthetic.dense_lp_exact(m, n, seed)
    tr = []
    out = ox.solve_relaxation(ox.MatrixData(cols, b, [], 0, 0, m, 0, c, [None] * n), trace=tr.append)
    assert out["status"] == "optimal"
    assert ref.trace == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
    assert abs(ref.objective - float(out["objective"])) <= 1e-9 * abs(float(out["objective"]))


@pytest.mark.parametrize("m,n,seed", [(12, 10, 1), (20, 30, 5), (40, 25, 4), (60, 90, 2)])
def test_sparse_two_phase_f64_trace_equals_exact(m, n, seed):
    md = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed))
    ref = relp_f64.OracleF64(md)
    status = ref.run()
    tr = []
    out = ox.solve_relaxation(to_exact(md), trace=tr.append, check=(m <= 20))
    assert status == out["status"] == "optimal"
    assert ref.trace == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
    assert any(e["phase"] == 1 for e in tr) and any(e["phase"] == 2 for e in tr)
    assert abs(ref.objective - float(out["objective"])) <= 1e-9 * max(1.0, abs(float(out["objective"])))
    tab = out["tableau"]
    np.testing.assert_allclose(ref.b(), [float(v) for v in tab.im.b], rtol=1e-9, atol=1e-9)
    assert ref.basis().tolist() == tab.im.basis_indices


def test_reference_pins_through_f64_oracle():
    """problem_2 (src/tests/problem_2.rs) with FirstProfitable in both phases."""
    cons = np.asfortranarray([[3.0, 2, 1, 0, 0], [5, 1, 1, 1, 0], [2, 5, 1, 0, 1]])
    md = MatrixData(nr_normal=5, nr_eq=3, nr_range=0, nr_le=0, nr_ge=0, b=np.array([1.0, 3, 4]), cost=np.ones(5),
                    upper_bound=np.full(5, np.inf), dense=cons).ensure_csc()
    ref = relp_f64.OracleF64(md, phase_one_rule=0, phase_two_rule=0)
    assert ref.run(through_phases=False) == "phase_one_done"
    assert abs(ref.objective - 4.5) < 1e-12
    np.testing.assert_allclose(ref.minus_pi(), [2.5, -1, -1], atol=1e-12)
    np.testing.assert_allclose(ref.b(), [0.5, 2.5, 1.5], atol=1e-12)
    assert ref.basis().tolist() == [1, 3, 4]
    assert ref.run() == "optimal"
    assert abs(ref.objective - 4.5) < 1e-12


def test_generator_is_deterministic_and_rational():
    a = synthetic.dense_lp(16, 24, 5)
    b = synthetic.dense_lp(16, 24, 5)
    assert np.array_equal(a["A"], b["A"]) and np.array_equal(a["b"], b["b"])
    nums = synthetic.dense_numerators(16, 24, 5)
    assert nums["A_num"].min() >= 1 and nums["A_num"].max() <= 999
    assert (a["b"] > 0).all() and (a["c"] < 0).all()


def test_f64_oracle_equals_exact_on_random_lps_with_every_row_kind_and_outcome():
    """60 small LPs from `synthetic.mixed_lp` (ranges, bounds, negative costs, contradictory rows): status and
    pivot trace of the C oracle equal the exact oracle's, including the rows removed at the phase switch and the
    wrapped index of a surviving artificial (`usize_sub`)."""
    rng = np.random.default_rng(11)
    seen = {"optimal": 0, "unbounded": 0, "infeasible": 0}
    removed = 0
    for case in range(60):
        m, n = int(rng.integers(6, 28)), int(rng.integers(4, 36))
        md = MatrixData.from_sparse_dict(synthetic.mixed_lp(
            m, n, 300 + case, nnz_per_col=int(rng.integers(2, 5)), frac_eq=float(rng.uniform(0, 0.3)),
            frac_range=float(rng.uniform(0, 0.3)), frac_ge=float(rng.uniform(0, 0.3)), frac_bounded=float(rng.uniform(0, 0.6)),
            frac_negative_cost=float(rng.choice([0.0, 0.0, 0.1, 0.3])), infeasible=bool(rng.random() < 0.15)))
        ref = relp_f64.OracleF64(md)
        status = ref.run(100000)
        removed += bool(ref.filtered_rows())
        tr = []
        out = ox.solve_relaxation(to_exact(md), trace=tr.append)
        assert out["status"] == status, case
        seen[status] += 1
        exact = [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
        squeezed = [(p, q, r, lv if lv < (1 << 62) else None) for (p, q, r, lv) in exact]
        got = [(p, q, r, lv if lv < (1 << 30) else None) for (p, q, r, lv) in ref.trace]
        assert got == squeezed, case
        if status == "optimal":
            assert abs(ref.objective - float(out["objective"])) <= 1e-9 * max(1.0, abs(float(out["objective"]))), case
    assert min(seen.values()) >= 3 and removed >= 2, (seen, removed)


def test_f64_oracle_equals_exact_on_the_edge_cases():
    from edge_lps import CASES
    for name, problem in CASES.items():
        md = problem.ensure_csc()
        ref = relp_f64.OracleF64(md)
        status = ref.run(10000)
        tr = []
        out = ox.solve_relaxation(to_exact(md), trace=tr.append)
        assert out["status"] == status, name
        assert ref.trace == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr], name
