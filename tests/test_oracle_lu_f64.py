"""The f64 CPU oracle's SECOND back-end, `LUDecomposition` with its Forrest-Tomlin-style update file (oracle/relp_f64_lu.h) --
what the reference's binary runs (`src/bin/main.rs:52`: `Carry<_, LUDecomposition<_>>`) and what `bench.py` times beside the
LU engine as `cpu_baseline` (kind "port").

Pinned three ways:
  * the reference's own known answers for this back-end: `lower_upper/mod.rs:605-867` (the five `change_basis` cases incl. the
    Elble-Sahinidis 5 x 5: eta values, the rotated U, every column and row of the updated inverse), `mod.rs:488-603` (solves with
    the identity and an off-diagonal L), `decomposition/mod.rs:301-491` (factorisation cases, `wikipedia_example2`'s FTRAN answers);
  * the exact oracle (`oracle/relp_exact.py`, LUDecomposition over Fractions): identical pivot traces on the C2 parity shadows and
    on two-phase sparse LPs;
  * the first back-end in the same arithmetic (`BasisInverseRows`): identical traces and objectives on the reference's problem
    files (adlittle's exact trace through `test_mps_pipeline.py` transitively), which is the instrument VERDICT r3 asked for --
    a divergence of the LU GPU engine from the rows oracle can now be told apart from a divergence from its own arithmetic.

One f64 reading was needed (documented in relp_f64_lu.h): the factorisation drops an entry whose cancellation leaves a residue
below 1e-11 of the operands -- exact arithmetic drops exact zeros (decomposition/mod.rs:178); without it Netlib SHARE1B picks such
a residue as a Markowitz pivot and the inverse is wrong by 6 %."""
from fractions import Fraction as Fr

import numpy as np
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, synthetic
from oracle import relp_exact as ox
from oracle import relp_f64
from oracle.relp_f64 import LUF64


def dense(cols, m):
    out = np.zeros(m)
    for i, v in cols:
        out[i] = float(v)
    return out


def upper_columns(lu):
    u, _, _ = lu.factor("upper")
    return [[(i, u[i, j]) for i in range(lu.m) if u[i, j] != 0.0] for j in range(lu.m)]


def test_solves_with_identity_and_offdiagonal():
    """lower_upper/mod.rs:488-603."""
    ident = LUF64.identity(2)
    for col in ([], [(0, 1)], [(1, 1)], [(0, 1), (1, 1)]):
        np.testing.assert_array_equal(ident.generate_column(col), dense(col, 2))
    off = LUF64.from_triangles(2, [[(1, 1.0)]], [[(0, 1.0)], [(1, 1.0)]])
    np.testing.assert_array_equal(off.generate_column([]), [0, 0])
    np.testing.assert_array_equal(off.generate_column([(0, 1)]), [1, -1])
    np.testing.assert_array_equal(off.generate_column([(1, 1)]), [0, 1])


def test_change_basis_no_change_and_from_identity():
    """lower_upper/mod.rs:605-668."""
    lu = LUF64.identity(3)
    lu.generate_column([(1, 1)])
    assert lu.change_basis(1)
    (pivot, values), = lu.updates()
    assert pivot == 1 and not values.any()
    assert upper_columns(lu) == [[(0, 1.0)], [(1, 1.0)], [(2, 1.0)]]
    lu = LUF64.identity(2)
    np.testing.assert_array_equal(lu.generate_column([(0, 1), (1, 1)]), [1, 1])
    assert lu.change_basis(0)
    assert upper_columns(lu) == [[(0, 1.0)], [(0, 1.0), (1, 1.0)]]
    (pivot, values), = lu.updates()
    assert pivot == 0 and not values.any()


def test_change_basis_5x5_no_r():
    """lower_upper/mod.rs:670-701."""
    lu = LUF64.identity(5)
    lu.generate_column([(0, 2), (1, 3), (2, 5), (3, 7)])
    assert lu.change_basis(1)
    assert upper_columns(lu) == [[(0, 1.0)], [(1, 1.0)], [(2, 1.0)], [(3, 1.0)], [(0, 2.0), (1, 5.0), (2, 7.0), (4, 3.0)]]
    (pivot, values), = lu.updates()
    assert pivot == 1 and not values.any()


def test_change_basis_4x4():
    """lower_upper/mod.rs:703-771: eta (3: 5/6), the rotated U, every column and row of the updated inverse."""
    m = 4
    lu = LUF64.from_triangles(m, [[]] * m, [[(0, 1)], [(1, 1)], [(2, 4)], [(1, 5), (3, 6)]])
    # the spike is handed over as the reference's test does: FTRAN of the entering column (2, 3, 4 on rows 1..3) through L = I
    # and no earlier updates leaves it as it is
    lu.generate_column([(1, 2), (2, 3), (3, 4)])
    assert lu.change_basis(1)
    (pivot, values), = lu.updates()
    assert pivot == 1
    np.testing.assert_allclose(values, [0, 0, 0, 5 / 6], atol=1e-15)
    got = upper_columns(lu)
    want = [[(0, 1)], [(1, 4)], [(2, 6)], [(1, 3), (2, 4), (3, -8 / 6)]]
    for g, w in zip(got, want):
        assert [i for i, _ in g] == [i for i, _ in w]
        np.testing.assert_allclose([v for _, v in g], [v for _, v in w], rtol=1e-14)
    cols = {0: [1, 0, 0, 0], 1: [0, -3 / 4, 9 / 16, 1 / 2], 2: [0, 0, 1 / 4, 0], 3: [0, 5 / 8, -15 / 32, -1 / 4]}
    for j, want_col in cols.items():
        np.testing.assert_allclose(lu.generate_column([(j, 1)]), want_col, atol=1e-14)
    rows = {0: [1, 0, 0, 0], 1: [0, -3 / 4, 0, 5 / 8], 2: [0, 9 / 16, 1 / 4, -15 / 32], 3: [0, 1 / 2, 0, -1 / 4]}
    for i, want_row in rows.items():
        np.testing.assert_allclose(lu.basis_inverse_row(i), want_row, atol=1e-14)


def test_change_basis_elble_sahinidis_5x5():
    """lower_upper/mod.rs:773-867."""
    m = 5
    lu = LUF64.from_triangles(m, [[]] * m, [[(0, 11)], [(0, 12), (1, 22)], [(0, 13), (1, 23), (2, 33)],
                                            [(0, 14), (1, 24), (2, 34), (3, 44)], [(0, 15), (1, 25), (2, 35), (3, 45), (4, 55)]])
    # the reference passes the spike (12, 22, 32, 42) directly; it is the FTRAN-through-L of itself (L = I, no updates yet)
    lu.generate_column([(0, 12), (1, 22), (2, 32), (3, 42)])
    assert lu.change_basis(1)
    (pivot, values), = lu.updates()
    assert pivot == 1
    np.testing.assert_allclose(values, [0, 0, 23 / 33, (24 * 33 - 34 * 23) / (33 * 44), 43 / 7986], rtol=1e-14, atol=1e-16)
    want_u = [[(0, 11)], [(0, 13), (1, 33)], [(0, 14), (1, 34), (2, 44)], [(0, 15), (1, 35), (2, 45), (3, 55)],
              [(0, 12), (1, 32), (2, 42), (4, -215 / 363)]]
    for g, w in zip(upper_columns(lu), want_u):
        assert [i for i, _ in g] == [i for i, _ in w]
        np.testing.assert_allclose([v for _, v in g], [v for _, v in w], rtol=1e-13)
    cols = {0: [1 / 11, 0, 0, 0, 0], 1: [-2 / 11, -363 / 215, -1 / 43, 693 / 430, 0], 2: [1 / 11, 253 / 215, 2 / 43, -483 / 430, 0],
            3: [0, 1 / 86, -1 / 43, 1 / 86, 0], 4: [0, 1 / 110, 0, -3 / 110, 1 / 55]}
    for j, want_col in cols.items():
        np.testing.assert_allclose(lu.generate_column([(j, 1)]), want_col, atol=1e-14)
    np.testing.assert_allclose(lu.generate_column([(0, 1), (1, 1)]), [-1 / 11, -363 / 215, -1 / 43, 693 / 430, 0], atol=1e-14)
    rows = {0: [1 / 11, -2 / 11, 1 / 11, 0, 0], 1: [0, -363 / 215, 253 / 215, 1 / 86, 1 / 110], 2: [0, -1 / 43, 2 / 43, -1 / 43, 0],
            3: [0, 693 / 430, -483 / 430, 1 / 86, -3 / 110], 4: [0, 0, 0, 0, 1 / 55]}
    for i, want_row in rows.items():
        np.testing.assert_allclose(lu.basis_inverse_row(i), want_row, atol=1e-14)


REFERENCE_MATRICES = {       # decomposition/mod.rs:315-420, by columns
    "identity_2": [[(0, 1)], [(1, 1)]], "identity_3": [[(0, 1)], [(1, 1)], [(2, 1)]],
    "offdiagonal_upper": [[(0, 1)], [(0, 1), (1, 1)]], "offdiagonal_lower": [[(0, 1), (1, 1)], [(1, 1)]],
    "offdiagonal_both": [[(0, 1), (1, 1)], [(0, 1)]], "wikipedia_example": [[(0, 4), (1, 6)], [(0, 3), (1, 3)]],
    "wikipedia_example2": [[(0, -1), (1, 1)], [(0, 1.5), (1, -1)]],
}


@pytest.mark.parametrize("name", sorted(REFERENCE_MATRICES))
def test_reference_factorisation_cases(name):
    cols = REFERENCE_MATRICES[name]
    m = len(cols)
    lu = LUF64.invert(cols)
    assert lu is not None
    a = np.zeros((m, m))
    for j, c in enumerate(cols):
        for i, v in c:
            a[i, j] = v
    inv = np.linalg.inv(a)
    for j in range(m):
        np.testing.assert_allclose(lu.generate_column([(j, 1)]), inv[:, j], atol=1e-14)
        np.testing.assert_allclose(lu.basis_inverse_row(j), inv[j, :], atol=1e-14)
    if name == "wikipedia_example2":                         # decomposition/mod.rs:470-489
        np.testing.assert_allclose(lu.generate_column([(0, 1)]), [2, 2], atol=1e-14)
        np.testing.assert_allclose(lu.generate_column([(1, 1)]), [3, 2], atol=1e-14)
    # P B Q = L U with a unit lower and an upper triangle
    lo, rf, cf = lu.factor("lower")
    up, _, _ = lu.factor("upper")
    pbq = np.zeros((m, m))
    for i in range(m):
        for j in range(m):
            pbq[rf[i], cf[j]] = a[i, j]
    np.testing.assert_allclose((lo + np.eye(m)) @ up, pbq, atol=1e-14)
    assert np.allclose(np.triu(lo), 0) and np.allclose(np.tril(up, -1), 0)


def test_a_singular_matrix_is_reported():
    assert LUF64.invert([[(0, 1), (1, 2)], [(0, 2), (1, 4)]]) is None


def test_random_replacement_sequences_against_numpy():
    """Column replacements far beyond the reference's ten pending updates: after every change_basis the columns and rows of the
    inverse equal numpy's for the matrix as it is then."""
    rng = np.random.default_rng(11)
    for m in (3, 12, 40):
        a = np.eye(m) * rng.integers(1, 4, m) + (rng.random((m, m)) < 0.15) * rng.integers(-3, 4, (m, m))
        while abs(np.linalg.det(a)) < 1e-6:
            a += np.eye(m)
        lu = LUF64.invert([[(i, a[i, j]) for i in range(m) if a[i, j] != 0] for j in range(m)])
        for step in range(30):
            pos = int(rng.integers(0, m))
            col = np.where(rng.random(m) < 0.3, rng.integers(-3, 4, m), 0).astype(float)
            col[pos] += 5.0
            trial = a.copy()
            trial[:, pos] = col
            if abs(np.linalg.det(trial)) < 1e-3 or np.linalg.cond(trial) > 1e8:
                continue
            lu.generate_column([(i, col[i]) for i in range(m) if col[i] != 0])
            assert lu.change_basis(pos)
            a = trial
            inv = np.linalg.inv(a)
            scale = max(1.0, np.abs(inv).max())
            for j in rng.choice(m, min(m, 5), replace=False):
                np.testing.assert_allclose(lu.generate_column([(int(j), 1)]), inv[:, j], atol=1e-9 * scale)
                np.testing.assert_allclose(lu.basis_inverse_row(int(j)), inv[j, :], atol=1e-9 * scale)


def to_exact(md):
    cols = []
    for j in range(md.nr_normal):
        s, t = md.col_ptr[j], md.col_ptr[j + 1]
        cols.append([(int(md.row_idx[p]), Fr(md.values[p])) for p in range(s, t)])
    ub = [None if not np.isfinite(u) else Fr(u) for u in md.upper_bound]
    return ox.MatrixData(cols, [Fr(v) for v in md.b], [Fr(v) for v in md.ranges], md.nr_eq, md.nr_range, md.nr_le,
                         md.nr_ge, [Fr(c) for c in md.cost], ub)


@pytest.mark.parametrize("m,n,seed", [(8, 8, 1), (32, 48, 7)])
def test_dense_shadow_trace_equals_the_exact_lu_back_end(m, n, seed):
    lp = synthetic.dense_lp(m, n, seed)
    ref = relp_f64.OracleF64(MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]).ensure_csc(), basis_inverse=1)
    assert ref.run() == "optimal"
    cols, b, c = synthetic.dense_lp_exact(m, n, seed)
    tr = []
    out = ox.solve_relaxation(ox.MatrixData(cols, b, [], 0, 0, m, 0, c, [None] * n), ox.LUDecomposition, trace=tr.append)
    assert out["status"] == "optimal"
    assert ref.trace == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
    assert abs(ref.objective - float(out["objective"])) <= 1e-9 * abs(float(out["objective"]))
    assert ref.lu_stats()["refactorisations"] >= len(tr) // 11 - 1


@pytest.mark.parametrize("m,n,seed", [(12, 10, 1), (20, 30, 5), (40, 25, 4)])
def test_sparse_two_phase_trace_equals_the_exact_lu_back_end(m, n, seed):
    md = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed))
    ref = relp_f64.OracleF64(md, basis_inverse=1)
    status = ref.run()
    tr = []
    out = ox.solve_relaxation(to_exact(md), ox.LUDecomposition, trace=tr.append)
    assert status == out["status"] == "optimal"
    assert ref.trace == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
    assert abs(ref.objective - float(out["objective"])) <= 1e-9 * max(1.0, abs(float(out["objective"])))
    np.testing.assert_allclose(ref.b(), [float(v) for v in out["tableau"].im.b], rtol=1e-9, atol=1e-9)


FILES = [("burkardt/adlittle.mps", False), ("netlib/AFIRO.SIF", True), ("netlib/SC50A.SIF", True), ("netlib/SC205.SIF", True),
         ("netlib/SHARE1B.SIF", True), ("netlib/SHARE2B.SIF", True), ("netlib/LOTFI.SIF", True), ("netlib/BOEING2.SIF", True),
         ("netlib/BORE3D.SIF", True), ("netlib/SCAGR7.SIF", True), ("netlib/STOCFOR1.SIF", True), ("netlib/VTP-BASE.SIF", True)]


@pytest.mark.parametrize("path,fixed", FILES)
def test_both_back_ends_walk_the_same_pivots_on_the_reference_files(path, fixed):
    """Every refactorisation at the reference's cadence (more than 10 updates pending); BORE3D removes rows, BOEING2 carries an
    artificial variable with a wrapped index into phase 2 (the LU back-end factorises its unit column; the reference's release
    build would index its provider out of range at the next refactorisation)."""
    from lp_files import load
    gf, ex, md, emd = load(path, fixed=fixed)
    rows = relp_f64.OracleF64(md)
    lu = relp_f64.OracleF64(md, basis_inverse=1)
    assert rows.run() == "optimal" and lu.run() == "optimal"
    assert lu.trace == rows.trace
    assert abs(lu.objective - rows.objective) <= 1e-9 * max(1.0, abs(rows.objective))
    assert lu.filtered_rows() == rows.filtered_rows()
    assert lu.lu_stats()["refactorisations"] >= len(lu.trace) // 12
    np.testing.assert_allclose(lu.b(), rows.b(), rtol=1e-8, atol=1e-8)
