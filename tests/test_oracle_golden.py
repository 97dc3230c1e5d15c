"""Pin the exact oracle (oracle/relp_exact.py) against every known-answer test the reference
holds for the pivot path.  Each test names the reference test it mirrors
(paths relative to /root/reference/).  The expected values below are DATA taken from those
tests (inputs and expected outputs), re-typed as Python literals.
"""
from fractions import Fraction as Fr

import pytest

from oracle import relp_exact as ox
from oracle.relp_exact import (BasisInverseRows, Carry, EtaFile, FirstProfitable, FullPermutation,
                               LUDecomposition, MatrixData, NonArtificial, RotateToBack,
                               SteepestDescent, SwapPermutation, Tableau, ColumnAndSpike)


def R(a, b=1):
    return Fr(a, b)


def sv(pairs):
    return [(i, Fr(v)) for i, v in pairs]


def dense_rows_to_columns(rows, ncols):
    """``ColumnMajor::from_test_data``: dense row list -> sparse columns (zeros dropped)."""
    cols = [[] for _ in range(ncols)]
    for i, row in enumerate(rows):
        for j, v in enumerate(row):
            if v != 0:
                cols[j].append((i, Fr(v)))
    return cols


def dense_to_sparse(values):
    return [(i, Fr(v)) for i, v in enumerate(values) if v != 0]


# ------------------------------------------------------------------------------------------
# Fixtures: src/tests/problem_1.rs and src/tests/problem_2.rs
# ------------------------------------------------------------------------------------------
def problem_2_matrix_data():
    """src/tests/problem_2.rs:73-114 (Papadimitriou-Steiglitz 3x5, all equality rows)."""
    cons = dense_rows_to_columns([[3, 2, 1, 0, 0], [5, 1, 1, 1, 0], [2, 5, 1, 0, 1]], 5)
    return MatrixData(cons, [1, 3, 4], [], 3, 0, 0, 0, [1] * 5, [None] * 5)


def problem_1_matrix_data():
    """src/tests/problem_1.rs:313-365 (Wikipedia TESTPROB after standardisation)."""
    cons = dense_rows_to_columns([[0, -1, 1], [1, 0, 1]], 3)
    return MatrixData(cons, [6, 10], [], 1, 0, 0, 1, [1, 4, 9], [4, 2, None])


def problem_2_tableau():
    """tableau/mod.rs:378-403 helper ``tableau``."""
    md = problem_2_matrix_data()
    bi = BasisInverseRows([dense_to_sparse([1, 0, 0]), dense_to_sparse([-1, 1, 0]), dense_to_sparse([-1, 0, 1])])
    carry = Carry(-6, [1, -1, -1], [1, 2, 3], [2, 3, 4], bi)
    return Tableau(carry, [2, 3, 4], NonArtificial(md))


# ------------------------------------------------------------------------------------------
# src/tests/problem_2.rs::conversion_pipeline
# ------------------------------------------------------------------------------------------
def test_problem_2_pipeline():
    md = problem_2_matrix_data()
    t1 = Tableau.new_partially_artificial(BasisInverseRows, md)
    # artificial_tableau_form (:116-139)
    assert t1.im.minus_objective == -8
    assert t1.im.minus_pi == [-1, -1, -1]
    assert t1.im.b == [1, 3, 4]
    assert t1.im.basis_indices == [0, 1, 2]
    assert t1.kind.column_to_row == [0, 1, 2]
    assert t1.im.basis_inverse == BasisInverseRows.identity(3)

    res = ox.phase_one_primal(t1, FirstProfitable(), check=True)
    assert res[0] == "feasible" and res[1] == []
    t2 = Tableau.from_artificial(res[3], res[2], res[4], md)
    # tableau_form (:141-174)
    assert t2.im.minus_objective == R(-9, 2)
    assert t2.im.minus_pi == [R(5, 2), -1, -1]
    assert t2.im.b == [R(1, 2), R(5, 2), R(3, 2)]
    assert t2.im.basis_indices == [1, 3, 4]
    assert t2.im.basis_inverse.rows_ == [
        dense_to_sparse([R(1, 2), 0, 0]), dense_to_sparse([R(-1, 2), 1, 0]), dense_to_sparse([R(-5, 2), 0, 1])]
    assert t2.basis_columns == {1, 3, 4}

    out = ox.phase_two_primal(t2, FirstProfitable(), check=True)
    assert out == ("optimal", [(1, R(1, 2)), (3, R(5, 2)), (4, R(3, 2))])


# ------------------------------------------------------------------------------------------
# src/tests/problem_1.rs::conversion_pipeline (from the MatrixData stage onwards)
# ------------------------------------------------------------------------------------------
def test_problem_1_pipeline():
    md = problem_1_matrix_data()
    assert md.nr_rows() == 4 and md.nr_columns() == 6
    t1 = Tableau.new_partially_artificial(BasisInverseRows, md)
    # artificial_tableau_form (:376-401)
    assert t1.im.minus_objective == -16
    assert t1.im.minus_pi == [-1, -1, 0, 0]
    assert t1.im.b == [6, 10, 4, 2]
    assert t1.im.basis_indices == [0, 1, 2 + 4, 2 + 5]
    assert t1.kind.column_to_row == [0, 1]

    res = ox.phase_one_primal(t1, FirstProfitable(), check=True)
    assert res[0] == "feasible" and res[1] == []
    t2 = Tableau.from_artificial(res[3], res[2], res[4], md)
    # tableau_form (:403-431)
    assert t2.im.minus_objective == -58
    assert t2.im.minus_pi == [4, -13, 12, 0]
    assert t2.im.b == [6, 0, 4, 2]
    assert t2.im.basis_indices == [2, 1, 0, 5]
    assert t2.im.basis_inverse.rows_ == [
        dense_to_sparse([0, 1, -1, 0]), dense_to_sparse([-1, 1, -1, 0]),
        dense_to_sparse([0, 0, 1, 0]), dense_to_sparse([1, -1, 1, 1])]

    out = ox.phase_two_primal(t2, FirstProfitable(), check=True)
    assert out == ("optimal", [(0, R(4)), (2, R(6)), (5, R(2))])
    # objective 54 = sum c_j x_j + fixed cost (-4 from the shift of YTWO): :100-106
    x = md.reconstruct_solution(out[1])
    assert x == [(0, R(4)), (2, R(6))]
    # standardised costs (1, 4, 9); YTWO was shifted by 1 (fixed cost -(1*4)) -- :306
    assert sum(md.costs[j] * v for j, v in x) == 58
    assert t2.objective_function_value() == 58


# ------------------------------------------------------------------------------------------
# tableau/mod.rs tests (:406-519) and strategy/pivot_rule.rs tests (:137-167)
# ------------------------------------------------------------------------------------------
def test_tableau_cost():
    md = problem_2_matrix_data()
    art = Tableau.new_partially_artificial(BasisInverseRows, md)
    assert art.objective_function_value() == 8
    assert problem_2_tableau().objective_function_value() == 6


def test_tableau_relative_cost():
    md = problem_2_matrix_data()
    art = Tableau.new_partially_artificial(BasisInverseRows, md)
    assert art.relative_cost(0) == 0
    assert art.relative_cost(art.kind.nr_artificial_variables() + 0) == -10
    t = problem_2_tableau()
    assert t.relative_cost(0) == -3
    assert t.relative_cost(1) == -3
    assert t.relative_cost(2) == 0


def test_tableau_generate_column():
    md = problem_2_matrix_data()
    art = Tableau.new_partially_artificial(BasisInverseRows, md)
    j = art.kind.nr_artificial_variables() + 0
    assert art.generate_column(j).column == sv([(0, 3), (1, 5), (2, 2)])
    assert art.relative_cost(j) == -10
    t = problem_2_tableau()
    assert t.generate_column(0).column == sv([(0, 3), (1, 2), (2, -1)])
    assert t.relative_cost(0) == -3


def test_tableau_bring_into_basis():
    md = problem_2_matrix_data()
    art = Tableau.new_partially_artificial(BasisInverseRows, md)
    j = art.kind.nr_artificial_variables() + 0
    col = art.generate_column(j)
    row = art.select_primal_pivot_row(col.column)
    cost = art.relative_cost(j)
    art.bring_into_basis(j, row, col, cost)
    assert art.is_in_basis(j) and not art.is_in_basis(0)
    assert art.objective_function_value() == R(14, 3)

    t = problem_2_tableau()
    col = t.generate_column(1)
    row = t.select_primal_pivot_row(col.column)
    cost = t.relative_cost(1)
    t.bring_into_basis(1, row, col, cost)
    assert t.is_in_basis(1)
    assert t.objective_function_value() == R(9, 2)


def test_tableau_create_bfs_no_candidate():
    """tableau/mod.rs:483-518 ``create_tableau``."""
    md = problem_2_matrix_data()
    m = 3
    bi = BasisInverseRows([dense_to_sparse([1, 0, 0]), dense_to_sparse([-1, 1, 0]), dense_to_sparse([-1, 0, 1])])
    carry = Carry(0, [1, 1, 1], [1, 2, 3], [m + 2, m + 3, m + 4], bi)
    t = Tableau(carry, [m + 2, m + 3, m + 4], NonArtificial(md))
    assert FirstProfitable().select_primal_pivot_column(t) is None


def test_pivot_rule_find_profitable_column():
    md = problem_2_matrix_data()
    art = Tableau.new_partially_artificial(BasisInverseRows, md)
    sel = FirstProfitable().select_primal_pivot_column(art)
    assert sel is not None and sel[0] == 3
    # tableau_form of problem_2 is already optimal for FirstProfitable
    bi = BasisInverseRows([dense_to_sparse([R(1, 2), 0, 0]), dense_to_sparse([R(-1, 2), 1, 0]),
                           dense_to_sparse([R(-5, 2), 0, 1])])
    carry = Carry(R(-9, 2), [R(5, 2), -1, -1], [R(1, 2), R(5, 2), R(3, 2)], [1, 3, 4], bi)
    t = Tableau(carry, [1, 3, 4], NonArtificial(md))
    assert FirstProfitable().select_primal_pivot_column(t) is None


def test_pivot_rule_find_pivot_row():
    md = problem_2_matrix_data()
    art = Tableau.new_partially_artificial(BasisInverseRows, md)
    assert art.select_primal_pivot_row(sv([(0, 3), (1, 5), (2, 2)])) == 0
    assert art.select_primal_pivot_row(sv([(0, 2), (1, 1), (2, 5)])) == 0
    bi = BasisInverseRows([dense_to_sparse([R(1, 2), 0, 0]), dense_to_sparse([R(-1, 2), 1, 0]),
                           dense_to_sparse([R(-5, 2), 0, 1])])
    carry = Carry(R(-9, 2), [R(5, 2), -1, -1], [R(1, 2), R(5, 2), R(3, 2)], [1, 3, 4], bi)
    t = Tableau(carry, [1, 3, 4], NonArtificial(md))
    assert t.select_primal_pivot_row(sv([(0, 3), (1, 2), (2, -1)])) == 0
    assert t.select_primal_pivot_row(sv([(0, 2), (1, -1), (2, 3)])) == 0


# ------------------------------------------------------------------------------------------
# two_phase/mod.rs tests (:134-210)
# ------------------------------------------------------------------------------------------
def test_two_phase_simplex_from_tableau_form():
    md = problem_2_matrix_data()
    bi = BasisInverseRows([dense_to_sparse([R(1, 2), 0, 0]), dense_to_sparse([R(-1, 2), 1, 0]),
                           dense_to_sparse([R(-5, 2), 0, 1])])
    carry = Carry(R(-9, 2), [R(5, 2), -1, -1], [R(1, 2), R(5, 2), R(3, 2)], [1, 3, 4], bi)
    t = Tableau(carry, [1, 3, 4], NonArtificial(md))
    out = ox.phase_two_primal(t, FirstProfitable())
    assert out[0] == "optimal"
    assert t.objective_function_value() == R(9, 2)


@pytest.mark.parametrize("BI", [BasisInverseRows, LUDecomposition])
def test_solve_matrix(BI):
    """two_phase/mod.rs:147-161 (``Carry<S, LUDecomposition<S>>``; also run with rows)."""
    md = problem_2_matrix_data()
    out = ox.solve_relaxation(md, BI, check=True)
    assert out["status"] == "optimal"
    assert out["bfs"] == [(1, R(1, 2)), (3, R(5, 2)), (4, R(3, 2))]


@pytest.mark.parametrize("BI", [BasisInverseRows, LUDecomposition])
def test_solve_relaxation_1(BI):
    """two_phase/mod.rs:163-210."""
    cons = dense_rows_to_columns([[1, 0], [1, 1]], 2)
    md = MatrixData(cons, [R(3, 2), R(5, 2)], [], 0, 0, 2, 0, [-2, -1], [None, None])
    out = ox.solve_relaxation(md, BI, check=True)
    assert out["status"] == "optimal"
    assert out["bfs"] == [(0, R(3, 2)), (1, R(1))]


# ------------------------------------------------------------------------------------------
# carry/basis_inverse_rows.rs tests (:248-292)
# ------------------------------------------------------------------------------------------
def test_basis_inverse_rows_invert_identity():
    cols = [[(0, R(1))], [(1, R(1))]]
    assert BasisInverseRows.invert(cols) == BasisInverseRows.identity(2)


# ------------------------------------------------------------------------------------------
# lower_upper/eta_file.rs tests (:158-260)
# ------------------------------------------------------------------------------------------
def test_eta_empty():
    for (pivot, n) in [(0, 1), (0, 2), (1, 2)]:
        eta = EtaFile([], pivot, n)
        v = sv([(0, 1)])
        eta.apply_left(v)
        assert v == sv([(0, 1)])
        eta.apply_right(v)
        assert v == sv([(0, 1)])


def test_eta_single_value_2():
    eta = EtaFile(sv([(1, 1)]), 0, 2)
    v = sv([(0, 13), (1, 17)])
    eta.apply_right(v)
    assert v == sv([(0, 13 - 17), (1, 17)])
    v = sv([(0, 13), (1, 17)])
    eta.apply_left(v)
    assert v == sv([(0, 13), (1, 17 - 13)])
    v = []
    eta.apply_right(v)
    assert v == []
    eta.apply_left(v)
    assert v == []


def test_eta_two_values_3():
    eta = EtaFile(sv([(1, 5), (2, 7)]), 0, 3)
    v = sv([(0, 13), (1, 17), (2, 19)])
    eta.apply_right(v)
    assert v == sv([(0, 13 - 5 * 17 - 7 * 19), (1, 17), (2, 19)])
    v = sv([(0, 13), (1, 17), (2, 19)])
    eta.apply_left(v)
    assert v == sv([(0, 13), (1, -5 * 13 + 17), (2, -7 * 13 + 19)])


def test_eta_one_value_3():
    eta = EtaFile(sv([(1, 5)]), 0, 3)
    v = sv([(0, 13), (1, 17), (2, 19)])
    eta.apply_right(v)
    assert v == sv([(0, 13 - 5 * 17), (1, 17), (2, 19)])
    v = sv([(0, 13), (1, 17), (2, 19)])
    eta.apply_left(v)
    assert v == sv([(0, 13), (1, -5 * 13 + 17), (2, 19)])
    eta = EtaFile(sv([(2, 5)]), 0, 3)
    v = sv([(0, 13), (1, 17), (2, 19)])
    eta.apply_right(v)
    assert v == sv([(0, 13 - 5 * 19), (1, 17), (2, 19)])
    v = sv([(0, 13), (1, 17), (2, 19)])
    eta.apply_left(v)
    assert v == sv([(0, 13), (1, 17), (2, -5 * 13 + 19)])


def test_eta_many():
    eta = EtaFile(sv([(1, 2), (2, 3), (5, 5), (7, 7), (11, 11), (12, 13)]), 0, 14)
    v = sv([(0, 17), (1, 19), (3, 23), (5, 29), (6, 31), (9, 37), (11, 41)])
    eta.apply_right(v)
    assert v == sv([(0, 17 - 2 * 19 - 5 * 29 - 11 * 41), (1, 19), (3, 23), (5, 29), (6, 31), (9, 37), (11, 41)])
    v = sv([(0, 13), (1, 19), (3, 23), (5, 29), (6, 31), (9, 37), (11, 41)])
    eta.apply_left(v)
    assert v == sv([(0, 13), (1, 19 - 2 * 13), (2, -3 * 13), (3, 23), (5, 29 - 5 * 13), (6, 31), (7, -7 * 13),
                    (9, 37), (11, 41 - 11 * 13), (12, -13 * 13)])


# ------------------------------------------------------------------------------------------
# lower_upper/decomposition/mod.rs tests (:301-491)
# ------------------------------------------------------------------------------------------
def lu(rp, cp, lower, upper, updates=None):
    return LUDecomposition(rp, cp, [sv(c) for c in lower], [sv(c) for c in upper], updates or [])


def test_lu_identity():
    assert LUDecomposition.rows([sv([(0, 1)]), sv([(1, 1)])]) == lu(
        FullPermutation.identity(2), FullPermutation.identity(2), [[]], [[(0, 1)], [(1, 1)]])
    assert LUDecomposition.rows([sv([(0, 1)]), sv([(1, 1)]), sv([(2, 1)])]) == lu(
        FullPermutation.identity(3), FullPermutation.identity(3), [[], []], [[(0, 1)], [(1, 1)], [(2, 1)]])


def test_lu_offdiagonal():
    assert LUDecomposition.rows([sv([(0, 1), (1, 1)]), sv([(1, 1)])]) == lu(
        FullPermutation.identity(2), FullPermutation.identity(2), [[]], [[(0, 1)], [(0, 1), (1, 1)]])
    assert LUDecomposition.rows([sv([(0, 1)]), sv([(0, 1), (1, 1)])]) == lu(
        FullPermutation.identity(2), FullPermutation.identity(2), [[(1, 1)]], [[(0, 1)], [(1, 1)]])
    assert LUDecomposition.rows([sv([(0, 1), (1, 1)]), sv([(0, 1)])]) == lu(
        FullPermutation([1, 0]), FullPermutation.identity(2), [[(1, 1)]], [[(0, 1)], [(1, 1)]])


def test_lu_wikipedia_examples():
    assert LUDecomposition.rows([sv([(0, 4), (1, 3)]), sv([(0, 6), (1, 3)])]) == lu(
        FullPermutation.identity(2), FullPermutation.identity(2),
        [[(1, R(3, 2))]], [[(0, 4)], [(0, 3), (1, R(-3, 2))]])
    expected = lu(FullPermutation.identity(2), FullPermutation.identity(2),
                  [[(1, -1)]], [[(0, -1)], [(0, R(3, 2)), (1, R(1, 2))]])
    assert LUDecomposition.rows([sv([(0, -1), (1, R(3, 2))]), sv([(0, 1), (1, -1)])]) == expected
    assert expected.generate_column(sv([(0, 1)])).column == sv([(0, 2), (1, 2)])
    assert expected.generate_column(sv([(1, 1)])).column == sv([(0, 3), (1, 2)])


def test_subtract_multiple_of_row():
    f = ox.subtract_multiple_of_row_from_other_row
    cases = [
        ([], 1, [], []),
        ([], 1, [(1, 1)], [(1, -1)]),
        ([(1, 1)], 1, [], [(1, 1)]),
        ([(1, 1)], 1, [(2, 3)], [(1, 1), (2, -3)]),
        ([(1, 1)], 1, [(1, 3)], [(1, -2)]),
        ([(1, 1)], R(1, 3), [(1, 3)], []),
        ([(1, 1)], 1, [(0, 3)], [(0, -3), (1, 1)]),
    ]
    for (edit, ratio, other, expected) in cases:
        edit = sv(edit)
        f(edit, Fr(ratio), sv(other))
        assert edit == sv(expected)


# ------------------------------------------------------------------------------------------
# lower_upper/permutation tests
# ------------------------------------------------------------------------------------------
def test_rotate_to_back_roundtrip_and_values():
    q = RotateToBack(1, 5)
    assert [q.forward(i) for i in range(5)] == [0, 4, 1, 2, 3]
    assert [q.backward(q.forward(i)) for i in range(5)] == list(range(5))
    items = sv([(0, 10), (1, 11), (3, 13)])
    q.forward_sorted(items)
    assert items == sv([(0, 10), (2, 13), (4, 11)])
    q.backward_sorted(items)
    assert items == sv([(0, 10), (1, 11), (3, 13)])
    items = sv([(2, 12), (4, 14)])
    q.forward_sorted(items)
    assert items == sv([(1, 12), (3, 14)])
    q.backward_sorted(items)
    assert items == sv([(2, 12), (4, 14)])


def test_full_and_swap_permutation():
    p = FullPermutation([2, 0, 1])
    assert [p.forward(i) for i in range(3)] == [2, 0, 1]
    assert [p.backward(p.forward(i)) for i in range(3)] == [0, 1, 2]
    p.invert()
    assert [p.forward(i) for i in range(3)] == [1, 2, 0]
    s = SwapPermutation(0, 2, 4)
    items = sv([(0, 5), (1, 6), (3, 7)])
    s.forward_sorted(items)
    assert items == sv([(1, 6), (2, 5), (3, 7)])
    items = sv([(0, 5), (2, 6)])
    s.forward_sorted(items)
    assert items == sv([(0, 6), (2, 5)])


# ------------------------------------------------------------------------------------------
# lower_upper/mod.rs tests: tri-solves (:488-603) and change_basis known answers (:605-867)
# ------------------------------------------------------------------------------------------
def test_lu_matmul_identity():
    ident = LUDecomposition.identity(2)
    for col in ([], sv([(0, 1)]), sv([(1, 1)]), sv([(0, 1), (1, 1)])):
        assert ident.invert_upper_right(list(col)) == col
        assert ident.invert_upper_left(list(col)) == col
        assert ident.invert_lower_right(list(col)) == col
        assert ident.invert_lower_left(list(col)) == col


def test_lu_matmul_offdiagonal():
    off = lu(FullPermutation.identity(2), FullPermutation.identity(2), [[(1, 1)]], [[(0, 1)], [(1, 1)]])
    assert off.generate_column([]).column == []
    assert off.generate_column(sv([(0, 1)])).column == sv([(0, 1), (1, -1)])
    assert off.generate_column(sv([(1, 1)])).column == sv([(1, 1)])


def test_lu_change_basis_no_change():
    initial = LUDecomposition.identity(3)
    spike = sv([(1, 1)])
    initial.change_basis(1, ColumnAndSpike(list(spike), list(spike)))
    expected = LUDecomposition.identity(3)
    expected.updates.append((EtaFile([], 1, 3), RotateToBack(1, 3)))
    assert initial == expected


def test_lu_change_basis_from_identity_2():
    ident = LUDecomposition.identity(2)
    spike = sv([(0, 1), (1, 1)])
    ident.change_basis(0, ColumnAndSpike(list(spike), list(spike)))
    assert ident == lu(FullPermutation.identity(2), FullPermutation.identity(2), [[]],
                       [[(0, 1)], [(0, 1), (1, 1)]], [(EtaFile([], 0, 2), RotateToBack(0, 2))])


def test_lu_change_basis_5x5_no_r():
    m = 5
    initial = LUDecomposition.identity(m)
    spike = sv([(0, 2), (1, 3), (2, 5), (3, 7)])
    initial.change_basis(1, ColumnAndSpike(list(spike), list(spike)))
    assert initial == lu(FullPermutation.identity(m), FullPermutation.identity(m), [[]] * (m - 1),
                         [[(0, 1)], [(1, 1)], [(2, 1)], [(3, 1)], [(0, 2), (1, 5), (2, 7), (4, 3)]],
                         [(EtaFile([], 1, m), RotateToBack(1, m))])


def test_lu_change_basis_4x4():
    m = 4
    initial = lu(FullPermutation.identity(m), FullPermutation.identity(m), [[]] * (m - 1),
                 [[(0, 1)], [(1, 1)], [(2, 4)], [(1, 5), (3, 6)]])
    spike = sv([(1, 2), (2, 3), (3, 4)])
    initial.change_basis(1, ColumnAndSpike(list(spike), list(spike)))
    mod = initial
    assert mod == lu(FullPermutation.identity(m), FullPermutation.identity(m), [[]] * (m - 1),
                     [[(0, 1)], [(1, 4)], [(2, 6)], [(1, 3), (2, 4), (3, -R(8, 6))]],
                     [(EtaFile(sv([(3, R(5, 6))]), 1, m), RotateToBack(1, m))])
    assert mod.generate_column(sv([(0, 1)])).column == sv([(0, 1)])
    assert mod.generate_column(sv([(1, 1)])).column == sv([(1, R(-3, 4)), (2, R(9, 16)), (3, R(1, 2))])
    assert mod.generate_column(sv([(2, 1)])).column == sv([(2, R(1, 4))])
    assert mod.generate_column(sv([(3, 1)])).column == sv([(1, R(5, 8)), (2, R(-15, 32)), (3, R(-1, 4))])
    assert mod.basis_inverse_row(0) == sv([(0, 1)])
    assert mod.basis_inverse_row(1) == sv([(1, R(-3, 4)), (3, R(5, 8))])
    assert mod.basis_inverse_row(2) == sv([(1, R(9, 16)), (2, R(1, 4)), (3, R(-15, 32))])
    assert mod.basis_inverse_row(3) == sv([(1, R(1, 2)), (3, R(-1, 4))])


def test_lu_change_basis_elble_sahinidis_5x5():
    """lower_upper/mod.rs:773-867."""
    m = 5
    initial = lu(FullPermutation.identity(m), FullPermutation.identity(m), [[]] * (m - 1), [
        [(0, 11)],
        [(0, 12), (1, 22)],
        [(0, 13), (1, 23), (2, 33)],
        [(0, 14), (1, 24), (2, 34), (3, 44)],
        [(0, 15), (1, 25), (2, 35), (3, 45), (4, 55)],
    ])
    spike = sv([(0, 12), (1, 22), (2, 32), (3, 42)])
    initial.change_basis(1, ColumnAndSpike(list(spike), list(spike)))
    mod = initial
    expected = lu(FullPermutation.identity(m), FullPermutation.identity(m), [[]] * (m - 1), [
        [(0, 11)],
        [(0, 13), (1, 33)],
        [(0, 14), (1, 34), (2, 44)],
        [(0, 15), (1, 35), (2, 45), (3, 55)],
        [(0, 12), (1, 32), (2, 42), (4, R(-215, 363))],
    ], [(EtaFile(sv([(2, R(23, 33)), (3, R(24 * 33 - 34 * 23, 33 * 44)), (4, R(43, 7986))]), 1, m),
         RotateToBack(1, m))])
    assert mod == expected
    col = lambda j: mod.generate_column(sv([(j, 1)])).column
    assert col(0) == sv([(0, R(1, 11))])
    assert col(1) == sv([(0, R(-2, 11)), (1, R(-363, 215)), (2, R(-1, 43)), (3, R(693, 430))])
    assert col(2) == sv([(0, R(1, 11)), (1, R(253, 215)), (2, R(2, 43)), (3, R(-483, 430))])
    assert col(3) == sv([(1, R(1, 86)), (2, R(-1, 43)), (3, R(1, 86))])
    assert col(4) == sv([(1, R(1, 110)), (3, R(-3, 110)), (4, R(1, 55))])
    assert mod.generate_column(sv([(0, 1), (1, 1)])).column == sv(
        [(0, R(-1, 11)), (1, R(-363, 215)), (2, R(-1, 43)), (3, R(693, 430))])
    assert mod.basis_inverse_row(0) == sv([(0, R(1, 11)), (1, R(-2, 11)), (2, R(1, 11))])
    assert mod.basis_inverse_row(1) == sv([(1, R(-363, 215)), (2, R(253, 215)), (3, R(1, 86)), (4, R(1, 110))])
    assert mod.basis_inverse_row(2) == sv([(1, R(-1, 43)), (2, R(2, 43)), (3, R(-1, 43))])
    assert mod.basis_inverse_row(3) == sv([(1, R(693, 430)), (2, R(-483, 430)), (3, R(1, 86)), (4, R(-3, 110))])
    assert mod.basis_inverse_row(4) == sv([(4, R(1, 55))])


# ------------------------------------------------------------------------------------------
# Cross-checks between the two basis-inverse back-ends (the reference instantiates the same
# test bodies with both, e.g. tests/burkardt/test.rs:44 vs :63): in exact arithmetic the pivot
# trace must be identical.
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("make", [problem_1_matrix_data, problem_2_matrix_data])
def test_backends_produce_identical_traces(make):
    traces = []
    for BI in (BasisInverseRows, LUDecomposition):
        tr = []
        out = ox.solve_relaxation(make(), BI, trace=tr.append, check=True)
        assert out["status"] == "optimal"
        traces.append((tr, out["bfs"], out["objective"]))
    assert traces[0] == traces[1]


def test_steepest_descent_is_dantzig_first_index_wins():
    """pivot_rule.rs:113-125: strict ``<`` keeps the lowest index among equal minima."""
    t = problem_2_tableau()   # relative costs: (-3, -3, 0, 0, 0)
    assert SteepestDescent().select_primal_pivot_column(t) == (0, R(-3))
