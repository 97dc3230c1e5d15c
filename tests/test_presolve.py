"""Presolve restatement (rust-lp_amd/presolve.py) pinned by the reference's own known-answer tests:
tests/golden/presolve_changes.json holds the input problem and the expected `Changes` (or error) of every
test of /root/reference/src/data/linear_program/general_form/presolve/test/changes.rs, extracted as data by
scripts/gen_presolve_fixtures.py."""
import json
import os
from fractions import Fraction

import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import general_form, presolve

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "presolve_changes.json")))


def fr(v):
    return None if v is None else Fraction(v[0], v[1])


def ctype(t):
    return ("R", fr(t[1])) if t[0] == "R" else (t[0],)


def removed(v):
    if v[0] == "solved":
        return ("solved", fr(v[1]))
    return ("function", fr(v[1]), [(j, fr(c)) for (j, c) in v[2]])


def build(case):
    ncols = case["ncols"]
    columns = [[] for _ in range(ncols)]
    for i, row in enumerate(case["rows"]):
        for j, v in enumerate(row):
            if fr(v) != 0:
                columns[j].append((i, fr(v)))
    variables = [general_form.Variable(fr(v["cost"]), fr(v["lower_bound"]), fr(v["upper_bound"]), fr(v["shift"]), v["flipped"])
                 for v in case["variables"]]
    return general_form.GeneralForm(case["objective"] == "max", columns, [ctype(t) for t in case["constraint_types"]],
                                    [fr(v) for v in case["b"]], variables, ["x"] * ncols, fr(case["fixed_cost"]))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_presolve_changes_match_the_reference_known_answers(case):
    gf = build(case)
    kind, expected = case["expected"]
    if kind == "err":
        with pytest.raises(presolve.Infeasible if expected == "infeasible" else presolve.Unbounded):
            presolve.compute_presolve_changes(gf)
        return
    got = presolve.compute_presolve_changes(gf)
    assert got["b"] == {int(k): fr(v) for k, v in expected["b"].items()}
    assert got["constraints"] == {i: ctype(t) for (i, t) in expected["constraints"]}
    assert got["fixed_cost"] == fr(expected["fixed_cost"])
    assert got["bounds"] == {tuple(json.loads(k)): fr(v) for k, v in expected["bounds"].items()}
    assert got["removed_variables"] == [(j, removed(v)) for (j, v) in expected["removed_variables"]]
    assert got["constraints_marked_removed"] == expected["constraints_marked_removed"]


def test_presolve_applied_solves_the_reference_example_completely():
    """presolve/test/with_application.rs:27-129: six variables, four rows; bound rows, a fixed variable and
    domain propagation determine every variable, so presolve ends in `FiniteOptimum` (here: `Solved`)."""
    F = Fraction
    rows = [[2, 0, 0, 0, 0, 0], [3, 5, 0, 0, 0, 0], [7, 11, 13, 0, 0, 0], [17, 19, 23, 0, 29, 31]]
    columns = [[(i, F(rows[i][j])) for i in range(4) if rows[i][j]] for j in range(6)]
    x2_lower = (F(103) - F(101) / 2 * 3) / 5
    V = general_form.Variable
    variables = [V(F(211), None, None), V(F(223), x2_lower, None), V(F(227), None, None), V(F(-229), None, F(131)),
                 V(F(233), F(-30736, 65 * 29), F(123)), V(F(0), F(5), None)]
    names = ["XONE", "XTWO", "XTHREE", "XFOUR", "XFIVE", "XSIX"]
    gf = general_form.GeneralForm(False, columns, [("E",), ("L",), ("G",), ("E",)], [F(101), F(103), F(107), F(109)],
                                  variables, names, F(1))
    with pytest.raises(general_form.Solved) as info:
        gf.presolve()
    expected_cost = (F(1) + F(211 * 101, 2) + F(223 * -97, 10) + F(227 * -699, 65) + F(-229 * 131) + F(233 * -30736, 1885))
    assert info.value.objective == expected_cost
    assert info.value.values == {"XONE": F(101, 2), "XTWO": x2_lower,
                                 "XTHREE": (F(-3601, 5) + F(29 * 30736, 1885)) / 23, "XFOUR": F(131),
                                 "XFIVE": F(-30736, 65 * 29), "XSIX": F(5)}
