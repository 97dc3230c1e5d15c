"""`engine.solve_verified` (beyond the reference: legs of (configuration, engine), an outcome stands only when it verifies) on files
of the reference's Netlib directory that need DIFFERENT legs -- profiles/r04_corpus_sweep.md: BNL1 and MAROS not on the first leg, GREENBEA under the safeguards on the LU engine -- and on one that the first leg solves.
Expected optima: HiGHS on the same standardised LP (tests/golden/corpus/index.json; NOT the reference: parity unpinned) and the
reference's pin where it holds one (GREENBEA, tests/netlib/test.rs)."""
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import engine
import corpus

pytestmark = pytest.mark.gpu
AS_READ = tuple(leg for leg in engine.VERIFIED_LEGS if leg[0] == "read")


@pytest.mark.parametrize("name, leg", [("AFIRO", ("robust", "lu")), ("GREENBEA", ("robust", "lu")), ("MAROS", None),
                                       ("BNL1", None)])
def test_solve_verified_reaches_the_optimum_and_says_which_leg_did(name, leg):
    md, fixed = corpus.load(name)
    rec = corpus.index()[name]
    oc, t, report = engine.solve_verified(md, legs=AS_READ)        # (the data as read; with the scaled legs in between the order differs)
    try:
        assert oc == engine.OPTIMAL and report["verified"], report
        obj = t.objective_function_value() + fixed
        want = rec["highs_objective"]
        assert abs(obj - want) <= 1e-6 * max(1.0, abs(want)), (obj, want, report)
        if rec.get("reference_pin") is not None:
            assert abs(obj - rec["reference_pin"]) <= max(rec["reference_tolerance"], 1e-9 * abs(rec["reference_pin"]))
        last = report["legs"][-1]
        if leg is not None:
            assert (last["config"], last["engine"]) == leg, report
        else:
            assert len(report["legs"]) > 1, report             # (which leg gets there depends on the slicing of the runs: not pinned)
        ident, basic, min_b = last["check_basis"]
        assert ident <= engine.VERIFY_IDENTITY and basic <= engine.VERIFY_BASIC and min_b >= engine.VERIFY_MIN_B
    finally:
        if t is not None:
            t.close()


def test_solve_verified_does_not_accept_an_optimum_on_an_infeasible_basis():
    """SCORPION under the literal rules ends `optimal` at 1858.44 on a basis with b_i = -0.27 -- the reference's artificial-index
    quirk, on every engine (tests/test_scorpion.py) -- where the optimum is 1878.1248227381 (tests/netlib/test.rs:128-134): the check
    turns that leg down and the next one is accepted."""
    md, fixed = corpus.load("SCORPION")
    oc, t, report = engine.solve_verified(md, legs=(("read", "default", engine.ENGINE_LU), ("read", "robust", engine.ENGINE_LU)))
    try:
        first = report["legs"][0]
        assert first["outcome"] == "optimal" and first["check_basis"][2] < -0.1, report
        assert oc == engine.OPTIMAL and report["verified"] and report["legs"][-1]["config"] == "robust", report
        assert abs(t.objective_function_value() + fixed - 1878.1248227381) <= 1e-6
    finally:
        if t is not None:
            t.close()


@pytest.mark.parametrize("name", ["PEROLD", "PILOT-JA", "PILOTNOV", "PILOT4", "BNL1", "MODSZK1"])
def test_scaled_legs_solve_what_no_leg_solves_on_the_data_as_read(name):
    """MatrixData.scaled (powers of two) in front: PEROLD, PILOT-JA, PILOTNOV end `no_row_phase_one` / at the pivot limit on every
    engine and configuration as read (profiles/r04_corpus_sweep.md) and in a verified optimum equal to HiGHS's when scaled; PILOT4,
    BNL1, MODSZK1 need a late leg as read and are solved by the first scaled one.  The solution comes back in the units of the data."""
    import numpy as np
    md, fixed = corpus.load(name)
    want = corpus.index()[name]["highs_objective"]
    oc, t, report = engine.solve_verified(md)
    try:
        assert oc == engine.OPTIMAL and report["verified"] and report["scaled"] and len(report["legs"]) <= 3, report
        assert abs(t.objective_function_value() + fixed - want) <= 1e-6 * max(1.0, abs(want))
        x = np.zeros(md.nr_normal)
        for j, v in md.unscale_bfs(t.current_bfs(), report["row_scale"], report["column_scale"]):
            if j < md.nr_normal:
                x[j] = v
        assert abs(float(np.dot(np.asarray(md.cost), x)) + fixed - want) <= 1e-6 * max(1.0, abs(want))
    finally:
        if t is not None:
            t.close()


@pytest.mark.parametrize("name", ["TUFF", "DEGEN3", "CYCLE"])
def test_the_largest_coefficient_rule_in_phase_one_ends_the_cycling(name):
    """TUFF, DEGEN3, CYCLE: 300,000+ pivots without an end under the reference's phase-1 rule (FirstProfitableWithMemory) on every
    engine and configuration, and under Bland's rule (profiles/r04_stall_probe.md); the `robust-dantzig` legs -- the safeguards with
    PivotRule::SteepestDescent in phase 1 as well -- end in a verified optimum equal to HiGHS's."""
    md, fixed = corpus.load(name)
    want = corpus.index()[name]["highs_objective"]
    oc, t, report = engine.solve_verified(md, legs=tuple(leg for leg in engine.VERIFIED_LEGS if leg[1] == "robust-dantzig"))
    try:
        assert oc == engine.OPTIMAL and report["verified"], report
        assert abs(t.objective_function_value() + fixed - want) <= 1e-6 * max(1.0, abs(want))
    finally:
        if t is not None:
            t.close()


def test_infeasible_stands_only_when_two_engines_say_so_on_the_data_as_read():
    """x0 + x1 <= 1, x0 + x1 >= 2: every leg ends `infeasible`; it is a verified outcome after all legs have run and two engines
    said so on the data as read, and not when only scaled legs ran (on scaled data two engines agreed on a wrong `infeasible` for
    WOODW: absolute tolerances on scaled rows)."""
    import numpy as np
    from rust_lp_amd import MatrixData
    md = MatrixData(nr_normal=2, nr_eq=0, nr_range=0, nr_le=1, nr_ge=1, b=np.array([1.0, 2.0]), cost=np.array([1.0, 1.0]),
                    upper_bound=np.full(2, np.inf), col_ptr=np.array([0, 2, 4], dtype=np.int64), row_idx=np.array([0, 1, 0, 1], dtype=np.int32),
                    values=np.ones(4))
    oc, t, report = engine.solve_verified(md)
    assert t is None and oc == engine.INFEASIBLE and report["verified"] and len(report["legs"]) == len(engine.VERIFIED_LEGS) == 13, report
    oc, t, report = engine.solve_verified(md, legs=tuple(leg for leg in engine.VERIFIED_LEGS if leg[0] == "scaled"))
    assert t is None and oc == engine.INFEASIBLE and not report["verified"], report
