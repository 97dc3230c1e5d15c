"""`engine.solve_verified` (beyond the reference: legs of (configuration, engine), an outcome stands only when it verifies) on files
of the reference's Netlib directory that need DIFFERENT legs -- profiles/r04_corpus_sweep.md: BNL1 only under the literal rules, MAROS
not on the first leg, GREENBEA under the safeguards on the LU engine -- and on one that the first leg solves.
Expected optima: HiGHS on the same standardised LP (tests/golden/corpus/index.json; NOT the reference: parity unpinned) and the
reference's pin where it holds one (GREENBEA, tests/netlib/test.rs)."""
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import engine
import corpus

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name, leg", [("AFIRO", ("robust", "lu")), ("GREENBEA", ("robust", "lu")), ("MAROS", None),
                                       ("BNL1", ("default", "lu"))])
def test_solve_verified_reaches_the_optimum_and_says_which_leg_did(name, leg):
    md, fixed = corpus.load(name)
    rec = corpus.index()[name]
    oc, t, report = engine.solve_verified(md)
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
    oc, t, report = engine.solve_verified(md, legs=(("default", engine.ENGINE_LU), ("robust", engine.ENGINE_LU)))
    try:
        first = report["legs"][0]
        assert first["outcome"] == "optimal" and first["check_basis"][2] < -0.1, report
        assert oc == engine.OPTIMAL and report["verified"] and report["legs"][-1]["config"] == "robust", report
        assert abs(t.objective_function_value() + fixed - 1878.1248227381) <= 1e-6
    finally:
        if t is not None:
            t.close()
