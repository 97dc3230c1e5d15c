"""Netlib SCORPION: the one pin the reference holds and marks `#[ignore = "Incorrect optimal value."]`
(tests/netlib/test.rs:128-134, 1878.1248227381 +- 1e-2).

The finding (missing 3): the "incorrect" value is the reference's own artificial-index bug at work.  Phase 1 leaves 29
artificial variables basic at zero level (the LP has 29 redundant equality rows after presolve); `remove_artificial_basis_variables`
pivots 27 of them out and pushes the INDICES of the other two as redundant rows (phase_one.rs:252) -- 226 and 227, which name other
rows than the artificials' own -- so two non-redundant rows are deleted and the relaxation's optimum, 1858.4421809618, comes out.
The literal rules reproduce that on the f64 oracle and on all three engines with the same 593 pivots; `RELP_ARTIFICIAL_TEXTBOOK`
(27 of the stuck artificials sit in foreign basis positions and are exchanged into their own rows first) removes the 29 redundant
rows and ends at the pin to 1e-10, which is also what HiGHS returns for the standardised LP (tests/golden/corpus/index.json)."""
import pytest

import rust_lp_amd  # noqa: F401
from oracle import relp_f64

PIN = 1878.1248227381
REFERENCES_OWN_VALUE = 1858.4421809618


def test_oracle_reproduces_the_references_incorrect_value_and_the_pin():
    import corpus
    md, fixed = corpus.load("SCORPION")
    assert abs(corpus.index()["SCORPION"]["highs_objective"] - PIN) < 1e-9
    lit = relp_f64.OracleF64(md)
    assert lit.run() == "optimal"
    assert abs(lit.objective + fixed - REFERENCES_OWN_VALUE) < 1e-8 and lit.filtered_rows() == [226, 227] and len(lit.trace) == 593
    fix = relp_f64.OracleF64(md, artificial_removal=1)
    assert fix.run() == "optimal"
    assert abs(fix.objective + fixed - PIN) < 1e-9 and len(fix.filtered_rows()) == 29 and fix.nr_position_exchanges == 27
    # the LU back-end walks the same pivots under the literal rules
    lu = relp_f64.OracleF64(md, basis_inverse=1, lu_threshold=0.1)
    assert lu.run() == "optimal" and lu.trace == lit.trace


@pytest.mark.gpu
@pytest.mark.parametrize("kind,block", [(0, 0), (1, 32), (2, -1)])
def test_engines_reproduce_both_values(kind, block):
    import corpus
    from rust_lp_amd import engine
    md, fixed = corpus.load("SCORPION")
    lit = relp_f64.OracleF64(md)
    lit.run()
    t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=4096)
    assert t.solve_relaxation() == engine.OPTIMAL
    assert abs(t.objective_function_value() + fixed - REFERENCES_OWN_VALUE) < 1e-7 and t.nr_rows() == md.nr_rows - 2
    assert t.trace() == lit.trace
    t.close()
    t = engine.Tableau(md, engine=kind, update_block=block, artificial_removal=engine.ARTIFICIAL_TEXTBOOK)
    assert t.solve_relaxation() == engine.OPTIMAL
    assert abs(t.objective_function_value() + fixed - PIN) < 1e-8 and t.nr_rows() == md.nr_rows - 29
    t.close()
