"""The engines beside the f64 oracles of their own back-ends on the reference's Netlib directory.  First the LU engine beside the f64 oracle of the reference's OWN back-end (`LUDecomposition` + eta file at the reference's cadence,
oracle/relp_f64_lu.h) on the small files of the reference's Netlib directory (tests/golden/corpus; rows <= 420): the whole pivot
sequence, the objective and the number of rows, under `relp_default_config` -- the reference's rules literally.  The reference holds
no pins for most of these files ("parity unpinned" with respect to the reference itself); what is pinned is the build's device
path against the CPU restatement of the reference's arithmetic, pivot by pivot, with HiGHS's optimum of the same standardised LP as
the outside check.  Left out: the files on which the ORACLE does not reach the optimum under the literal rules (FORPLAN, D6CUBE:
pivot limit; TUFF, SCFXM1, STAIR, SCSD8: a zero pivot / singular basis; SCORPION: the reference's artificial-index quirk,
tests/test_scorpion.py)."""
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import engine
from oracle import relp_f64
import corpus

pytestmark = pytest.mark.gpu

NAMES = ["AFIRO", "KB2", "SC50B", "SC50A", "ADLITTLE", "BLEND", "BEACONFD", "VTP-BASE", "SHARE2B", "STOCFOR1", "SC105", "SCAGR7",
         "BORE3D", "SHARE1B", "RECIPELP", "BRANDY", "ISRAEL", "LOTFI", "SCSD1", "SC205", "BOEING2", "BANDM", "SCTAP1", "AGG2", "AGG3",
         "SCSD6", "CAPRI", "SHIP04S", "STANDATA", "SHIP08S", "SHIP04L", "SHIP12S"]


@pytest.mark.parametrize("device_factorisation", [False, True])
@pytest.mark.parametrize("name", NAMES)
def test_lu_engine_walks_the_lu_oracles_pivots_on_the_small_netlib_files(name, device_factorisation):
    md, fixed = corpus.load(name)
    want = corpus.index()[name]["highs_objective"]
    ref = relp_f64.OracleF64(md.ensure_csc(), basis_inverse=1, lu_threshold=0.1)
    assert ref.run() == "optimal"
    assert abs(ref.objective + fixed - want) <= 1e-6 * max(1.0, abs(want))
    t = engine.Tableau(md, engine=engine.ENGINE_LU, update_block=11, trace_capacity=1 << 15)
    t.lu_set_device_factorisation(device_factorisation)      # (the refactorisation on the host, and by k_lu_factor / k_lu_schedules)
    try:
        assert t.solve_relaxation() == engine.OPTIMAL
        tr = t.trace()
        same = next((k for k, (a, b) in enumerate(zip(tr, ref.trace)) if a != b), min(len(tr), len(ref.trace)))
        assert tr == ref.trace, f"{name}: common prefix {same} of {len(ref.trace)} pivots"
        assert abs(t.objective_function_value() - ref.objective) <= 1e-9 * max(1.0, abs(ref.objective))
        assert t.nr_rows() == ref.m
        st = t.lu_device_factorisation_stats()
        assert (st["device_factorisations"] > 0) == device_factorisation and st["host_fallbacks"] == 0
    finally:
        t.close()


PART_WAYS = {"FFFFF800", "WOODW"}        # both arithmetic orders are f64: these two part ways and meet at the optimum
MID = ["SCAGR25", "ETAMACRO", "FINNIS", "FFFFF800", "BOEING1", "SCRS8", "STANDMPS", "BNL1", "SHELL", "GFRD-PNC", "SEBA", "SHIP08L",
       "CZPROB", "FIT1D", "SCTAP2", "SHIP12L", "WOODW"]


@pytest.mark.parametrize("name", MID)
def test_lu_engine_and_lu_oracle_on_the_mid_size_netlib_files(name):
    """430 - 1,050 rows, 500 - 4,600 pivots: the same comparison (host factorisation): 15 of the 17 files pivot for pivot, two
    (PART_WAYS) for a prefix of at least 100 pivots and to the same optimum.  (MAROS is not here: the oracle solves it, the LU engine
    parts ways and ends `unbounded` under the literal rules -- profiles/r04_corpus_sweep.md; `relp_robust_config` and
    `solve_verified` reach its optimum.)  Files on which the oracle itself fails under
    the literal rules are left out (SCFXM2, SCFXM3, PEROLD, the PILOTs: a zero pivot; MODSZK1, TRUSS: the pivot limit); DEGEN2 and
    25FV47 take the oracle 24 s each (25FV47 has its own test in tests/test_gpu_lu_vs_lu_oracle.py)."""
    md, fixed = corpus.load(name)
    want = corpus.index()[name]["highs_objective"]
    ref = relp_f64.OracleF64(md.ensure_csc(), basis_inverse=1, lu_threshold=0.1)
    assert ref.run() == "optimal"
    assert abs(ref.objective + fixed - want) <= 1e-6 * max(1.0, abs(want))
    t = engine.Tableau(md, engine=engine.ENGINE_LU, update_block=11, trace_capacity=1 << 15)
    try:
        assert t.solve_relaxation() == engine.OPTIMAL
        tr = t.trace()
        same = next((k for k, (a, b) in enumerate(zip(tr, ref.trace)) if a != b), min(len(tr), len(ref.trace)))
        print(f"{name}: common prefix {same} of {len(ref.trace)} pivots (engine: {len(tr)})")
        if name in PART_WAYS:
            # (a reduced cost that is zero within rounding decides differently: WOODW at pivot 191 enters column 1389 on the device,
            # 1390 in the oracle; from there the two walk different vertices to the same optimum)
            assert same >= 100 and tr != ref.trace
            assert abs(t.objective_function_value() - ref.objective) <= 1e-7 * max(1.0, abs(ref.objective))
        else:
            assert tr == ref.trace, f"{name}: common prefix {same} of {len(ref.trace)} pivots"
            assert abs(t.objective_function_value() - ref.objective) <= 1e-9 * max(1.0, abs(ref.objective))
    finally:
        t.close()


@pytest.mark.parametrize("kind", [engine.ENGINE_REVISED, engine.ENGINE_TABLEAU], ids=["revised", "tableau"])
@pytest.mark.parametrize("name", NAMES)
def test_explicit_inverse_and_tableau_engines_walk_the_rows_oracles_pivots(name, kind):
    """The same small files on the explicit-inverse engine (`Carry<_, BasisInverseRows<_>>`) and on the dense tableau, beside the f64
    oracle of THAT back-end (oracle/relp_f64.c): whole pivot sequences under `relp_default_config`."""
    md, fixed = corpus.load(name)
    want = corpus.index()[name]["highs_objective"]
    ref = relp_f64.OracleF64(md.ensure_csc())
    assert ref.run() == "optimal"
    assert abs(ref.objective + fixed - want) <= 1e-6 * max(1.0, abs(want))
    t = engine.Tableau(md, engine=kind, trace_capacity=1 << 15)
    try:
        assert t.solve_relaxation() == engine.OPTIMAL
        tr = t.trace()
        same = next((k for k, (a, b) in enumerate(zip(tr, ref.trace)) if a != b), min(len(tr), len(ref.trace)))
        assert tr == ref.trace, f"{name}: common prefix {same} of {len(ref.trace)} pivots"
        assert abs(t.objective_function_value() - ref.objective) <= 1e-9 * max(1.0, abs(ref.objective))
    finally:
        t.close()
