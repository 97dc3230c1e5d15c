"""MPS reader + standardisation + solution reconstruction against the
reference's own file-driven known answers, solved with the exact oracle (small files) or the f64 C
oracle (larger ones).  NB: presolve is not restated, so these pin optimal values, not the
reference's post-presolve MatrixData."""
from fractions import Fraction as Fr

import numpy as np
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import mps
from oracle import relp_exact as ox
from oracle import relp_f64

from lp_files import exact_solve, load


def test_number_parsing_is_exact():
    """io/mps/number/parse.rs:53-121."""
    assert mps.parse_number("1") == 1 and mps.parse_number("-.022") == Fr(-22, 1000)
    assert mps.parse_number("12.5") == Fr(25, 2) and mps.parse_number("3.") == 3 and mps.parse_number(".5") == Fr(1, 2)
    with pytest.raises(mps.MPSError):
        mps.parse_number("1e3")


def test_rows_are_sorted_by_name_and_two_pairs_per_line():
    """parse/mod.rs:296 (rows sorted by name) and free.rs `five_and_six` (third pair dropped:
    tests/cook small_example loses `r2 2`, which is why its optimum is -243/4)."""
    gf, ex, md, emd = load("cook/small_example.mps")
    status, obj, _ = exact_solve(gf, emd, ox.LUDecomposition)
    assert status == "optimal" and obj == Fr(-243, 4)                       # tests/cook/test.rs:32


def test_burkardt_testprob():
    gf, ex, md, emd = load("burkardt/testprob.mps")
    status, obj, sol = exact_solve(gf, emd, ox.LUDecomposition)
    assert (status, obj) == ("optimal", 54)                                # tests/burkardt/test.rs:170-183
    assert (sol["X1"], sol["X2"], sol["X3"]) == (4, -1, 6)


def test_burkardt_maros():
    gf, ex, md, emd = load("burkardt/maros.mps")
    status, obj, sol = exact_solve(gf, emd, ox.LUDecomposition)
    assert (status, obj) == ("optimal", Fr(385, 3))                        # tests/burkardt/test.rs:128-146
    assert (sol["VOL1"], sol["VOL2"], sol["VOL3"], sol["VOL4"]) == (Fr(10, 3), Fr(40, 3), 20, 0)


def test_burkardt_nazareth_is_unbounded():
    gf, ex, md, emd = load("burkardt/nazareth.mps")
    assert exact_solve(gf, emd, ox.LUDecomposition)[0] == "unbounded"       # tests/burkardt/test.rs:149-157


def test_burkardt_afiro_exact():
    gf, ex, md, emd = load("burkardt/afiro.mps")
    status, obj, sol = exact_solve(gf, emd, ox.LUDecomposition)
    assert (status, obj) == ("optimal", Fr(-406659, 875))                  # tests/burkardt/test.rs:56-110
    # `Solution::is_probably_equal_to(.., 0.1)` (solution.rs:47-79): equal objective and more than 10 % of
    # the values equal (the LP has alternative optima)
    pins = {"X01": 80, "X02": Fr(51, 2), "X03": Fr(109, 2), "X04": Fr(424, 5), "X06": Fr(255, 14), "X14": Fr(255, 14),
            "X16": 999, "X22": 500, "X23": Fr(11898, 25), "X24": Fr(602, 25), "X26": 215, "X36": Fr(11898, 35),
            "X37": Fr(11898, 35), "X07": 0, "X08": 0, "X09": 0, "X10": 0, "X11": 0, "X12": 0, "X13": 0, "X15": 0,
            "X25": 0, "X28": 0, "X29": 0, "X30": 0, "X31": 0, "X32": 0, "X33": 0, "X34": 0, "X35": 0, "X38": 0, "X39": 0}
    assert set(pins) == set(sol)
    assert sum(1 for k, v in pins.items() if sol[k] == v) / len(pins) > 0.1


def test_burkardt_adlittle_exact_optimum_and_f64_trace():
    """Config C1: the exact optimum of tests/burkardt/test.rs:34-54, and the f64 C oracle walks the
    exact pivot sequence (needs the tie band on reduced costs: d_37 == d_38 exactly at pivot 72)."""
    gf, ex, md, emd = load("burkardt/adlittle.mps")
    tr = []
    status, obj, _ = exact_solve(gf, emd, ox.BasisInverseRows, trace=tr.append)
    assert status == "optimal"
    assert obj == Fr(24975305659811992079614961229, 120651674036153428931840)
    ref = relp_f64.OracleF64(md)
    assert ref.run() == "optimal"
    assert ref.trace == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
    assert abs(ref.objective + float(gf.fixed_cost) - float(obj)) <= 1e-9 * float(obj)


UNICAMP = [("model_data_1", Fr(123, 38)), ("model_data_3_1", 70), ("model_data_3_2", 180), ("model_data_3_3", 245),
           ("model_data_3_4", 2250), ("model_data_4", 7), ("model_data_6", 28)]


# full solutions the reference compares with `assert_eq!` (tests/unicamp/test.rs:40-101)
UNICAMP_VALUES = {
    "model_data_3_1": {"SUP1": Fr(200, 3), "SUP2": Fr(100, 3), "SUP3": Fr(100)},
    "model_data_3_2": {"SUP1": Fr(25), "SUP2": Fr(75)},
    "model_data_3_3": {"SUP1": Fr(100), "SUP2": Fr(150)},
    "model_data_3_4": {"RAW1": Fr(5), "RAW2": Fr(3), "RAW3": Fr(4), "PRODUCT": Fr(500)},
    "model_data_4": {"COL01": Fr(1), "COL02": Fr(2), "COL03": Fr(2)},
}


@pytest.mark.parametrize("name,objective", UNICAMP)
def test_unicamp(name, objective):
    """tests/unicamp/test.rs (the non-ignored cases), `Carry<_, LUDecomposition<_>>`: exact objective, and the
    exact variable values where the reference asserts them (presolve + standardisation + reconstruction)."""
    gf, ex, md, emd = load(f"unicamp/{name}.mps")
    status, obj, sol = exact_solve(gf, emd, ox.LUDecomposition)
    assert (status, obj) == ("optimal", objective)
    if name in UNICAMP_VALUES:
        assert sol == UNICAMP_VALUES[name]


NETLIB = [("AFIRO", -464.75314, 1e-3), ("SC50A", -6.457507706e+01, 1e-5), ("SC50B", -70, 1e-10),
          ("KB2", -1.749900130e+03, 1e-3), ("SC105", -5.220206121e+01, 1e-3), ("ADLITTLE", 2.254949632e+05, 1e-3),
          ("STOCFOR1", -4.113197622e+04, 1e-3), ("BLEND", -30.81215, 1e-3), ("SCAGR7", -2.331389824e+06, 1e-1),
          ("SC205", -5.220206121e+01, 1e-5), ("SHARE2B", -4.157322407e+02, 1e-5), ("RECIPELP", -0.266616e3, 1e-2),
          ("LOTFI", -0.2526470606188e2, 1e-6), ("VTP-BASE", 0.1298314624613613657395984384889e6, 1e-2),
          ("SHARE1B", -0.7658931857918568112797274346007e5, 1e-3)]
# BORE3D and BOEING2 need the reference's presolve (rust-lp_amd/presolve.py): on the un-presolved problem
# the phase-1 end state differs and the pinned optimum is not reached.
NETLIB.append(("BORE3D", 0.13730803942084927215581987251301e4, 1e-2))


@pytest.mark.parametrize("name,objective,tol", NETLIB)
def test_netlib_objectives_f64_oracle(name, objective, tol):
    """tests/netlib/test.rs:10-125 (fixed-format parser, tests/netlib/mod.rs:54) through the f64 C
    oracle; tolerances are the reference's."""
    gf, ex, md, emd = load(f"netlib/{name}.SIF", fixed=True)
    ref = relp_f64.OracleF64(md)
    assert ref.run() == "optimal"
    assert abs(ref.objective + float(gf.fixed_cost) - objective) < max(tol, 1e-9 * abs(objective))


def test_boeing2_exact_oracle_reaches_the_pinned_optimum():
    """tests/netlib/test.rs:114-119 through presolve + the exact oracle.  The path is the one the reference's
    release-built integration tests take (see `usize_sub` in oracle/relp_exact.py): an artificial variable
    that re-entered the basis in a foreign row survives phase 1 with a wrapped index and stays basic at
    value zero."""
    gf, ex, md, emd = load("netlib/BOEING2.SIF", fixed=True)
    status, obj, _ = exact_solve(gf, emd)
    assert status == "optimal"
    assert abs(float(obj) - (-0.31501872801520287870462195913263e3)) < 1e-3


def test_boeing2_f64_oracle_walks_the_exact_path():
    """The f64 oracle keeps the surviving artificial like the release-built reference does (wrapped index,
    squeezed into int32 with the order preserved) and walks the exact oracle's 416 pivots."""
    gf, ex, md, emd = load("netlib/BOEING2.SIF", fixed=True)
    tr = []
    status, obj, _ = exact_solve(gf, emd, trace=tr.append)
    ref = relp_f64.OracleF64(md)
    assert ref.run(max_iters=20000) == status == "optimal"
    assert abs(ref.objective + float(gf.fixed_cost) - float(obj)) < 1e-9 * abs(float(obj))
    na = 123                                                   # artificial variables of the presolved problem

    def squeeze(j):                                            # (a - na) mod 2^64  ->  INT32_MAX - (na - 1 - a)
        return j if j < (1 << 62) else (1 << 31) - 1 - (na - 1 - (j - (1 << 64) + na))
    assert ref.trace == [(e["phase"], e["entering"], e["row"], squeeze(e["leaving"])) for e in tr]


@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "SC50B"])
def test_netlib_small_exact_equals_f64_trace(name):
    gf, ex, md, emd = load(f"netlib/{name}.SIF", fixed=True)
    tr = []
    status, obj, _ = exact_solve(gf, emd, trace=tr.append)
    ref = relp_f64.OracleF64(md)
    assert ref.run() == status == "optimal"
    assert ref.trace == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]


def test_free_and_fixed_parsers_agree_on_adlittle():
    from rust_lp_amd import general_form
    import os
    from lp_files import GOLDEN
    text = open(os.path.join(GOLDEN, "netlib", "ADLITTLE.SIF")).read()
    a, b = mps.parse(text), mps.parse_fixed(text)
    assert a.rows == b.rows and a.columns == b.columns and a.rhss == b.rhss and a.cost_values == b.cost_values


def test_25fv47_f64_oracle_under_the_default_tolerances():
    """Config C3.  tests/netlib/test.rs:152-158 pins 5.5018459e+03 (8 digits; the reference ignores the test
    as too expensive for exact arithmetic).  The default tolerances (the engine's relp_default_config: tol_pivot
    1e-5, the others 1e-7 .. 1e-11) reach it; a pivot tolerance of 1e-6 or below accepts rounding noise of B^-1
    as a pivot element on this unscaled file and loses feasibility in phase 1."""
    gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
    assert relp_f64.DEFAULT_TOLERANCES["tol_pivot"] == 1e-5
    ref = relp_f64.OracleF64(md)
    assert ref.run(max_iters=100000) == "optimal"
    assert abs(ref.objective + float(gf.fixed_cost) - 5.5018459e+03) < 1e-4


def test_acc_tight4_standardises_and_the_oracle_walks_phase_one():
    """MIPLIB acc-tight4 (tests/miplib/test.rs:14-18, ignored by the reference as too expensive): the file goes through
    the reader, presolve and standardisation (3,285 constraint rows, every one of the 1,620 columns bounded), and the
    f64 oracle starts phase 1 on it -- a few hundred of the > 400,000 pivots it would need."""
    gf, ex, md, emd = load("miplib/acc-tight4.mps", fixed=False)
    assert (md.nr_eq, md.nr_range, md.nr_le, md.nr_ge, md.nr_normal) == (297, 0, 756, 2232, 1620)
    assert int(np.isfinite(md.upper_bound).sum()) == 1620 and float(gf.fixed_cost) == 0.0
    ref = relp_f64.OracleF64(md)
    assert ref.run(300) == "iteration_limit"
    assert ref.phase == 1 and len(ref.trace) == 300 and all(ph == 1 for ph, *_ in ref.trace)
