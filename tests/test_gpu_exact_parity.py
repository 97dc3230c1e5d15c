"""The f64 GPU engines beside the EXACT-rational oracle (oracle/relp_exact.py: the reference's own arithmetic, `RationalBig`
restated with `fractions.Fraction`) on the Netlib, burkardt and cook files of the reference's test suite that the exact oracle solves in seconds: the
whole pivot sequence of both phases and the optimum.  The reference computes in exact rationals, so this -- not agreement with an f64
restatement -- is "the same pivots as the reference": every tie is decided by the reference's rule on exact numbers, and the f64
engines have to land on the same side of every comparison."""
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import engine

pytestmark = pytest.mark.gpu

FILES = ["netlib/AFIRO.SIF", "netlib/SC50B.SIF", "netlib/SC50A.SIF", "netlib/KB2.SIF", "netlib/BLEND.SIF", "netlib/SC105.SIF",
         "netlib/STOCFOR1.SIF", "netlib/SHARE2B.SIF", "netlib/VTP-BASE.SIF", "netlib/RECIPELP.SIF", "netlib/SCAGR7.SIF", "netlib/BORE3D.SIF",
         "netlib/ADLITTLE.SIF", "netlib/SC205.SIF", "netlib/LOTFI.SIF", "netlib/SHARE1B.SIF", "netlib/BOEING2.SIF",
         "burkardt/afiro.mps", "burkardt/testprob.mps", "burkardt/maros.mps", "burkardt/adlittle.mps", "cook/small_example.mps"]
ENGINES = [("revised", engine.ENGINE_REVISED, 0), ("tableau", engine.ENGINE_TABLEAU, -1), ("lu", engine.ENGINE_LU, 11)]
_exact = {}


def exact_trace(name):
    if name not in _exact:
        from lp_files import exact_solve, load
        gf, ex, md, emd = load(name, fixed=name.endswith(".SIF"))
        tr = []
        status, obj, sol = exact_solve(gf, emd, trace=tr.append)
        assert status == "optimal"
        _exact[name] = (gf, md, [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr], obj)
    return _exact[name]


@pytest.mark.parametrize("ename,kind,block", ENGINES, ids=[e[0] for e in ENGINES])
@pytest.mark.parametrize("name", FILES)
def test_f64_engines_walk_the_exact_pivot_sequence(name, ename, kind, block):
    gf, md, want, obj = exact_trace(name)
    t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 14)
    try:
        assert t.solve_relaxation() == engine.OPTIMAL
        tr = t.trace()
        same = next((k for k, (a, b) in enumerate(zip(tr, want)) if a != b), min(len(tr), len(want)))
        assert tr == want, f"{name} / {ename}: common prefix {same} of {len(want)} exact pivots (engine: {len(tr)})"
        got = t.objective_function_value() + float(gf.fixed_cost)
        assert abs(got - float(obj)) <= 1e-9 * max(1.0, abs(float(obj)))
    finally:
        t.close()
