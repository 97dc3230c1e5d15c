"""`MatrixData.scaled` / `unscale_bfs` (beyond the reference: geometric scaling by powers of two in front of the engines).  CPU tier:
the f64 oracle (test infrastructure) solves an LP of the reference's Netlib directory as read and scaled -- same optimum, and the
scaled solution brought back is the unscaled one."""
import numpy as np
import pytest

import rust_lp_amd  # noqa: F401
import corpus
from oracle import relp_f64


def bfs_of(o):
    return sorted((int(j), float(v)) for j, v in zip(o.basis(), o.b()) if v != 0.0)


@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "SC50B", "ADLITTLE", "BLEND", "BOEING2"])
def test_scaled_lp_has_the_same_optimum_and_its_solution_comes_back(name):
    md, fixed = corpus.load(name)
    smd, r, s = md.scaled()
    assert np.all(np.log2(r) == np.round(np.log2(r))) and np.all(np.log2(s) == np.round(np.log2(s)))       # exact in f64
    a0, a1 = np.abs(md.ensure_csc().values), np.abs(smd.values)
    assert a1[a1 > 0].max() / a1[a1 > 0].min() <= a0[a0 > 0].max() / a0[a0 > 0].min()                       # never a wider spread
    plain, scaled = relp_f64.OracleF64(md.ensure_csc()), relp_f64.OracleF64(smd)
    assert plain.run() == "optimal" and scaled.run() == "optimal"
    assert abs(plain.objective - scaled.objective) <= 1e-9 * max(1.0, abs(plain.objective))
    back = dict(md.unscale_bfs(bfs_of(scaled), r, s))
    # the brought-back solution satisfies the constraints of the LP as read: A x (+ slacks) = b on every row
    x = np.zeros(md.nr_normal)
    for j, v in back.items():
        if j < md.nr_normal:
            x[j] = v
    csc = md.ensure_csc()
    ax = np.zeros(md.nr_constraints)
    for j in range(md.nr_normal):
        lo, hi = csc.col_ptr[j], csc.col_ptr[j + 1]
        ax[csc.row_idx[lo:hi]] += csc.values[lo:hi] * x[j]
    b = np.asarray(md.b, dtype=float)
    tol = 1e-7 * max(1.0, np.abs(b).max())
    eq = slice(0, md.nr_eq)
    le = slice(md.nr_eq + md.nr_range, md.nr_eq + md.nr_range + md.nr_le)
    ge = slice(md.nr_eq + md.nr_range + md.nr_le, md.nr_constraints)
    assert np.all(np.abs(ax[eq] - b[eq]) <= tol) and np.all(ax[le] <= b[le] + tol) and np.all(ax[ge] >= b[ge] - tol)
    assert np.all(x >= -tol) and np.all(x <= np.asarray(md.upper_bound) + tol)
    assert abs(float(np.dot(np.asarray(md.cost), x)) - plain.objective) <= 1e-7 * max(1.0, abs(plain.objective))
    # the slacks come back in the row's units: a <= row's slack is b - a x
    o_le = md.nr_normal + md.nr_range
    for k in range(md.nr_le):
        row = md.nr_eq + md.nr_range + k
        assert abs(back.get(o_le + k, 0.0) - (b[row] - ax[row])) <= tol
