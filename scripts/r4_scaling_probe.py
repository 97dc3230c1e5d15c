"""Probe: geometric scaling by powers of two (rows, then columns, four sweeps) in front of the engines, on the files of the
reference's Netlib directory that no leg of engine.solve_verified solves."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa
from rust_lp_amd import engine
from rust_lp_amd.matrix_data import MatrixData
import corpus


def scaled(md, sweeps=4):
    md = md.ensure_csc()
    m, n = md.nr_constraints, md.nr_normal
    cols = np.repeat(np.arange(n), np.diff(md.col_ptr))
    rows = np.asarray(md.row_idx)
    a = np.abs(np.asarray(md.values, dtype=float))
    r, s = np.ones(m), np.ones(n)
    nz = a > 0
    for _ in range(sweeps):
        v = a * r[rows] * s[cols]
        lo = np.full(m, np.inf); hi = np.zeros(m)
        np.minimum.at(lo, rows[nz], v[nz]); np.maximum.at(hi, rows[nz], v[nz])
        ok = hi > 0
        r[ok] /= np.sqrt(lo[ok] * hi[ok])
        v = a * r[rows] * s[cols]
        lo = np.full(n, np.inf); hi = np.zeros(n)
        np.minimum.at(lo, cols[nz], v[nz]); np.maximum.at(hi, cols[nz], v[nz])
        ok = hi > 0
        s[ok] /= np.sqrt(lo[ok] * hi[ok])
    r = 2.0 ** np.round(np.log2(r)); s = 2.0 ** np.round(np.log2(s))
    out = MatrixData(md.nr_normal, md.nr_eq, md.nr_range, md.nr_le, md.nr_ge, np.asarray(md.b) * r, np.asarray(md.cost) * s,
                     np.asarray(md.upper_bound) / s, np.asarray(md.ranges) * r[md.nr_eq:md.nr_eq + md.nr_range] if md.nr_range else np.zeros(0),
                     md.col_ptr, md.row_idx, np.asarray(md.values) * r[rows] * s[cols])
    return out, r, s


names = sys.argv[1:] or ["TUFF", "PEROLD", "PILOT-JA", "PILOTNOV", "DEGEN3", "CYCLE", "PILOT87", "DFL001", "PILOT4", "BNL1", "MAROS", "MODSZK1"]
idx = corpus.index()
for name in names:
    md, fixed = corpus.load(name)
    smd, r, s = scaled(md)
    a0 = np.abs(np.asarray(md.values)); a1 = np.abs(np.asarray(smd.values))
    t0 = time.perf_counter()
    oc, t, report = engine.solve_verified(smd, seconds_per_leg=25.0)
    obj = t.objective_function_value() + fixed if t is not None and oc == engine.OPTIMAL else None
    if t is not None:
        t.close()
    want = idx[name]["highs_objective"]
    ok = obj is not None and abs(obj - want) <= 1e-6 * max(1.0, abs(want))
    print(name, f"|a| {a0[a0>0].min():.1e}..{a0.max():.1e} -> {a1[a1>0].min():.1e}..{a1.max():.1e}", engine.OUTCOME_NAMES.get(oc, oc), report["verified"], obj, want, "OK" if ok else "--",
          [(l["config"], l["engine"], l.get("outcome"), l.get("pivots")) for l in report["legs"]], f"{time.perf_counter()-t0:.1f}s", flush=True)
