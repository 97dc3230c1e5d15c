"""Generates tests/golden/corpus/: every LP of the reference's Netlib directory (/root/reference/tests/netlib/problem_files,
104 entries of which the reference's tests touch 21) as the standardised `MatrixData` the pivot engine is handed -- the build's
own MPS reader (rust-lp_amd/mps.py, fixed format like tests/netlib/mod.rs:54), presolve and standardisation
(rust-lp_amd/general_form.py) run HERE, in the build container, once; the GPU tier loads the arrays.  A fixture is data: the
CSC arrays, right-hand sides, costs, bounds, the objective's fixed part, and -- as an independent check that is NOT the
reference ("parity unpinned": the reference holds no value for these files) -- the optimum HiGHS (scipy.optimize.linprog)
finds for that very standardised LP.

Files the reader rejects (as the reference's own reader would: tests never ran them) are listed with the message.

usage: python scripts/gen_corpus_fixture.py [--jobs 4] [NAME ...]        (about 20 minutes)"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "/root/reference/tests/netlib/problem_files"
OUT = os.path.join(ROOT, "tests", "golden", "corpus")
# tests/netlib/test.rs: the pins the reference holds (value, tolerance, ignored?)
REFERENCE_PINS = {
    "ADLITTLE": (2.254949632e5, 1e-3), "AFIRO": (-464.75314, 1e-5), "SC50A": (-64.575077, 1e-5), "SC50B": (-70.0, 1e-10),
    "KB2": (-1.74990012991e3, 1e-5), "SC105": (-52.202061, 1e-5), "STOCFOR1": (-41131.976, 1e-3), "BLEND": (-30.812150, 1e-5),
    "SCAGR7": (-2331389.8, 1e-1), "SC205": (-52.202061, 1e-5), "SHARE2B": (-4.1573224074e2, 1e-5), "RECIPELP": (-266.616, 1e-3),
    "LOTFI": (-25.264706, 1e-5), "VTP-BASE": (1.2983146246e5, 1e-4), "SHARE1B": (-7.6589318579e4, 1e-3),
    "BOEING2": (-3.1501872802e2, 1e-3), "BORE3D": (1.3730803942e3, 1e-2), "SCORPION": (1.8781248227e3, 1e-2),
    "GREENBEA": (-7.2555248130e7, 1e0), "GREENBEB": (-4.3022602612e6, 1e1), "25FV47": (5.5018459e3, 1e-4),
    "80BAU3B": (9.872241924e5, 1e-4),
}


def highs(md):
    """HiGHS on the standardised LP: min c'x, rows [== | range | <= | >=], 0 <= x <= ub."""
    from scipy.optimize import linprog
    from scipy.sparse import csc_matrix, vstack
    mc = md.nr_constraints
    A = csc_matrix((md.values, md.row_idx, md.col_ptr), shape=(mc, md.nr_normal)).tocsr()
    ne, nr, nl = md.nr_eq, md.nr_range, md.nr_le
    lo = ne + nr + nl
    ub_rows = [A[ne:ne + nr], -A[ne:ne + nr], A[ne + nr:lo], -A[lo:]]
    ub_rhs = [md.b[ne:ne + nr], -(md.b[ne:ne + nr] - md.ranges), md.b[ne + nr:lo], -md.b[lo:mc]]
    A_ub = vstack([r for r in ub_rows if r.shape[0]]) if any(r.shape[0] for r in ub_rows) else None
    b_ub = np.concatenate([r for r in ub_rhs if len(r)]) if A_ub is not None else None
    res = linprog(md.cost, A_ub=A_ub, b_ub=b_ub, A_eq=A[:ne] if ne else None, b_eq=md.b[:ne] if ne else None,
                  bounds=[(0, None if not np.isfinite(u) else u) for u in md.upper_bound], method="highs")
    return int(res.status), (float(res.fun) if res.status == 0 else None)


def one(name):
    import rust_lp_amd  # noqa: F401
    from rust_lp_amd import general_form, mps
    t0 = time.time()
    rec = {"name": name}
    try:
        m = mps.import_file(os.path.join(SRC, name + ".SIF"), True)
    except Exception as e:      # noqa: BLE001  (the reader's own rejection is the finding)
        rec["reader_error"] = f"{type(e).__name__}: {e}"[:300]
        return rec
    try:
        gf = general_form.GeneralForm.from_mps(m)
        ex = gf.derive_matrix_data_exact()
        md = gf.to_matrix_data(ex)
    except Exception as e:      # noqa: BLE001
        rec["standardise_error"] = f"{type(e).__name__}: {e}"[:300]
        return rec
    rec.update(nr_normal=int(md.nr_normal), nr_eq=int(md.nr_eq), nr_range=int(md.nr_range), nr_le=int(md.nr_le), nr_ge=int(md.nr_ge),
               nr_rows=int(md.nr_rows), nr_columns=int(md.nr_columns), nnz=int(len(md.values)), fixed_cost=float(gf.fixed_cost),
               front_end_seconds=round(time.time() - t0, 1))
    try:
        st, fun = highs(md)
        rec["highs_status"] = st
        rec["highs_objective"] = None if fun is None else fun + float(gf.fixed_cost)
    except Exception as e:      # noqa: BLE001
        rec["highs_error"] = f"{type(e).__name__}: {e}"[:200]
    if name in REFERENCE_PINS:
        rec["reference_pin"], rec["reference_tolerance"] = REFERENCE_PINS[name]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), counts=np.array([md.nr_normal, md.nr_eq, md.nr_range, md.nr_le, md.nr_ge], dtype=np.int64),
                        col_ptr=md.col_ptr, row_idx=md.row_idx, values=md.values, b=md.b, ranges=md.ranges, cost=md.cost,
                        upper_bound=md.upper_bound, fixed_cost=np.array([float(gf.fixed_cost)]))
    rec["seconds"] = round(time.time() - t0, 1)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    names = args.names or sorted(f[:-4] for f in os.listdir(SRC) if f.endswith(".SIF"))
    # largest files last would leave one worker alone at the end: largest first
    names.sort(key=lambda n: -os.path.getsize(os.path.join(SRC, n + ".SIF")))
    index_path = os.path.join(OUT, "index.json")
    index = {}
    if os.path.exists(index_path) and args.names:
        index = {r["name"]: r for r in json.load(open(index_path))}
    with mp.Pool(args.jobs, maxtasksperchild=1) as pool:
        for rec in pool.imap_unordered(one, names):
            index[rec["name"]] = rec
            print(json.dumps(rec), flush=True)
            json.dump(sorted(index.values(), key=lambda r: r["name"]), open(index_path, "w"), indent=1)


if __name__ == "__main__":
    main()
