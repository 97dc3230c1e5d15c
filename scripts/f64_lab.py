"""Numerics lab (not product code, not a checker): the reference's two-phase simplex rules in f64 on scipy's sparse LU,
to find out which f64 safeguards a given LP needs before they are built into the engines.  Fresh factors every K pivots
(product-form etas in between), optional recomputation of b and -pi at a refactorisation, absolute or relative pivot
tolerance.  usage: python scripts/f64_lab.py FILE.SIF [K=48] [refresh=1] [rel=0] [tol_pivot=1e-5] [max_iters]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lp_amd  # noqa: E402,F401
from rust_lp_amd import general_form, mps  # noqa: E402


def all_columns(md):
    """[normal | range slack | <= slack | >= slack | bound slack | range-bound slack] over rows
    [== | range | <= | >= | bound rows | range-bound rows] (matrix_data.rs:198-268, 308-348)."""
    ne, nr, nl, ng = md.nr_eq, md.nr_range, md.nr_le, md.nr_ge
    mc = ne + nr + nl + ng
    ub = np.asarray(md.upper_bound)
    bounded = np.nonzero(np.isfinite(ub))[0]
    nb = len(bounded)
    m = mc + nb + nr
    A = sp.csc_matrix((md.values, md.row_idx, md.col_ptr), shape=(mc, md.nr_normal))
    brow = sp.csc_matrix((np.ones(nb), (np.arange(nb), bounded)), shape=(nb, md.nr_normal))
    top = sp.vstack([A, brow, sp.csc_matrix((nr, md.nr_normal))]).tocsc()
    blocks = [top]
    def unit(rows, sign, rows2=None):
        k = len(rows)
        r = list(rows); c = list(range(k)); v = [sign] * k
        if rows2 is not None:
            r += list(rows2); c += list(range(k)); v += [1.0] * k
        return sp.csc_matrix((v, (r, c)), shape=(m, k))
    blocks.append(unit(range(ne, ne + nr), 1.0, range(mc + nb, mc + nb + nr)))
    blocks.append(unit(range(ne + nr, ne + nr + nl), 1.0))
    blocks.append(unit(range(ne + nr + nl, mc), -1.0))
    blocks.append(unit(range(mc, mc + nb), 1.0))
    blocks.append(unit(range(mc + nb, m), 1.0))
    M = sp.hstack(blocks).tocsc()
    rhs = np.concatenate([md.b, ub[bounded], md.ranges])
    cost = np.concatenate([md.cost, np.zeros(M.shape[1] - md.nr_normal)])
    # initial basis candidates: <= slacks, bound slacks, range-bound slacks (matrix_data.rs:432-452)
    n0 = md.nr_normal
    init = {}
    for k in range(nl):
        init[ne + nr + k] = n0 + nr + k
    for k in range(nb):
        init[mc + k] = n0 + nr + nl + ng + k
    for k in range(nr):
        init[mc + nb + k] = n0 + nr + nl + ng + nb + k
    return M, rhs, cost, init


class Lab:
    def __init__(self, md, K=48, refresh=True, rel=False, tol_pivot=1e-5, tol_cost=1e-7, tol_tie=1e-9, tol_zero=1e-11):
        self.M, self.rhs, self.cost, init = all_columns(md)
        self.m, self.nprov = self.M.shape
        art_rows = [i for i in range(self.m) if i not in init]
        self.na = len(art_rows)
        self.art_rows = art_rows
        E = sp.csc_matrix((np.ones(self.na), (art_rows, range(self.na))), shape=(self.m, self.na))
        self.C = sp.hstack([E, self.M]).tocsc()          # artificial columns first (partially.rs:72-80)
        self.CT = self.C.T.tocsr()
        self.n = self.C.shape[1]
        self.basis = np.zeros(self.m, dtype=np.int64)
        for k, r in enumerate(art_rows):
            self.basis[r] = k
        for r, j in init.items():
            self.basis[r] = self.na + j
        self.in_basis = np.zeros(self.n, dtype=bool)
        self.in_basis[self.basis] = True
        self.phase = 1
        self.K, self.refresh, self.rel = K, refresh, rel
        self.tp, self.tc, self.tt, self.tz = tol_pivot, tol_cost, tol_tie, tol_zero
        self.b = self.rhs.copy()
        self.c1 = np.concatenate([np.ones(self.na), np.zeros(self.nprov)])
        self.c2 = np.concatenate([np.zeros(self.na), self.cost])
        self.last = -1
        self.stable = int(os.environ.get("LAB_STABLE", "0"))
        self.delta = float(os.environ.get("LAB_DELTA", "1e-9"))
        self.iters = 0
        self.recent = []
        self.factor(first=True)

    def c(self):
        return self.c1 if self.phase == 1 else self.c2

    def factor(self, first=False):
        B = self.C[:, self.basis]
        try:
            self.lu = spla.splu(B.tocsc(), permc_spec="COLAMD", diag_pivot_thresh=0.1)
        except RuntimeError:
            print(f"singular basis at iteration {self.iters}; last pivots: {self.recent[-self.K:]}")
            raise
        self.etas = []
        if self.refresh or first:
            self.b = self.lu.solve(self.rhs)
            self.minus_pi = -self.lu.solve(self.c()[self.basis], trans="T")

    def ftran(self, a):
        x = self.lu.solve(a)
        for (r, eta) in self.etas:
            xr = x[r]
            if xr != 0.0:
                x += eta * xr
                x[r] = eta[r] * xr
        return x

    def btran_unit(self, r):
        v = np.zeros(self.m)
        v[r] = 1.0
        for (rr, eta) in reversed(self.etas):
            v[rr] = eta @ v
        return self.lu.solve(v, trans="T")

    def price(self):
        d = self.c() + self.CT @ self.minus_pi
        ok = (~self.in_basis) & (d < -self.tc)
        if self.phase == 2:
            ok[:self.na] = False
        if self.phase == 1:
            if not ok.any():
                self.last = -1
                return None
            start = self.last if self.last >= 0 else 0
            idx = np.nonzero(ok[start:])[0]
            j = start + idx[0] if len(idx) else int(np.nonzero(ok[:start])[0][0])
            self.last = int(j)
            return int(j), d[j]
        if not ok.any():
            return None
        dmin = d[ok].min()
        band = ok & (d <= dmin + self.tt * max(1.0, abs(dmin)))
        j = int(np.nonzero(band)[0][0])
        return j, d[j]

    def ratio(self, alpha):
        thr = self.tp * (max(1.0, np.abs(alpha).max()) if self.rel else 1.0)
        pos = alpha > thr
        if not pos.any():
            return None
        bb = np.where(self.b <= self.tz, 0.0, self.b)
        ratios = np.full(self.m, np.inf)
        ratios[pos] = bb[pos] / alpha[pos]
        mn = ratios.min()
        if self.stable == 2:                                   # Harris: relaxed minimum, then the largest pivot below it
            relaxed = np.full(self.m, np.inf)
            relaxed[pos] = (bb[pos] + self.delta) / alpha[pos]
            band = ratios <= relaxed.min()
        else:
            band = ratios <= mn + self.tt * max(1.0, abs(mn))
        cand = np.nonzero(band)[0]
        if self.stable:
            amax = alpha[cand].max()
            cand = cand[alpha[cand] >= amax * (1 - 1e-12)]
        return int(cand[np.argmin(self.basis[cand])])

    def pivot(self, q, dq, r, alpha):
        ar = alpha[r]
        br = self.b[r] / ar
        nz = alpha != 0
        self.b[nz] -= alpha[nz] * br
        self.b[r] = br
        eta = -alpha / ar
        eta[r] = 1.0 / ar
        self.etas.append((r, eta))
        leaving = self.basis[r]
        self.basis[r] = q
        self.in_basis[leaving] = False
        self.in_basis[q] = True
        rho = self.btran_unit(r)
        self.minus_pi -= dq * rho
        self.iters += 1
        self.recent.append((self.iters, q, r, float(ar), float(np.abs(alpha).max())))
        if len(self.etas) >= self.K:
            self.factor()

    def objective(self):
        return float(self.c()[self.basis] @ self.b)

    def run_phase(self, max_iters, log=2000):
        t0 = time.time()
        while self.iters < max_iters:
            pr = self.price()
            if pr is None:
                return "no_candidate"
            q, dq = pr
            alpha = self.ftran(np.asarray(self.C[:, q].todense()).ravel())
            alpha[np.abs(alpha) < 0] = 0
            r = self.ratio(alpha)
            if r is None:
                return "no_row"
            self.pivot(q, dq, r, alpha)
            if self.iters % log == 0:
                print(f"  it {self.iters} phase {self.phase} obj {self.objective():.10g} min b {self.b.min():.3g} "
                      f"{time.time() - t0:.0f}s", flush=True)
        return "limit"

    def finish_phase_one(self, textbook):
        removed = []
        arts = sorted(int(v) for v in self.basis if v < self.na)
        for a in arts:
            r = int(np.nonzero(self.basis == a)[0][0])
            rho = self.btran_unit(r)
            row = self.CT @ rho                              # tableau row r over all columns
            d = self.c() + self.CT @ self.minus_pi
            elig = (~self.in_basis) & (np.arange(self.n) >= self.na) & (np.abs(row) > self.tp)
            if not textbook:
                elig &= np.abs(d) <= self.tc
            idx = np.nonzero(elig)[0]
            if len(idx):
                q = int(idx[0])
                alpha = self.ftran(np.asarray(self.C[:, q].todense()).ravel())
                self.pivot(q, d[q], r, alpha)
            else:
                removed.append((a, r))
        return removed


def main():
    path = sys.argv[1]
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    refresh = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    rel = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    tol_pivot = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-5
    max_iters = int(sys.argv[6]) if len(sys.argv) > 6 else 200000
    m = mps.import_file(path, True)
    gf = general_form.GeneralForm.from_mps(m)
    ex = gf.derive_matrix_data_exact()
    md = gf.to_matrix_data(ex)
    lab = Lab(md, K=K, refresh=refresh, rel=rel, tol_pivot=tol_pivot)
    print(f"m {lab.m} n {lab.n} artificials {lab.na} K {K} refresh {refresh} rel {rel} tol_pivot {tol_pivot}")
    out = lab.run_phase(max_iters)
    print("phase 1:", out, lab.iters, "objective", lab.objective())
    if out != "no_candidate" or abs(lab.objective()) > 1e-6 * max(1.0, abs(lab.rhs).sum()):
        return
    removed = lab.finish_phase_one(textbook=True)
    print("artificials left after zero-level pivots (textbook rule):", removed)
    if removed:
        return
    lab.phase = 2
    lab.factor(first=True)
    out = lab.run_phase(max_iters)
    print("phase 2:", out, lab.iters, "objective", lab.objective() + float(gf.fixed_cost), "min b", lab.b.min())


if __name__ == "__main__":
    main()
