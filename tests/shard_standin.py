"""numpy stand-in for the `relp_shard_*` entry points (test infrastructure): the same shard
protocol as rust-lp_amd/csrc (message layouts, padding, tie rules) computed on the CPU so that the
multi-rank orchestration in rust_lp_amd.sharded can be exercised with the gloo backend."""
import numpy as np

RUNNING, OPTIMAL, UNBOUNDED, PHASE_ONE_DONE = 0, 1, 2, 4


class NumpyShardOps:
    def __init__(self, rank, world, m, n, A_local, col_lo, b, c, tol_cost=1e-7, tol_pivot=1e-5, tol_zero=1e-11,
                 tol_tie=1e-9):
        self.rank, self.world, self.m, self.n = rank, world, m, n
        self.A, self.col_lo, self.col_hi = A_local, col_lo, col_lo + A_local.shape[1]
        self.c = c
        self.tol = (tol_cost, tol_pivot, tol_zero, tol_tie)
        stride = -(-m // world)
        stride += stride % 2
        self.row_stride = stride
        self.row_lo = min(m, rank * stride)
        self.row_hi = min(m, self.row_lo + stride)
        self.candidate_len = 3 + m + ((3 + m) % 2)
        self.rho_len = -(-m // 16) * 16
        self.Binv = np.eye(m)[self.row_lo:self.row_hi].copy()
        self.b = b.astype(np.float64).copy()
        self.minus_pi = np.zeros(m)
        self.basis = np.arange(n, n + m)
        self.in_basis = np.zeros(n + m, dtype=bool)
        self.in_basis[self.basis] = True
        self.minus_obj = 0.0
        self.phase, self.outcome, self.iterations = 1, RUNNING, 0
        self.trace = []
        self.alpha = np.zeros(m)

    def set_stream(self, _):
        pass

    # ---- PRICE -------------------------------------------------------------------------------
    def price(self, cand):
        out = cand.numpy()
        out[:] = 0.0
        out[0] = np.inf
        if self.outcome != RUNNING:
            return
        d = np.full(self.n + self.m, np.inf)
        cost = self.c if self.phase == 2 else np.zeros(self.n)
        d[self.col_lo:self.col_hi] = cost[self.col_lo:self.col_hi] + self.minus_pi @ self.A
        d[self.n:] = self.minus_pi                      # +1 slack columns, zero cost
        ok = (~self.in_basis) & (d < -self.tol[0])
        if ok.any():
            dmin = d[ok].min()
            band = ok & (d <= dmin + self.tol[3] * max(1.0, abs(dmin)))      # Dantzig tie band, lowest index
            j = int(np.nonzero(band)[0][0])
            aq = np.zeros(self.m)
            if j < self.n:
                aq[:] = self.A[:, j - self.col_lo]
            else:
                aq[j - self.n] = 1.0
            out[0], out[1], out[2] = dmin, j, d[j]
            out[3:3 + self.m] = aq

    def select_column(self, cands, count):
        if self.outcome != RUNNING:
            return
        msgs = cands.numpy().reshape(count, self.candidate_len)
        best = None
        for g in range(count):
            if np.isfinite(msgs[g, 0]) and (best is None or (msgs[g, 0], msgs[g, 1]) < (msgs[best, 0], msgs[best, 1])):
                best = g
        if best is not None:
            bound = msgs[best, 0] + self.tol[3] * max(1.0, abs(msgs[best, 0]))
            for g in range(count):
                if msgs[g, 0] <= bound and msgs[g, 1] < msgs[best, 1]:
                    best = g
        if best is None:
            self.outcome = 1          # no candidate
            return
        self.q, self.d_q = int(msgs[best, 1]), float(msgs[best, 2])
        self.aq = msgs[best, 3:3 + self.m].copy()

    # ---- FTRAN / RATIO -----------------------------------------------------------------------
    def ftran(self, alpha_slice):
        out = alpha_slice.numpy()
        out[:] = 0.0
        if self.outcome != RUNNING:
            return
        out[:self.row_hi - self.row_lo] = self.Binv @ self.aq

    def ratio(self, slices, count, rho):
        rh = rho.numpy()
        rh[:] = 0.0
        if self.outcome != RUNNING:
            return
        self.alpha = slices.numpy()[:count * self.row_stride][:self.m].copy()
        tc, tp, tz, tt = self.tol
        pos = self.alpha > tp
        if not pos.any():
            self.outcome = 2
            return
        bb = np.where(self.b <= tz, 0.0, self.b)
        ratios = np.where(pos, bb / np.where(pos, self.alpha, 1.0), np.inf)
        mn = ratios.min()
        tie = pos & (ratios <= mn + tt * max(1.0, abs(mn)))
        rows = np.nonzero(tie)[0]
        r = int(rows[np.argmin(self.basis[rows])])
        self.r, self.alpha_r, self.b_r = r, float(self.alpha[r]), float(self.b[r])
        if self.row_lo <= r < self.row_hi:
            rh[:self.m] = self.Binv[r - self.row_lo] / self.alpha_r

    # ---- UPDATE ------------------------------------------------------------------------------
    def update(self, rho):
        if self.outcome != RUNNING:
            return
        rh = rho.numpy()[:self.m]
        r, br = self.r, self.b_r / self.alpha_r
        self.minus_pi -= self.d_q * rh
        nb = self.b - self.alpha * br
        nb[r] = br
        self.b = nb
        self.minus_obj -= self.d_q * br
        leaving = int(self.basis[r])
        self.basis[r] = self.q
        self.in_basis[leaving] = False
        self.in_basis[self.q] = True
        a_loc = self.alpha[self.row_lo:self.row_hi]
        new = self.Binv - np.outer(a_loc, rh)
        if self.row_lo <= r < self.row_hi:
            new[r - self.row_lo] = rh
        self.Binv = new
        self.trace.append((self.phase, self.q, r, leaving))
        self.iterations += 1

    def poll(self):
        if self.outcome == 1:
            if self.phase == 1:
                self.phase, self.outcome = 2, RUNNING
                return PHASE_ONE_DONE, self.iterations
            return OPTIMAL, self.iterations
        if self.outcome == 2:
            return UNBOUNDED, self.iterations
        return RUNNING, self.iterations


class NumpyTableauShardOps:
    """Stand-in for the tableau engine's shard protocol: stored columns [structural | slack] split
    contiguously over the ranks, one candidate message [key, j, d_j, alpha(m)] per pivot, everything
    else local (relp_shard_price / relp_shard_select_column / relp_shard_pivot)."""
    tableau = True
    update_block = 0

    def __init__(self, rank, world, m, n, A_full, b, c, tol_cost=1e-7, tol_pivot=1e-5, tol_zero=1e-11, tol_tie=1e-9):
        self.m, self.n = m, n
        n_store = n + m
        per = -(-n_store // world)
        per += per % 2
        self.c_lo, self.c_hi = min(n_store, rank * per), min(n_store, rank * per + per)
        full = np.hstack([A_full, np.eye(m)])
        self.T = full[:, self.c_lo:self.c_hi].copy()                     # owned columns of the tableau
        self.d = np.zeros(self.c_hi - self.c_lo)                         # phase 1: all zero (slack basis)
        self.cost = np.concatenate([c, np.zeros(m)])[self.c_lo:self.c_hi]
        self.tol = (tol_cost, tol_pivot, tol_zero, tol_tie)
        self.row_stride, self.rho_len = 2, 2
        self.candidate_len = 3 + m + ((3 + m) % 2)
        self.b = b.astype(np.float64).copy()
        self.basis = np.arange(n, n + m)
        self.in_basis = np.zeros(n_store, dtype=bool)
        self.in_basis[self.basis] = True
        self.minus_obj = 0.0
        self.phase, self.outcome, self.iterations = 1, RUNNING, 0
        self.trace = []

    def set_stream(self, _):
        pass

    def price(self, cand):
        out = cand.numpy()
        out[:] = 0.0
        out[0] = np.inf
        if self.outcome != RUNNING:
            return
        idx = np.arange(self.c_lo, self.c_hi)
        ok = (~self.in_basis[idx]) & (self.d < -self.tol[0])
        if ok.any():
            dmin = self.d[ok].min()
            band = ok & (self.d <= dmin + self.tol[3] * max(1.0, abs(dmin)))
            k = int(np.nonzero(band)[0][0])
            out[0], out[1], out[2] = dmin, idx[k], self.d[k]
            out[3:3 + self.m] = self.T[:, k]

    def select_column(self, cands, count):
        if self.outcome != RUNNING:
            return
        msgs = cands.numpy().reshape(count, self.candidate_len)
        best = None
        for g in range(count):
            if np.isfinite(msgs[g, 0]) and (best is None or (msgs[g, 0], msgs[g, 1]) < (msgs[best, 0], msgs[best, 1])):
                best = g
        if best is None:
            self.outcome = 1
            return
        bound = msgs[best, 0] + self.tol[3] * max(1.0, abs(msgs[best, 0]))
        for g in range(count):
            if msgs[g, 0] <= bound and msgs[g, 1] < msgs[best, 1]:
                best = g
        self.q, self.d_q = int(msgs[best, 1]), float(msgs[best, 2])
        self.alpha = msgs[best, 3:3 + self.m].copy()

    def pivot(self):
        if self.outcome != RUNNING:
            return
        tc, tp, tz, tt = self.tol
        pos = self.alpha > tp
        if not pos.any():
            self.outcome = 2
            return
        bb = np.where(self.b <= tz, 0.0, self.b)
        ratios = np.where(pos, bb / np.where(pos, self.alpha, 1.0), np.inf)
        mn = ratios.min()
        rows = np.nonzero(pos & (ratios <= mn + tt * max(1.0, abs(mn))))[0]
        r = int(rows[np.argmin(self.basis[rows])])
        ar = self.alpha[r]
        row = self.T[r].copy()
        self.d -= (self.d_q / ar) * row
        if self.c_lo <= self.q < self.c_hi:
            self.d[self.q - self.c_lo] = 0.0
        newT = self.T - np.outer(self.alpha, row / ar)
        newT[r] = row / ar
        self.T = newT
        br = self.b[r] / ar
        nb = self.b - self.alpha * br
        nb[r] = br
        self.b = nb
        self.minus_obj -= self.d_q * br
        leaving = int(self.basis[r])
        self.basis[r] = self.q
        self.in_basis[leaving] = False
        self.in_basis[self.q] = True
        self.trace.append((self.phase, self.q, r, leaving))
        self.iterations += 1

    def poll(self):
        if self.outcome == 1:
            if self.phase == 1:
                self.phase, self.outcome = 2, RUNNING
                self.d = self.cost.copy()            # slack basis: c_B = 0, so d = c
                return PHASE_ONE_DONE, self.iterations
            return OPTIMAL, self.iterations
        if self.outcome == 2:
            return UNBOUNDED, self.iterations
        return RUNNING, self.iterations
