#!/usr/bin/env python3
"""numpy prototype of the device-side Forrest-Tomlin scheme of the LU engine (rust-lp_amd/csrc/relp_kernels_ft.hip).

Design check, not product code: the factors L, U0 of the last refactorisation are NEVER modified (their level
schedules stay valid); an update moves pivot p to the back of the pivot order by
  * masking row p / column p of U0 (x[p] reads as 0 in the U0 sweeps, row p writes 0),
  * a new "tail slot" t: spike column (sparse part over original pivots + dense part T[:, t] over tail slots),
    row eta r_t (sparse part over original pivots + dense part E[t, :] over tail slots),
which is the reference's update (lower_upper/mod.rs:92-155: r = u_bar U^-1, R = I - e_p r', spike into column p,
rotate p to the back) in coordinates that need no physical rotation.  This file checks FTRAN / BTRAN / update against
a dense inverse over random replacement sequences, repeated replacements of one position included.

  python scripts/ft_prototype.py
"""
import numpy as np


class FT:
    def __init__(self, B, tcap=16):
        m = B.shape[0]
        self.m, self.tcap = m, tcap
        # P B Q = L U with partial pivoting on rows only (prototype): rowperm[k] = original row of step k
        import scipy.linalg as sla
        P, L, U = sla.lu(B)                      # B = P L U  ->  P' B = L U
        self.rowperm = np.argmax(P, axis=0)      # (P' B)[k, :] = B[rowperm[k], :]
        assert np.allclose(B[self.rowperm, :], L @ U)
        self.colperm = np.arange(m)              # Q = I in the prototype
        self.L, self.U0 = L, U
        self.t = 0
        self.slot_pivot = np.full(tcap, -1)
        self.live = np.zeros(tcap, bool)
        self.prev = np.full(tcap, -1)            # previous slot of the same pivot
        self.tslot = np.full(m, -1)              # live slot of a pivot, -1 = original
        self.spike_list = [dict() for _ in range(tcap)]   # sparse over pivots that were original at the time
        self.eta_list = [dict() for _ in range(tcap)]
        self.T = np.zeros((tcap, tcap))          # T[s', s]: spike s in the row of tail slot s' (diag included)
        self.E = np.zeros((tcap, tcap))          # E[s, s']: eta s on tail slot s'
        self.dead_diag = np.zeros(m, bool)       # rows of U0 whose diagonal is patched to 0

    # ---- U0 sweeps with masking -------------------------------------------------------------------
    def _u0_forward(self, x):
        """Solve U0 z = x over original pivots (tail pivots masked: read as 0, write 0)."""
        m = self.m
        for k in range(m - 1, -1, -1):
            s = x[k] - self.U0[k, k + 1:] @ x[k + 1:]
            x[k] = 0.0 if self.dead_diag[k] else s / self.U0[k, k]
        return x

    def _u0_transposed(self, x):
        """Solve y' U0 = x' over original pivots (masked the same way)."""
        m = self.m
        for k in range(m):
            s = x[k] - self.U0[:k, k] @ x[:k]
            x[k] = 0.0 if self.dead_diag[k] else s / self.U0[k, k]
        return x

    # ---- eta file ----------------------------------------------------------------------------------
    def _etas_forward(self, x):
        t = self.t
        dots = np.zeros(t)
        for s in range(t):                                   # phase A: sparse parts against the pre-chain x
            dots[s] = sum(v * x[j] for j, v in self.eta_list[s].items())
        val = np.zeros(t)
        for s in range(t):                                   # phase B: the chain over slots
            base = val[self.prev[s]] if self.prev[s] >= 0 else x[self.slot_pivot[s]]
            val[s] = base - dots[s] - self.E[s, :s] @ val[:s]
        for s in range(t):
            if self.live[s]:
                x[self.slot_pivot[s]] = val[s]
        return x

    def _etas_reverse(self, v):
        t = self.t
        u = np.zeros(t)
        nxt = np.full(t, -1)
        for s in range(t):
            if self.prev[s] >= 0:
                nxt[self.prev[s]] = s
        for s in range(t - 1, -1, -1):
            base = u[nxt[s]] if nxt[s] >= 0 else v[self.slot_pivot[s]]
            u[s] = base - self.E[s + 1:t, s] @ u[s + 1:t]
        # tail pivots: value after every eta that saw them as a tail pivot = u at their FIRST slot
        for s in range(t):
            if self.prev[s] < 0:
                v[self.slot_pivot[s]] = u[s]
        for s in range(t - 1, -1, -1):                       # sparse parts (order irrelevant mathematically)
            for j, val in self.eta_list[s].items():
                v[j] -= val * u[s]
        return v

    # ---- FTRAN / BTRAN -----------------------------------------------------------------------------
    def ftran(self, a):
        x = a[self.rowperm].astype(float)
        x = np.linalg.solve(self.L, x)
        x = self._etas_forward(x)
        spike = x.copy()
        t = self.t
        z = np.zeros(t)
        for s in range(t - 1, -1, -1):                       # tail solve
            if self.live[s]:
                z[s] = (x[self.slot_pivot[s]] - sum(self.T[s, s2] * z[s2] for s2 in range(s + 1, t) if self.live[s2])) / self.T[s, s]
        for s in range(t):                                   # push the spikes into the original rows
            if self.live[s]:
                for k, v in self.spike_list[s].items():
                    if self.tslot[k] < 0:
                        x[k] -= v * z[s]
        for s in range(t):
            if self.live[s]:
                x[self.slot_pivot[s]] = 0.0
        x = self._u0_forward(x)
        for s in range(t):
            if self.live[s]:
                x[self.slot_pivot[s]] = z[s]
        alpha = np.zeros(self.m)
        alpha[self.colperm] = x
        return alpha, spike

    def _ut_solve(self, c, first_slot=0):
        """y' U = c' for the CURRENT U; c indexed by pivot.  Returns y (pivot-indexed) and its tail part."""
        t = self.t
        y = c.astype(float).copy()
        ct = np.array([y[self.slot_pivot[s]] if self.live[s] else 0.0 for s in range(t)])
        for s in range(t):
            if self.live[s]:
                y[self.slot_pivot[s]] = 0.0
        y = self._u0_transposed(y)
        yt = np.zeros(t)
        for s in range(t):
            if self.live[s]:
                dot = sum(v * y[k] for k, v in self.spike_list[s].items() if self.tslot[k] < 0)
                yt[s] = (ct[s] - dot - sum(self.T[s2, s] * yt[s2] for s2 in range(s) if self.live[s2])) / self.T[s, s]
        return y, yt

    def btran(self, c):
        """z' B = c' with c indexed by basis position."""
        y, yt = self._ut_solve(c[self.colperm])
        for s in range(self.t):
            if self.live[s]:
                y[self.slot_pivot[s]] = yt[s]
        y = self._etas_reverse(y)
        w = np.linalg.solve(self.L.T, y)
        z = np.zeros(self.m)
        z[self.rowperm] = w
        return z

    # ---- update -------------------------------------------------------------------------------------
    def update(self, r, spike):
        """Basis position r is replaced by the column whose spike (ftran's second result) is given."""
        m, t = self.m, self.t
        assert t < self.tcap
        p = int(np.nonzero(self.colperm == r)[0][0])
        # u_bar = row p of the current U right of the diagonal
        ubar = np.zeros(m)
        sp_old = self.tslot[p]
        if sp_old < 0:
            for l in range(p + 1, m):
                if self.tslot[l] < 0:
                    ubar[l] = self.U0[p, l]
            ubar_t = np.array([self.spike_list[s].get(p, 0.0) if self.live[s] else 0.0 for s in range(t)])
        else:
            ubar_t = np.array([self.T[sp_old, s] if (self.live[s] and s > sp_old) else 0.0 for s in range(t)])
        # r' = u_bar' U^-1 (support: positions after p)
        c = ubar.copy()
        for s in range(t):
            if self.live[s]:
                c[self.slot_pivot[s]] = ubar_t[s]
        if sp_old >= 0:
            c[p] = 0.0
        y, yt = self._ut_solve(c)
        if sp_old >= 0:
            assert abs(yt[sp_old]) < 1e-12 and not np.any(np.abs(yt[:sp_old]) > 1e-12)
        # delete row p / column p from U
        if sp_old < 0:
            self.dead_diag[p] = True
        else:
            self.live[sp_old] = False
            self.T[sp_old, :] = 0.0
            self.T[:, sp_old] = 0.0
            self.T[sp_old, sp_old] = 1.0
        # new slot
        self.slot_pivot[t] = p
        self.prev[t] = sp_old
        self.eta_list[t] = {int(j): float(y[j]) for j in np.nonzero(y)[0] if self.tslot[j] < 0 and j != p}
        self.E[t, :t] = [yt[s] if self.live[s] else 0.0 for s in range(t)]
        diag = spike[p]
        diag -= sum(v * spike[j] for j, v in self.eta_list[t].items())
        diag -= sum(self.E[t, s] * spike[self.slot_pivot[s]] for s in range(t) if self.live[s])
        self.spike_list[t] = {int(k): float(spike[k]) for k in np.nonzero(spike)[0] if self.tslot[k] < 0 and k != p}
        for s in range(t):
            if self.live[s]:
                self.T[s, t] = spike[self.slot_pivot[s]]
        self.T[t, t] = diag
        self.live[t] = True
        self.tslot[p] = t
        self.t = t + 1


def main():
    rng = np.random.default_rng(5)
    worst = 0.0
    for trial in range(40):
        m = int(rng.integers(3, 14))
        B = rng.normal(size=(m, m)) * (rng.random((m, m)) < 0.6) + np.eye(m) * 3
        ft = FT(B.copy(), tcap=12)
        cur = B.copy()
        for step in range(12):
            a = rng.normal(size=m) * (rng.random(m) < 0.7)
            if rng.random() < 0.4 and step > 0:
                r = last_r                                   # replace the same position again
            else:
                r = int(rng.integers(0, m))
            alpha, spike = ft.ftran(a)
            ref = np.linalg.solve(cur, a)
            worst = max(worst, np.max(np.abs(alpha - ref)))
            assert np.allclose(alpha, ref, atol=1e-8), (trial, step, "ftran")
            if abs(alpha[r]) < 1e-3:
                continue
            ft.update(r, spike)
            cur[:, r] = a
            last_r = r
            inv = np.linalg.inv(cur)
            for i in range(m):
                e = np.zeros(m); e[i] = 1.0
                z = ft.btran(e)
                worst = max(worst, np.max(np.abs(z - inv[i])))
                assert np.allclose(z, inv[i], atol=1e-7), (trial, step, "btran", i)
            c = rng.normal(size=m)
            assert np.allclose(ft.btran(c), c @ inv, atol=1e-7)
    print("ft prototype ok, worst error", worst)


if __name__ == "__main__":
    main()
