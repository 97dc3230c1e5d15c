"""Host-side problem container mirroring the reference's ``MatrixData``
(/root/reference/src/algorithm/two_phase/matrix_provider/matrix_data.rs:54-90): a column-major
constraint matrix over rows ordered [== | range | <= | >=], the right-hand side, range widths,
costs and optional variable upper bounds.  Slack / bound-slack columns and bound rows stay
virtual (matrix_data.rs:37-52); the engine materialises them on the device.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np


@dataclass
class MatrixData:
    nr_normal: int
    nr_eq: int
    nr_range: int
    nr_le: int
    nr_ge: int
    b: np.ndarray
    cost: np.ndarray
    upper_bound: np.ndarray                      # +inf = no upper bound
    ranges: np.ndarray = field(default_factory=lambda: np.zeros(0))
    # CSC storage (always available on the host unless the matrix lives only on the device)
    col_ptr: Optional[np.ndarray] = None
    row_idx: Optional[np.ndarray] = None
    values: Optional[np.ndarray] = None
    # dense column-major storage (nr_constraints x nr_normal); may replace CSC
    dense: Optional[np.ndarray] = None

    @property
    def nr_constraints(self) -> int:
        return self.nr_eq + self.nr_range + self.nr_le + self.nr_ge

    @property
    def nr_bounds(self) -> int:
        return int(np.isfinite(self.upper_bound).sum())

    @property
    def nr_rows(self) -> int:
        """matrix_data.rs nr_rows = constraints + variable bounds + range-slack bounds."""
        return self.nr_constraints + self.nr_bounds + self.nr_range

    @property
    def nr_columns(self) -> int:
        """matrix_data.rs:403-409."""
        return self.nr_normal + self.nr_range + self.nr_le + self.nr_ge + self.nr_bounds + self.nr_range

    def ensure_csc(self) -> "MatrixData":
        if self.col_ptr is None:
            a = np.asarray(self.dense)
            col_ptr = [0]
            rows, vals = [], []
            for j in range(a.shape[1]):
                nz = np.nonzero(a[:, j])[0]
                rows.append(nz.astype(np.int32))
                vals.append(a[nz, j])
                col_ptr.append(col_ptr[-1] + len(nz))
            self.col_ptr = np.array(col_ptr, dtype=np.int64)
            self.row_idx = np.concatenate(rows) if rows else np.zeros(0, np.int32)
            self.values = np.concatenate(vals) if vals else np.zeros(0)
        return self

    def ensure_dense(self) -> "MatrixData":
        if self.dense is None:
            a = np.zeros((self.nr_constraints, self.nr_normal), order="F")
            for j in range(self.nr_normal):
                s, t = self.col_ptr[j], self.col_ptr[j + 1]
                a[self.row_idx[s:t], j] = self.values[s:t]
            self.dense = a
        return self

    # ---- beyond the reference: scaling in front of the engines (the reference solves the data as read) ---------------------------
    def scaled(self, sweeps: int = 4):
        """Geometric scaling: rows, then columns, by 1 / sqrt(min |a| * max |a|) of the line, `sweeps` times, every factor rounded to
        a power of two (so scaling and unscaling are exact in f64).  Returns (scaled MatrixData, row factors r, column factors s) with
        A' = R A S, b' = R b, ranges' = R ranges, c' = S c, upper bounds' = ub / s: x = S x', the objective value is unchanged.
        On the reference's Netlib directory this is what the PILOT family needs (|a| from 1e-5 to 6e6 -> 1e-2 to 8e1): PEROLD,
        PILOT-JA and PILOTNOV, unsolved by every engine and configuration on the data as read, solve on the first or second leg of
        engine.solve_verified (profiles/r04_corpus_verified.md)."""
        md = self.ensure_csc()
        m, n = md.nr_constraints, md.nr_normal
        cols = np.repeat(np.arange(n), np.diff(np.asarray(md.col_ptr)))
        rows = np.asarray(md.row_idx, dtype=np.int64)
        a = np.abs(np.asarray(md.values, dtype=np.float64))
        nz = a > 0
        r, s = np.ones(m), np.ones(n)
        for _ in range(sweeps):
            for line, count, other_is_col in ((rows, m, True), (cols, n, False)):
                v = a * r[rows] * s[cols]
                lo, hi = np.full(count, np.inf), np.zeros(count)
                np.minimum.at(lo, line[nz], v[nz])
                np.maximum.at(hi, line[nz], v[nz])
                ok = hi > 0
                f = np.ones(count)
                f[ok] = 1.0 / np.sqrt(lo[ok] * hi[ok])
                if other_is_col:
                    r *= f
                else:
                    s *= f
        r = 2.0 ** np.round(np.log2(r))
        s = 2.0 ** np.round(np.log2(s))
        ranges = np.asarray(md.ranges, dtype=np.float64) * r[md.nr_eq:md.nr_eq + md.nr_range] if md.nr_range else np.zeros(0)
        out = MatrixData(md.nr_normal, md.nr_eq, md.nr_range, md.nr_le, md.nr_ge, np.asarray(md.b, dtype=np.float64) * r,
                         np.asarray(md.cost, dtype=np.float64) * s, np.asarray(md.upper_bound, dtype=np.float64) / s, ranges,
                         np.asarray(md.col_ptr), np.asarray(md.row_idx), np.asarray(md.values, dtype=np.float64) * r[rows] * s[cols])
        return out, r, s

    def unscale_bfs(self, bfs, r: np.ndarray, s: np.ndarray):
        """A basic feasible solution [(column, value)] of `self.scaled()` in the units of `self`: structural columns and the slacks
        of their bounds times s, the slacks of a row (range, <=, >=, range bound) divided by the row's r (columns as in
        matrix_data.rs:403-409: normal | range slacks | <= slacks | >= slacks | bound slacks | range-bound slacks)."""
        bounded = np.nonzero(np.isfinite(np.asarray(self.upper_bound)))[0]
        o_range = self.nr_normal
        o_le = o_range + self.nr_range
        o_ge = o_le + self.nr_le
        o_bound = o_ge + self.nr_ge
        o_rb = o_bound + len(bounded)
        out = []
        for c, v in bfs:
            if c < o_range:
                out.append((c, v * s[c]))
            elif c < o_le:
                out.append((c, v / r[self.nr_eq + (c - o_range)]))
            elif c < o_ge:
                out.append((c, v / r[self.nr_eq + self.nr_range + (c - o_le)]))
            elif c < o_bound:
                out.append((c, v / r[self.nr_eq + self.nr_range + self.nr_le + (c - o_ge)]))
            elif c < o_rb:
                out.append((c, v * s[bounded[c - o_bound]]))
            else:
                out.append((c, v / r[self.nr_eq + (c - o_rb)]))
        return out

    @classmethod
    def from_dense_le(cls, A: np.ndarray, b: np.ndarray, c: np.ndarray) -> "MatrixData":
        """``min c'x, A x <= b, x >= 0`` (all rows `<=`, no bounds): the synthetic dense configs."""
        m, n = A.shape
        return cls(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=np.asarray(b, dtype=np.float64),
                   cost=np.asarray(c, dtype=np.float64), upper_bound=np.full(n, np.inf),
                   dense=np.asfortranarray(A, dtype=np.float64))

    @classmethod
    def from_sparse_dict(cls, d) -> "MatrixData":
        """From ``synthetic.sparse_lp`` / ``synthetic.mixed_lp``."""
        return cls(nr_normal=d["n"], nr_eq=d["nr_eq"], nr_range=d.get("nr_range", 0), nr_le=d["nr_le"],
                   nr_ge=d["nr_ge"], b=d["b"], cost=d["c"], upper_bound=d["ub"], ranges=d.get("ranges", np.zeros(0)),
                   col_ptr=d["col_ptr"], row_idx=d["row_idx"], values=d["values"])
