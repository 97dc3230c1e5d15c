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
