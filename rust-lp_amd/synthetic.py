"""Deterministic synthetic LP generators.

Every number is an integer-defined rational, produced by a counter-based SplitMix64 so that the
exact oracle (Fractions), the f64 CPU oracle, the numpy host path and the on-device fill kernel
(``relp_synth_fill_dense`` in csrc/relp_kernels_revised.hip) all see the same data:

  x(stream, idx) = mix(seed + stream * 0xD1B54A32D192ED03 + (idx + 1) * 0x9E3779B97F4A7C15)  (mod 2^64)
  A[i, j] = (1 + x(0, j*m + i) % 999) / 1000             in [0.001, 0.999]
  b[i]    = n * (1000 + x(1, i) % 1000) / 4000           in [n/4, n/2)
  c[j]    = -(1000 + x(2, j) % 1000) / 1000              in (-2, -1]

``dense_lp(m, n, seed)``: min c'x  s.t.  A x <= b, x >= 0.  All rows are `<=`, b > 0, so the
slack basis is feasible (0 artificials); A > 0 and c < 0 make it bounded.
"""
from __future__ import annotations

from fractions import Fraction
from typing import Dict

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
STREAM = np.uint64(0xD1B54A32D192ED03)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """Counter-based SplitMix64 output for counters ``idx`` (uint64 array)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(stream) * STREAM + (idx.astype(np.uint64) + np.uint64(1)) * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        return z ^ (z >> np.uint64(31))


def dense_numerators(m: int, n: int, seed: int) -> Dict[str, np.ndarray]:
    """Integer numerators: A_num (m x n, Fortran order) / 1000, b_num / 4000, c_num / 1000."""
    idx = np.arange(m * n, dtype=np.uint64)
    a_num = (np.uint64(1) + splitmix64(seed, 0, idx) % np.uint64(999)).astype(np.int64).reshape((m, n), order="F")
    b_num = (np.int64(n) * (1000 + (splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)))
    c_num = -(1000 + (splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64))
    return {"A_num": a_num, "b_num": b_num, "c_num": c_num}


def dense_lp(m: int, n: int, seed: int) -> Dict[str, np.ndarray]:
    """f64 arrays of the dense LP: A (m x n, column-major), b, c."""
    nums = dense_numerators(m, n, seed)
    return {
        "A": np.asfortranarray(nums["A_num"].astype(np.float64) / 1000.0),
        "b": nums["b_num"].astype(np.float64) / 4000.0,
        "c": nums["c_num"].astype(np.float64) / 1000.0,
        "m": m, "n": n, "seed": seed,
    }


def dense_lp_exact(m: int, n: int, seed: int):
    """The same LP as exact ``Fraction`` data: (columns, b, c) with columns as sorted sparse vectors."""
    nums = dense_numerators(m, n, seed)
    cols = [[(i, Fraction(int(nums["A_num"][i, j]), 1000)) for i in range(m)] for j in range(n)]
    b = [Fraction(int(v), 4000) for v in nums["b_num"]]
    c = [Fraction(int(v), 1000) for v in nums["c_num"]]
    return cols, b, c


def sparse_lp(m: int, n: int, seed: int, nnz_per_col: int = 6, frac_eq: float = 0.2, frac_ge: float = 0.1,
              frac_bounded: float = 0.2) -> Dict[str, object]:
    """Netlib-shaped sparse LP with ==, <= and >= rows and some upper-bounded variables.

    All data are small integers (exact in f64 and as Fractions).  Built around a feasible point
    x0 >= 0 (so phase 1 succeeds) with strictly positive costs (so phase 2 is bounded):
      rows [0, n_eq) are ==, then n_le rows <=, then n_ge rows >= (the MatrixData order, no ranges);
      A entries in [-9, 9] without 0; == and >= rows are sign-flipped where needed so that b >= 0.
    Returns CSC arrays over the m constraint rows, b, c and upper bounds (+inf = none).
    """
    idx = np.arange(n * nnz_per_col, dtype=np.uint64)
    rows = (splitmix64(seed, 3, idx) % np.uint64(m)).astype(np.int64).reshape(n, nnz_per_col)
    vals = ((splitmix64(seed, 4, idx) % np.uint64(19)).astype(np.int64) - 9).reshape(n, nnz_per_col)
    vals[vals == 0] = 1
    n_eq = int(m * frac_eq)
    n_ge = int(m * frac_ge)
    n_le = m - n_eq - n_ge
    x0 = (splitmix64(seed, 5, np.arange(n, dtype=np.uint64)) % np.uint64(4)).astype(np.int64)
    cols = []
    for j in range(n):
        seen = {}
        for r, v in zip(rows[j], vals[j]):
            seen[int(r)] = int(v)          # duplicates: last one wins
        cols.append(seen)
    ax0 = np.zeros(m, dtype=np.int64)
    for j, col in enumerate(cols):
        for r, v in col.items():
            ax0[r] += v * x0[j]
    flip = np.ones(m, dtype=np.int64)
    flip[:n_eq][ax0[:n_eq] < 0] = -1                       # == rows: b = |a.x0|
    flip[n_eq + n_le:][ax0[n_eq + n_le:] < 0] = -1         # >= rows: make a.x0 >= 0
    ax0 = ax0 * flip
    slack = (splitmix64(seed, 6, np.arange(m, dtype=np.uint64)) % np.uint64(5)).astype(np.int64)
    b = ax0.copy()
    b[n_eq:n_eq + n_le] = np.maximum(ax0[n_eq:n_eq + n_le], 0) + slack[n_eq:n_eq + n_le]
    b[n_eq + n_le:] = np.maximum(ax0[n_eq + n_le:] - slack[n_eq + n_le:], 0)
    col_ptr = [0]
    row_idx, values = [], []
    for col in cols:
        for r in sorted(col):
            row_idx.append(r)
            values.append(col[r] * int(flip[r]))
        col_ptr.append(len(row_idx))
    c = 1 + (splitmix64(seed, 7, np.arange(n, dtype=np.uint64)) % np.uint64(20)).astype(np.int64)
    bounded = (splitmix64(seed, 8, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64) < int(1000 * frac_bounded)
    ub = np.where(bounded, (x0 + 2).astype(np.float64), np.inf)
    return {
        "m": m, "n": n, "nr_eq": n_eq, "nr_range": 0, "nr_le": n_le, "nr_ge": n_ge,
        "col_ptr": np.array(col_ptr, dtype=np.int64), "row_idx": np.array(row_idx, dtype=np.int32),
        "values": np.array(values, dtype=np.float64), "b": b.astype(np.float64),
        "c": c.astype(np.float64), "ub": ub,
    }


def mixed_lp(m: int, n: int, seed: int, nnz_per_col: int = 4, frac_eq: float = 0.2, frac_range: float = 0.15,
             frac_ge: float = 0.15, frac_bounded: float = 0.3, frac_negative_cost: float = 0.0,
             infeasible: bool = False) -> Dict[str, object]:
    """Random LP in `MatrixData` form with every row kind ([== | range | <= | >=]), upper-bounded variables,
    optionally negative costs (phase 2 may be unbounded) and optionally a contradictory pair of equality rows
    (phase 1 ends infeasible).  Small integer data; feasible around a point x0 >= 0 unless ``infeasible``.
    Test generator (numpy `Generator`, deterministic in ``seed``); same dictionary layout as `sparse_lp`
    plus ``ranges`` (width r of `b - r <= a.x <= b` per range row)."""
    rng = np.random.default_rng(seed)
    n_eq, n_rg, n_ge = int(m * frac_eq), int(m * frac_range), int(m * frac_ge)
    if infeasible:
        n_eq = max(n_eq, 2)
    n_le = m - n_eq - n_rg - n_ge
    assert n_le >= 0
    x0 = rng.integers(0, 4, size=n)
    cols = []
    for j in range(n):
        rows = rng.integers(0, m, size=nnz_per_col)
        vals = rng.integers(-9, 10, size=nnz_per_col)
        vals[vals == 0] = 1
        cols.append({int(r): int(v) for r, v in zip(rows, vals)})
    if infeasible:                                           # row 1 := row 0, right-hand sides differ below
        for col in cols:
            col.pop(1, None)
            if 0 in col:
                col[1] = col[0]
    ax0 = np.zeros(m, dtype=np.int64)
    for j, col in enumerate(cols):
        for r, v in col.items():
            ax0[r] += v * x0[j]
    lo_ge = n_eq + n_rg + n_le
    flip = np.ones(m, dtype=np.int64)
    for i in list(range(n_eq + n_rg)) + list(range(lo_ge, m)):
        if ax0[i] < 0:
            flip[i] = -1
    if infeasible:
        flip[1] = flip[0]
    ax0 = ax0 * flip
    slack = rng.integers(0, 5, size=m)
    b = ax0.copy()
    ranges = np.zeros(n_rg, dtype=np.float64)
    for k in range(n_rg):
        i = n_eq + k
        b[i] = ax0[i] + slack[i]
        ranges[k] = slack[i] + int(rng.integers(0, 4)) + 1
    b[n_eq + n_rg:lo_ge] = np.maximum(ax0[n_eq + n_rg:lo_ge], 0) + slack[n_eq + n_rg:lo_ge]
    b[lo_ge:] = np.maximum(ax0[lo_ge:] - slack[lo_ge:], 0)
    if infeasible:
        b[1] = b[0] + 1
    col_ptr, row_idx, values = [0], [], []
    for col in cols:
        for r in sorted(col):
            row_idx.append(r)
            values.append(col[r] * int(flip[r]))
        col_ptr.append(len(row_idx))
    c = rng.integers(1, 21, size=n).astype(np.float64)
    c[rng.random(n) < frac_negative_cost] *= -1.0
    ub = np.where(rng.random(n) < frac_bounded, (x0 + 2).astype(np.float64), np.inf)
    return {"m": m, "n": n, "nr_eq": n_eq, "nr_range": n_rg, "nr_le": n_le, "nr_ge": n_ge,
            "col_ptr": np.array(col_ptr, dtype=np.int64), "row_idx": np.array(row_idx, dtype=np.int32),
            "values": np.array(values, dtype=np.float64), "b": b.astype(np.float64), "c": c, "ub": ub, "ranges": ranges}


def multicommodity_lp(nodes: int, arcs: int, commodities: int, seed: int, drop_one_node: bool = True) -> Dict[str, object]:
    """Multi-commodity minimum-cost flow, the shape of the reference's KEN-* / PDS-* files (tests/netlib/problem_files):
    per commodity k a node-arc incidence block (flow conservation, == rows), and one bundle capacity row (<=) per arc
    coupling the commodities.  Column (k, a): +1 in row (k, tail_a), -1 in row (k, head_a), +1 in the capacity row of a.

    Built around a feasible flow x0 >= 0 (phase 1 ends feasible) with strictly positive costs (phase 2 is bounded); all data
    small integers.  Conservation rows are sign-flipped where needed so that b >= 0.  ``drop_one_node``: leave out the
    conservation row of node 0 of every commodity (the rows of a block sum to zero: with it the LP is rank deficient and
    the solve goes through the artificial-removal / row-removal path).
    m = commodities * (nodes - drop) + arcs rows, n = commodities * arcs columns, 3 entries per column (2 for arcs at node 0).
    Same dictionary layout as `sparse_lp`."""
    V, E, K = nodes, arcs, commodities
    e = np.arange(E, dtype=np.uint64)
    tail = (splitmix64(seed, 11, e) % np.uint64(V)).astype(np.int64)
    hop = 1 + (splitmix64(seed, 12, e) % np.uint64(V - 1)).astype(np.int64)
    head = (tail + hop) % V                                             # never a loop
    tail[:V] = np.arange(V)
    head[:V] = (np.arange(V) + 1) % V                                   # a ring first: the graph is connected
    ke = np.arange(K * E, dtype=np.uint64)
    draw = (splitmix64(seed, 13, ke) % np.uint64(16)).astype(np.int64)
    x0 = np.where(draw < 3, draw + 1, 0).reshape(K, E)                  # ~ 1 in 5 arcs carries flow
    first = 0 if not drop_one_node else 1
    nv = V - first
    n_eq = K * nv
    m = n_eq + E
    # conservation right-hand sides and the sign of every conservation row
    bal = np.zeros((K, V), dtype=np.int64)
    for k in range(K):
        np.add.at(bal[k], tail, x0[k])
        np.subtract.at(bal[k], head, x0[k])
    sign = np.where(bal < 0, -1, 1)
    cap_slack = 1 + (splitmix64(seed, 14, e) % np.uint64(4)).astype(np.int64)
    b = np.concatenate([(bal * sign)[:, first:].reshape(-1), x0.sum(axis=0) + cap_slack]).astype(np.float64)
    # columns, commodity-major: rows sorted within a column
    kk = np.repeat(np.arange(K), E)
    tt, hh, aa = np.tile(tail, K), np.tile(head, K), np.tile(np.arange(E), K)
    r_t = kk * nv + (tt - first)
    r_h = kk * nv + (hh - first)
    v_t = sign[kk, tt].astype(np.float64)
    v_h = -sign[kk, hh].astype(np.float64)
    keep_t, keep_h = tt >= first, hh >= first
    rows3 = np.stack([np.where(keep_t, r_t, -1), np.where(keep_h, r_h, -1), n_eq + aa], axis=1)
    vals3 = np.stack([v_t, v_h, np.ones(K * E)], axis=1)
    order = np.argsort(np.where(rows3 < 0, np.iinfo(np.int64).max, rows3), axis=1, kind="stable")
    rows3 = np.take_along_axis(rows3, order, axis=1)
    vals3 = np.take_along_axis(vals3, order, axis=1)
    present = rows3 >= 0
    counts = present.sum(axis=1)
    col_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    c = 1 + (splitmix64(seed, 15, ke) % np.uint64(20)).astype(np.int64)
    return {
        "m": int(m), "n": int(K * E), "nr_eq": int(n_eq), "nr_range": 0, "nr_le": int(E), "nr_ge": 0,
        "col_ptr": col_ptr, "row_idx": rows3[present].astype(np.int32), "values": vals3[present].astype(np.float64),
        "b": b, "c": c.astype(np.float64), "ub": np.full(K * E, np.inf),
    }
