"""Exact-rational CPU restatement of RELP's revised-simplex pivot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  The product path (``rust-lp_amd/``) never imports this module.

Arithmetic: ``fractions.Fraction`` (canonical reduced fractions).  The reference computes in
``RationalBig`` = ``num::BigRational`` (crate ``num 0.4``, un-vendored, no lock file:
``Cargo.toml:17``); reduced-fraction arithmetic is canonical so every correct exact
implementation yields identical values.

Every class / function cites the reference file:line (relative to /root/reference/) that it
restates.  Sparse vectors are sorted lists of ``(index, Fraction)`` with exact zeros removed,
like the reference's ``SparseVector`` / ``Vec<(usize, F)>``.

Pinned by ``tests/test_oracle_golden.py`` against every known-answer test the reference holds
for this path: problem_1 / problem_2 carries, tableau unit pins, LU /
eta / permutation known answers, the Elble-Sahinidis 5x5 update, and end-to-end optima.
"""
from __future__ import annotations

import heapq
from bisect import bisect_left
from fractions import Fraction
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

F = Fraction

def usize_sub(j: int, k: int) -> int:
    """`j - k` on `usize` as the reference's integration tests see it: they are built with `--release`
    (.github/workflows/main.yml:34-38), i.e. without overflow checks, so the subtraction wraps.  It
    matters in one situation: an artificial variable that re-entered the basis in another row than its
    own survives `remove_artificial_basis_variables` (the zero-level pivot is made in its ORIGINAL row,
    phase_one.rs:236), and `basis_column -= nr_artificial` (carry/mod.rs:524, 663) turns its index into a
    huge one.  That index classifies as the last column group (`column_type`, matrix_data.rs:198-222:
    no cost), sorts last in Bland's tie-break, and the variable stays basic at value zero until the
    ratio test removes it.  Netlib BOEING2 walks through this state on its way to the pinned optimum."""
    return (j - k) % (1 << 64)

ZERO = Fraction(0)
ONE = Fraction(1)
SparseVec = List[Tuple[int, Fraction]]


# --------------------------------------------------------------------------------------------
# Small helpers
# --------------------------------------------------------------------------------------------
def _find(items: SparseVec, index: int) -> Tuple[bool, int]:
    """Binary search on the index key; (found, position) like Rust's ``binary_search_by_key``."""
    lo, hi = 0, len(items)
    while lo < hi:
        mid = (lo + hi) // 2
        if items[mid][0] < index:
            lo = mid + 1
        else:
            hi = mid
    return (lo < len(items) and items[lo][0] == index), lo


def sparse_get(items: SparseVec, index: int) -> Optional[Fraction]:
    found, pos = _find(items, index)
    return items[pos][1] if found else None


def remove_indices(vector: list, indices: Sequence[int]) -> None:
    """src/algorithm/utilities.rs:12-36 (in place; ``indices`` sorted, unique)."""
    drop = set(indices)
    vector[:] = [v for k, v in enumerate(vector) if k not in drop]


def remove_sparse_indices(vector: SparseVec, indices: Sequence[int]) -> None:
    """src/algorithm/utilities.rs:47-69: delete entries at ``indices`` and shift the rest down."""
    if not indices or not vector:
        return
    out = []
    skipped = 0
    for (i, v) in vector:
        while skipped < len(indices) and indices[skipped] < i:
            skipped += 1
        if skipped < len(indices) and indices[skipped] == i:
            continue
        out.append((i - skipped, v))
    vector[:] = out


class _OrderedMap:
    """Stand-in for ``BTreeMap<usize, F>`` with pop_first / pop_last (lower_upper/mod.rs:236-374).

    ``direction`` = +1 pops the smallest key, -1 the largest.  Lazy-deletion heap over a dict.
    """

    def __init__(self, items: Iterable[Tuple[int, Fraction]], direction: int):
        self.d: Dict[int, Fraction] = {}
        self.sign = direction
        self.heap: List[int] = []
        for k, v in items:
            self.d[k] = v
            self.heap.append(self.sign * k)
        heapq.heapify(self.heap)

    def pop(self) -> Optional[Tuple[int, Fraction]]:
        while self.heap:
            k = self.sign * heapq.heappop(self.heap)
            if k in self.d:
                return k, self.d.pop(k)
        return None

    def insert_or_shift_maybe_remove(self, index: int, change: Fraction) -> None:
        """lower_upper/mod.rs:359-374."""
        existing = self.d.get(index)
        if existing is None:
            self.d[index] = -change
            heapq.heappush(self.heap, self.sign * index)
        else:
            new = existing - change
            if new == 0:
                del self.d[index]
            else:
                self.d[index] = new


# --------------------------------------------------------------------------------------------
# Permutations  (lower_upper/permutation/*.rs)
# --------------------------------------------------------------------------------------------
class FullPermutation:
    """permutation/full.rs:13-76."""

    def __init__(self, forward: List[int]):
        self.fwd = list(forward)
        self.bwd = [0] * len(forward)
        for i, j in enumerate(forward):
            self.bwd[j] = i

    @classmethod
    def identity(cls, n: int) -> "FullPermutation":
        return cls(list(range(n)))

    def invert(self) -> None:
        self.fwd, self.bwd = self.bwd, self.fwd

    def forward(self, i: int) -> int:
        return self.fwd[i]

    def backward(self, i: int) -> int:
        return self.bwd[i]

    def __len__(self) -> int:
        return len(self.fwd)

    def forward_sorted(self, items: SparseVec) -> None:
        items[:] = sorted(((self.fwd[i], v) for i, v in items), key=lambda t: t[0])

    def backward_sorted(self, items: SparseVec) -> None:
        items[:] = sorted(((self.bwd[i], v) for i, v in items), key=lambda t: t[0])

    def backward_unsorted(self, items: SparseVec) -> None:
        items[:] = [(self.bwd[i], v) for i, v in items]

    def __eq__(self, other) -> bool:
        return isinstance(other, FullPermutation) and self.fwd == other.fwd and self.bwd == other.bwd

    def __repr__(self) -> str:
        return f"FullPermutation({self.fwd})"


class RotateToBack:
    """permutation/rotate_to_back.rs:15-110: ``index`` -> len-1, everything above moves down one."""

    def __init__(self, index: int, length: int):
        assert index < length
        self.index = index
        self.len = length

    def forward(self, i: int) -> int:
        if i < self.index:
            return i
        if i == self.index:
            return self.len - 1
        return i - 1

    def backward(self, i: int) -> int:
        if i < self.index:
            return i
        if i < self.len - 1:
            return i + 1
        return self.index

    def forward_sorted(self, items: SparseVec) -> None:
        found, pos = _find(items, self.index)
        if found:
            moved = items[pos][1]
            rest = [(i - 1, v) for i, v in items[pos + 1:]]
            items[pos:] = rest + [(self.len - 1, moved)]
        else:
            items[pos:] = [(i - 1, v) for i, v in items[pos:]]

    def backward_sorted(self, items: SparseVec) -> None:
        if not items:
            return
        _, pos = _find(items, self.index)
        if pos == len(items):
            return
        if items[-1][0] == self.len - 1:
            moved = items[-1][1]
            rest = [(i + 1, v) for i, v in items[pos:-1]]
            items[pos:] = [(self.index, moved)] + rest
        else:
            items[pos:] = [(i + 1, v) for i, v in items[pos:]]

    def backward_unsorted(self, items: SparseVec) -> None:
        items[:] = [(self.backward(i), v) for i, v in items]

    def __eq__(self, other) -> bool:
        return isinstance(other, RotateToBack) and (self.index, self.len) == (other.index, other.len)

    def __repr__(self) -> str:
        return f"RotateToBack({self.index}, {self.len})"


class SwapPermutation:
    """permutation/swap.rs:9-82."""

    def __init__(self, a: int, b: int, length: int):
        self.a, self.b, self.len = a, b, length

    def forward(self, i: int) -> int:
        if i == self.a:
            return self.b
        if i == self.b:
            return self.a
        return i

    backward = forward

    def forward_sorted(self, items: SparseVec) -> None:
        # Equivalent to swap.rs:40-69 (rotate the moved element into place); result is the
        # index-swapped vector, sorted.
        if self.a == self.b:
            return
        fa, _ = _find(items, self.a)
        fb, _ = _find(items, self.b)
        if not fa and not fb:
            return
        items[:] = sorted(((self.forward(i), v) for i, v in items), key=lambda t: t[0])

    backward_sorted = forward_sorted


# --------------------------------------------------------------------------------------------
# Eta file  (lower_upper/eta_file.rs)
# --------------------------------------------------------------------------------------------
def _update_value(difference: Fraction, found: bool, pos: int, new_index: int, vector: SparseVec) -> None:
    """eta_file.rs:136-156."""
    if difference != 0:
        if found:
            new = vector[pos][1] - difference
            if new == 0:
                del vector[pos]
            else:
                vector[pos] = (vector[pos][0], new)
        else:
            vector.insert(pos, (new_index, -difference))


class EtaFile:
    """eta_file.rs:14-134.  R = I + e_p r' with r' non-zero only right of the pivot."""

    def __init__(self, values: SparseVec, pivot: int, length: int):
        assert all(values[k][0] < values[k + 1][0] for k in range(len(values) - 1))
        assert not values or values[0][0] > pivot
        assert not values or values[-1][0] < length
        assert pivot < length
        self.values = list(values)
        self.pivot = pivot
        self.len = length

    def apply_left(self, vector: SparseVec) -> None:
        """eta_file.rs:49-65: x := x R (row-vector times matrix)."""
        found, pivot_pos = _find(vector, self.pivot)
        if found:
            for (j, value) in self.values:
                has, pos = _find(vector, j)
                # the pivot is left of every j, so its position is stable under inserts at pos
                difference = value * vector[pivot_pos][1]
                _update_value(difference, has, pos, j, vector)

    def apply_right(self, vector: SparseVec) -> None:
        """eta_file.rs:72-104: x := R x."""
        found, pivot_pos = _find(vector, self.pivot)
        total = ZERO
        e, v = 0, pivot_pos
        while e < len(self.values) and v < len(vector):
            ei, vi = self.values[e][0], vector[v][0]
            if ei < vi:
                e += 1
            elif ei == vi:
                total += self.values[e][1] * vector[v][1]
                e += 1
                v += 1
            else:
                v += 1
        _update_value(total, found, pivot_pos, self.pivot, vector)

    def update_spike_pivot_value(self, spike: SparseVec) -> None:
        """eta_file.rs:111-133."""
        found, pos = _find(spike, self.pivot)
        search = pos + 1 if found else pos
        tail = spike[search:]
        difference = ZERO
        for (j, value) in self.values:
            has, p = _find(tail, j)
            if has:
                difference += value * tail[p][1]
        _update_value(difference, found, pos, self.pivot, spike)

    def __eq__(self, other) -> bool:
        return (isinstance(other, EtaFile) and self.values == other.values
                and self.pivot == other.pivot and self.len == other.len)

    def __repr__(self) -> str:
        return f"EtaFile({self.values}, pivot={self.pivot}, len={self.len})"


# --------------------------------------------------------------------------------------------
# LU decomposition with Forrest-Tomlin-style updates  (lower_upper/**)
# --------------------------------------------------------------------------------------------
def subtract_multiple_of_row_from_other_row(to_edit: SparseVec, ratio: Fraction,
                                            being_removed: SparseVec):
    """decomposition/mod.rs:141-205.  Returns ((net_removed, net_added), cols_removed, cols_added)."""
    added: List[int] = []
    if not to_edit:
        for (j, v) in being_removed:
            to_edit.append((j, -ratio * v))
            added.append(j)
        return (0, len(being_removed)), [], added
    old = list(to_edit)
    removed: List[int] = []
    new: SparseVec = []
    index = 0
    for (j, old_value) in old:
        while index < len(being_removed) and being_removed[index][0] < j:
            new.append((being_removed[index][0], -ratio * being_removed[index][1]))
            added.append(being_removed[index][0])
            index += 1
        if index < len(being_removed) and being_removed[index][0] == j:
            product = ratio * being_removed[index][1]
            if product != old_value:
                new.append((j, old_value - product))
            else:
                removed.append(j)
            index += 1
        else:
            new.append((j, old_value))
    while index < len(being_removed):
        new.append((being_removed[index][0], -ratio * being_removed[index][1]))
        added.append(being_removed[index][0])
        index += 1
    to_edit[:] = new
    if len(new) < len(old):
        net = (len(old) - len(new), 0)
    elif len(new) == len(old):
        net = (0, 0)
    else:
        net = (0, len(new) - len(old))
    return net, removed, added


def markowitz_choose_pivot(row_counts, column_counts, rows: List[SparseVec], k: int) -> Tuple[int, int]:
    """decomposition/pivoting.rs:45-81: candidates in row order, stably sorted by column, first
    minimum of (r-1)(c-1) taken (Rust ``min_by_key`` keeps the first of equal minima)."""
    pairs = []
    for i in range(k, len(rows)):
        row = rows[i]
        _, first = _find(row, k)
        for (j, _v) in row[first:]:
            pairs.append((i, j))
    pairs.sort(key=lambda t: t[1])  # stable
    best = None
    best_key = None
    for (i, j) in pairs:
        key = (row_counts[i] - 1) * (column_counts[j] - 1)
        if best_key is None or key < best_key:
            best, best_key = (i, j), key
    return best


class LUDecomposition:
    """lower_upper/mod.rs:35-57: PBQ = LU plus a list of (EtaFile, RotateToBack) updates."""

    REFACTOR_AFTER = 10  # lower_upper/mod.rs:199-202 (``updates.len() > 10``)

    def __init__(self, row_permutation, column_permutation, lower, upper, updates=None):
        self.row_permutation: FullPermutation = row_permutation
        self.column_permutation: FullPermutation = column_permutation
        self.lower: List[SparseVec] = lower      # column major, unit diagonal implied, m-1 columns
        self.upper: List[SparseVec] = upper      # column major, diagonal stored last per column
        self.updates: List[Tuple[EtaFile, RotateToBack]] = updates or []

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def identity(cls, m: int) -> "LUDecomposition":
        """lower_upper/mod.rs:66-74."""
        return cls(FullPermutation.identity(m), FullPermutation.identity(m),
                   [[] for _ in range(m - 1)], [[(i, ONE)] for i in range(m)], [])

    @classmethod
    def invert(cls, columns: List[SparseVec]) -> "LUDecomposition":
        """lower_upper/mod.rs:76-90: gather the basis columns row-major, then factor."""
        m = len(columns)
        rows: List[SparseVec] = [[] for _ in range(m)]
        for j, column in enumerate(columns):
            for (i, value) in column:
                rows[i].append((j, Fraction(value)))
        return cls.rows(rows)

    @classmethod
    def rows(cls, rows: List[SparseVec]) -> "LUDecomposition":
        """decomposition/mod.rs:27-138: right-looking LU with Markowitz pivoting."""
        rows = [list(r) for r in rows]
        m = len(rows)
        assert m > 1  # decomposition/mod.rs:32
        row_perm = list(range(m))
        col_perm = list(range(m))
        lower_rm: List[SparseVec] = [[] for _ in range(m - 1)]
        nnz_row = [len(r) for r in rows]
        nnz_col = [0] * m
        for r in rows:
            for (j, _v) in r:
                nnz_col[j] += 1
        for k in range(m):
            pr, pc = markowitz_choose_pivot(nnz_row, nnz_col, rows, k)
            # swap (pr, pc) to (k, k): decomposition/mod.rs:219-268
            if pr != k:
                row_perm[pr], row_perm[k] = row_perm[k], row_perm[pr]
                nnz_row[pr], nnz_row[k] = nnz_row[k], nnz_row[pr]
                rows[pr], rows[k] = rows[k], rows[pr]
                if pr > 0 and k > 0:
                    lower_rm[pr - 1], lower_rm[k - 1] = lower_rm[k - 1], lower_rm[pr - 1]
            if pc != k:
                col_perm[pc], col_perm[k] = col_perm[k], col_perm[pc]
                nnz_col[pc], nnz_col[k] = nnz_col[k], nnz_col[pc]
                swap = SwapPermutation(pc, k, m)
                for r in rows:
                    swap.forward_sorted(r)
            for (j, _v) in rows[k]:
                nnz_row[k] -= 1
                nnz_col[j] -= 1
            current = rows[k]
            pivot_value = current[0][1]
            ratios = []
            for i in range(k + 1, m):
                row = rows[i]
                if row and row[0][0] == k:
                    ratios.append((i, row.pop(0)[1] / pivot_value))
                    nnz_row[i] -= 1
                    nnz_col[k] -= 1
            for (i, ratio) in ratios:
                net, removed, added = subtract_multiple_of_row_from_other_row(rows[i], ratio, current[1:])
                nnz_row[i] -= net[0]
                nnz_row[i] += net[1]
                for c in removed:
                    nnz_col[c] -= 1
                for c in added:
                    nnz_col[c] += 1
                lower_rm[i - 1].append((k, ratio))
        upper: List[SparseVec] = [[] for _ in range(m)]
        for i, row in enumerate(rows):
            for (j, value) in row:
                upper[j].append((i, value))
        lower: List[SparseVec] = [[] for _ in range(m - 1)]
        for idx, row in enumerate(lower_rm):
            i = idx + 1
            for (j, v) in row:
                lower[j].append((i, v))
        rp = FullPermutation(row_perm)
        rp.invert()
        cp = FullPermutation(col_perm)
        cp.invert()
        return cls(rp, cp, lower, upper, [])

    # -- BasisInverse interface -----------------------------------------------------------
    def m(self) -> int:
        return len(self.row_permutation)

    def should_refactor(self) -> bool:
        return len(self.updates) > self.REFACTOR_AFTER

    def generate_column(self, original_column: SparseVec) -> "ColumnAndSpike":
        """FTRAN, lower_upper/mod.rs:157-190."""
        rhs = [(self.row_permutation.forward(i), Fraction(v)) for (i, v) in original_column]
        w = self.invert_lower_right(rhs)
        for (eta, q) in self.updates:
            eta.apply_right(w)
            q.forward_sorted(w)
        spike = list(w)
        column = self.invert_upper_right(w)
        for (_eta, q) in reversed(self.updates):
            q.backward_unsorted(column)
        self.column_permutation.backward_unsorted(column)
        column.sort(key=lambda t: t[0])
        return ColumnAndSpike(column, spike)

    def generate_element(self, i: int, original_column: SparseVec) -> Optional[Fraction]:
        """lower_upper/mod.rs:192-197."""
        return sparse_get(self.generate_column(original_column).column, i)

    def basis_inverse_row(self, row: int) -> SparseVec:
        """BTRAN of a unit vector, lower_upper/mod.rs:204-222."""
        row = self.column_permutation.forward(row)
        for (_eta, q) in self.updates:
            row = q.forward(row)
        w = self.invert_upper_left([(row, ONE)])
        for (eta, q) in reversed(self.updates):
            q.backward_sorted(w)
            eta.apply_left(w)
        tuples = self.invert_lower_left(w)
        self.row_permutation.backward_sorted(tuples)
        return tuples

    def change_basis(self, pivot_row_index: int, column: "ColumnAndSpike") -> None:
        """Forrest-Tomlin-style update, lower_upper/mod.rs:92-155."""
        m = self.m()
        p = self.column_permutation.forward(pivot_row_index)
        for (_eta, q) in self.updates:
            p = q.forward(p)
        u_bar: SparseVec = []
        to_zero: List[Tuple[int, int]] = []
        for j in range(p + 1, m):
            found, pos = _find(self.upper[j], p)
            if found:
                u_bar.append((j, self.upper[j][pos][1]))
                to_zero.append((j, pos))
        r = self.invert_upper_left(u_bar)
        eta = EtaFile(r, p, m)
        for (j, pos) in to_zero:
            del self.upper[j][pos]
        spike = list(column.spike)
        eta.update_spike_pivot_value(spike)
        assert _find(spike, p)[0], "spike pivot value present (non-singular)"
        self.upper[p] = spike
        self.upper[p:] = self.upper[p + 1:] + [self.upper[p]]
        q = RotateToBack(p, m)
        for j in range(p, m):
            q.forward_sorted(self.upper[j])
        self.updates.append((eta, q))

    # -- triangular solves ----------------------------------------------------------------
    def invert_lower_right(self, rhs: SparseVec) -> SparseVec:
        """L y = rhs, lower_upper/mod.rs:236-255."""
        m = self.m()
        work = _OrderedMap(rhs, +1)
        result: SparseVec = []
        while True:
            item = work.pop()
            if item is None:
                break
            row, value = item
            if row != m - 1:
                for (i, l) in self.lower[row]:
                    work.insert_or_shift_maybe_remove(i, value * l)
            result.append((row, value))
        return result

    def invert_upper_right(self, rhs: SparseVec) -> SparseVec:
        """U x = rhs, lower_upper/mod.rs:257-271, 292-304."""
        work = _OrderedMap(rhs, -1)
        result: SparseVec = []
        while True:
            item = work.pop()
            if item is None:
                break
            row, value = item
            column = self.upper[row]
            assert column[-1][0] == row, "diagonal element stored last"
            x = value / column[-1][1]
            for (i, u) in column[:-1]:
                work.insert_or_shift_maybe_remove(i, x * u)
            result.append((row, x))
        result.reverse()
        return result

    def invert_lower_left(self, rhs: SparseVec) -> SparseVec:
        """y L = rhs (row vector), lower_upper/mod.rs:306-330."""
        work = _OrderedMap(rhs, -1)
        result: SparseVec = []
        while True:
            item = work.pop()
            if item is None:
                break
            column, value = item
            row = column
            for j in range(column):
                found, pos = _find(self.lower[j], row)
                if found:
                    work.insert_or_shift_maybe_remove(j, value * self.lower[j][pos][1])
            result.append((row, value))
        result.reverse()
        return result

    def invert_upper_left(self, rhs: SparseVec) -> SparseVec:
        """x U = rhs (row vector), lower_upper/mod.rs:332-356."""
        m = self.m()
        work = _OrderedMap(rhs, +1)
        result: SparseVec = []
        while True:
            item = work.pop()
            if item is None:
                break
            column, value = item
            row = column
            x = value / self.upper[column][-1][1]
            for j in range(column + 1, m):
                found, pos = _find(self.upper[j], row)
                if found:
                    work.insert_or_shift_maybe_remove(j, x * self.upper[j][pos][1])
            result.append((column, x))
        return result

    def __eq__(self, other) -> bool:
        return (isinstance(other, LUDecomposition)
                and self.row_permutation == other.row_permutation
                and self.column_permutation == other.column_permutation
                and self.lower == other.lower and self.upper == other.upper
                and self.updates == other.updates)

    def __repr__(self) -> str:
        return (f"LU(P={self.row_permutation}, Q={self.column_permutation}, L={self.lower}, "
                f"U={self.upper}, updates={self.updates})")


class ColumnAndSpike:
    """lower_upper/mod.rs:376-391: FTRAN result + the spike saved for ``change_basis``."""

    def __init__(self, column: SparseVec, spike: SparseVec):
        self.column = column
        self.spike = spike


class _PlainColumn:
    """``ColumnComputationInfo`` for ``BasisInverseRows`` (basis_inverse_rows.rs:207-215)."""

    def __init__(self, column: SparseVec):
        self.column = column


# --------------------------------------------------------------------------------------------
# Explicit row-major basis inverse  (carry/basis_inverse_rows.rs)
# --------------------------------------------------------------------------------------------
def sparse_sparse_inner_product(row: SparseVec, column: SparseVec) -> Fraction:
    """data/linear_algebra/vector/sparse.rs:82-106."""
    total = ZERO
    i = 0
    n = len(row)
    for (index, value) in column:
        while i < n and row[i][0] < index:
            i += 1
        if i < n and row[i][0] == index:
            total += row[i][1] * value
            i += 1
    return total


def add_multiple_of_row(target: SparseVec, multiple: Fraction, other: SparseVec) -> SparseVec:
    """data/linear_algebra/vector/sparse.rs:213-248 (exact zeros dropped, :235)."""
    new: SparseVec = []
    j = 0
    no = len(other)
    for (i, value) in target:
        while j < no and other[j][0] < i:
            new.append((other[j][0], multiple * other[j][1]))
            j += 1
        if j < no and other[j][0] == i:
            nv = value + multiple * other[j][1]
            if nv != 0:
                new.append((i, nv))
            j += 1
        else:
            new.append((i, value))
    for (jj, value) in other[j:]:
        new.append((jj, multiple * value))
    return new


class BasisInverseRows:
    """carry/basis_inverse_rows.rs:20-204."""

    def __init__(self, rows: List[SparseVec]):
        self.rows_ = rows

    @classmethod
    def identity(cls, m: int) -> "BasisInverseRows":
        return cls([[(i, ONE)] for i in range(m)])

    @classmethod
    def invert(cls, columns: List[SparseVec]) -> "BasisInverseRows":
        """basis_inverse_rows.rs:103-129: LU-invert, m unit FTRANs, transpose to rows."""
        m = len(columns)
        lu = LUDecomposition.invert(columns)
        row_major: List[SparseVec] = [[] for _ in range(m)]
        for j in range(m):
            for (i, value) in lu.generate_column([(j, ONE)]).column:
                row_major[i].append((j, value))
        return cls(row_major)

    def m(self) -> int:
        return len(self.rows_)

    def should_refactor(self) -> bool:
        return False  # basis_inverse_rows.rs:175-179

    def change_basis(self, pivot_row_index: int, column: _PlainColumn) -> None:
        """basis_inverse_rows.rs:131-142, 42-83."""
        col = column.column
        pivot_value = sparse_get(col, pivot_row_index)
        assert pivot_value is not None, "Pivot value can't be zero."
        self.rows_[pivot_row_index] = [(i, v / pivot_value) for (i, v) in self.rows_[pivot_row_index]]
        pivot_row = self.rows_[pivot_row_index]
        for (edit_row, value) in col:
            if edit_row != pivot_row_index:
                self.rows_[edit_row] = add_multiple_of_row(self.rows_[edit_row], -value, pivot_row)

    def generate_column(self, original_column: SparseVec) -> _PlainColumn:
        """basis_inverse_rows.rs:144-155."""
        out: SparseVec = []
        for i in range(self.m()):
            v = self.generate_element(i, original_column)
            if v is not None:
                out.append((i, v))
        return _PlainColumn(out)

    def generate_element(self, i: int, original_column: SparseVec) -> Optional[Fraction]:
        """basis_inverse_rows.rs:157-173."""
        e = sparse_sparse_inner_product(self.rows_[i], original_column)
        return e if e != 0 else None

    def basis_inverse_row(self, row: int) -> SparseVec:
        return list(self.rows_[row])

    def remove_basis_part(self, indices: Sequence[int]) -> None:
        """basis_inverse_rows.rs:190-204."""
        remove_indices(self.rows_, indices)
        for r in self.rows_:
            remove_sparse_indices(r, indices)

    def __eq__(self, other) -> bool:
        return isinstance(other, BasisInverseRows) and self.rows_ == other.rows_


# --------------------------------------------------------------------------------------------
# MatrixData provider  (matrix_provider/matrix_data.rs)
# --------------------------------------------------------------------------------------------
class MatrixData:
    """matrix_provider/matrix_data.rs:54-457: virtual matrix [A | slacks | bound slacks].

    ``constraints``: list of structural columns, each a sorted sparse vector over the
    ``nr_constraints`` rows ordered [== | range | <= | >=].  ``upper_bounds[j]`` is ``None`` or the
    variable's upper bound (lower bounds are 0 after standardisation).
    """

    def __init__(self, constraints: List[SparseVec], b: List[Fraction], ranges: List[Fraction],
                 nr_eq: int, nr_range: int, nr_le: int, nr_ge: int,
                 costs: List[Fraction], upper_bounds: List[Optional[Fraction]]):
        assert nr_eq + nr_range + nr_le + nr_ge == len(b)
        assert len(ranges) == nr_range
        assert len(costs) == len(constraints) == len(upper_bounds)
        self.constraints = [[(i, Fraction(v)) for (i, v) in c] for c in constraints]
        self.b = [Fraction(v) for v in b]
        self.ranges = [Fraction(v) for v in ranges]
        self.nr_eq, self.nr_range, self.nr_le, self.nr_ge = nr_eq, nr_range, nr_le, nr_ge
        self.costs = [Fraction(c) for c in costs]
        self.upper_bounds = [None if u is None else Fraction(u) for u in upper_bounds]
        # matrix_data.rs:168-177
        self.bound_to_var: List[int] = []
        self.var_to_bound: List[Optional[int]] = []
        for j, u in enumerate(self.upper_bounds):
            if u is not None:
                self.var_to_bound.append(len(self.bound_to_var))
                self.bound_to_var.append(j)
            else:
                self.var_to_bound.append(None)

    # -- sizes ----------------------------------------------------------------------------
    def nr_normal(self) -> int:
        return len(self.constraints)

    def nr_constraints(self) -> int:
        return self.nr_eq + self.nr_range + self.nr_le + self.nr_ge

    def nr_variable_bounds(self) -> int:
        return len(self.bound_to_var) + self.nr_range

    def nr_rows(self) -> int:
        return self.nr_constraints() + self.nr_variable_bounds()

    def nr_columns(self) -> int:
        return self.nr_normal() + self.nr_range + self.nr_le + self.nr_ge + self.nr_variable_bounds()

    def _col_starts(self) -> List[int]:
        """matrix_data.rs:227-239."""
        amounts = [self.nr_normal(), self.nr_range, self.nr_le, self.nr_ge, len(self.bound_to_var), self.nr_range]
        out, acc = [], 0
        for a in amounts:
            out.append(acc)
            acc += a
        return out

    def _row_starts(self) -> List[int]:
        """matrix_data.rs:257-268."""
        amounts = [self.nr_eq, self.nr_range, self.nr_le, self.nr_ge, len(self.bound_to_var), self.nr_range]
        out, acc = [], 0
        for a in amounts:
            out.append(acc)
            acc += a
        return out

    def column_type(self, j: int) -> Tuple[int, int]:
        """matrix_data.rs:198-222: (group, index in group)."""
        sep = self._col_starts()
        g = 0
        while g + 1 < 6 and j >= sep[g + 1]:
            g += 1
        return g, j - sep[g]

    # -- MatrixProvider -------------------------------------------------------------------
    def column(self, j: int) -> SparseVec:
        """matrix_data.rs:308-348."""
        g, k = self.column_type(j)
        rs = self._row_starts()
        if g == 0:
            col = list(self.constraints[k])
            bi = self.var_to_bound[k]
            if bi is not None:
                col.append((self.nr_constraints() + bi, ONE))
            return col
        if g == 1:
            return [(rs[1] + k, ONE), (rs[5] + k, ONE)]
        if g == 2:
            return [(rs[2] + k, ONE)]
        if g == 3:
            return [(rs[3] + k, -ONE)]
        if g == 4:
            return [(rs[4] + k, ONE)]
        return [(rs[5] + k, ONE)]

    def cost_value(self, j: int) -> Optional[Fraction]:
        """matrix_data.rs:350-357: ``None`` (zero) for every slack."""
        g, k = self.column_type(j)
        return self.costs[k] if g == 0 else None

    def right_hand_side(self) -> List[Fraction]:
        """matrix_data.rs:359-371: (b, upper bounds, ranges)."""
        return list(self.b) + [self.upper_bounds[j] for j in self.bound_to_var] + list(self.ranges)

    def pivot_element_indices(self) -> List[Tuple[int, int]]:
        """matrix_data.rs:432-452: (row, column) of <=-slacks, var-bound slacks, range-bound slacks."""
        cs, rs = self._col_starts(), self._row_starts()
        out = [(rs[2] + j, cs[2] + j) for j in range(self.nr_le)]
        out += [(rs[4] + j, cs[4] + j) for j in range(len(self.bound_to_var))]
        out += [(rs[5] + j, cs[5] + j) for j in range(self.nr_range)]
        return out

    def reconstruct_solution(self, column_values: SparseVec) -> SparseVec:
        """matrix_data.rs:415-424: drop every slack entry."""
        return [(j, v) for (j, v) in column_values if j < self.nr_normal()]


class RemoveRows:
    """matrix_provider/filter/generic_wrapper.rs:51,224-284: provider view with rows deleted."""

    def __init__(self, provider: MatrixData, rows_to_skip: List[int]):
        self.provider = provider
        self.rows_to_skip = list(rows_to_skip)

    def filtered_rows(self) -> List[int]:
        return self.rows_to_skip

    def column(self, j: int) -> SparseVec:
        col = self.provider.column(j)
        remove_sparse_indices(col, self.rows_to_skip)
        return col

    def cost_value(self, j: int):
        return self.provider.cost_value(j)

    def right_hand_side(self) -> List[Fraction]:
        rhs = self.provider.right_hand_side()
        remove_indices(rhs, self.rows_to_skip)
        return rhs

    def nr_rows(self) -> int:
        return self.provider.nr_rows() - len(self.rows_to_skip)

    def nr_columns(self) -> int:
        return self.provider.nr_columns()

    def reconstruct_solution(self, column_values: SparseVec) -> SparseVec:
        return self.provider.reconstruct_solution(column_values)


# --------------------------------------------------------------------------------------------
# Kinds  (tableau/kind/**)
# --------------------------------------------------------------------------------------------
class NonArtificial:
    """kind/non_artificial.rs:18-70."""

    def __init__(self, provider):
        self.provider = provider

    def initial_cost_value(self, j: int) -> Fraction:
        c = self.provider.cost_value(j)
        return ZERO if c is None else c

    def original_column(self, j: int) -> SparseVec:
        return self.provider.column(j)

    def nr_rows(self) -> int:
        return self.provider.nr_rows()

    def nr_columns(self) -> int:
        return self.provider.nr_columns()

    def nr_artificial_variables(self) -> int:
        return 0


class Fully:
    """kind/artificial/fully.rs:14-69: one artificial per row, numbered before all columns."""

    def __init__(self, provider):
        self.provider = provider

    def initial_cost_value(self, j: int) -> Fraction:
        return ONE if j < self.nr_artificial_variables() else ZERO

    def original_column(self, j: int) -> SparseVec:
        if j < self.nr_rows():
            return [(j, ONE)]
        return self.provider.column(j - self.nr_rows())

    def nr_rows(self) -> int:
        return self.provider.nr_rows()

    def nr_columns(self) -> int:
        return self.nr_rows() + self.provider.nr_columns()

    def nr_artificial_variables(self) -> int:
        return self.nr_rows()

    def pivot_row_from_artificial(self, a: int) -> int:
        return a


class Partially:
    """kind/artificial/partially.rs:18-110."""

    def __init__(self, provider, column_to_row: List[int]):
        self.provider = provider
        self.column_to_row = column_to_row

    def initial_cost_value(self, j: int) -> Fraction:
        return ONE if j < self.nr_artificial_variables() else ZERO

    def original_column(self, j: int) -> SparseVec:
        na = self.nr_artificial_variables()
        if j < na:
            return [(self.column_to_row[j], ONE)]
        return self.provider.column(j - na)

    def nr_rows(self) -> int:
        return self.provider.nr_rows()

    def nr_columns(self) -> int:
        return self.nr_artificial_variables() + self.provider.nr_columns()

    def nr_artificial_variables(self) -> int:
        return len(self.column_to_row)

    def pivot_row_from_artificial(self, a: int) -> int:
        return self.column_to_row[a]


# --------------------------------------------------------------------------------------------
# Carry  (inverse_maintenance/carry/mod.rs)
# --------------------------------------------------------------------------------------------
class Carry:
    """carry/mod.rs:45-65: (-obj, -pi dense, b dense, basis_indices, basis inverse)."""

    def __init__(self, minus_objective, minus_pi, b, basis_indices, basis_inverse):
        self.minus_objective: Fraction = Fraction(minus_objective)
        self.minus_pi: List[Fraction] = [Fraction(v) for v in minus_pi]
        self.b: List[Fraction] = [Fraction(v) for v in b]
        self.basis_indices: List[int] = list(basis_indices)
        self.basis_inverse = basis_inverse

    def m(self) -> int:
        return len(self.b)

    # -- constructors ---------------------------------------------------------------------
    @classmethod
    def create_for_fully_artificial(cls, BI, rhs: List[Fraction]) -> "Carry":
        """carry/mod.rs:358-379."""
        m = len(rhs)
        return cls(-sum(rhs, ZERO), [-ONE] * m, rhs, list(range(m)), BI.identity(m))

    @classmethod
    def create_for_partially_artificial(cls, BI, artificial_rows, free_basis_values, rhs, basis_indices):
        """carry/mod.rs:381-426."""
        m = len(rhs)
        objective = ZERO
        for index in artificial_rows:
            objective += rhs[index]
        arts = set(artificial_rows)
        minus_pi = [-ONE if row in arts else ZERO for row in range(m)]
        return cls(-objective, minus_pi, rhs, basis_indices, BI.identity(m))

    @staticmethod
    def create_minus_pi_from_artificial(basis_inverse, provider, basis: List[int]) -> List[Fraction]:
        """carry/mod.rs:214-248: -pi = -(c_B' B^-1) via m unit FTRANs."""
        m = basis_inverse.m()
        b_inverse_rows: List[SparseVec] = [[] for _ in range(m)]
        for j in range(m):
            for (i, v) in basis_inverse.generate_column([(j, ONE)]).column:
                b_inverse_rows[i].append((j, v))
        pi = [ZERO] * m
        for i, row in enumerate(b_inverse_rows):
            c = provider.cost_value(basis[i])
            if c is None:
                continue
            for (j, value) in row:
                pi[j] += value * c
        return [-v for v in pi]

    @staticmethod
    def create_minus_obj_from_artificial(provider, basis: List[int], b: List[Fraction]) -> Fraction:
        """carry/mod.rs:258-271."""
        objective = ZERO
        for row in range(provider.nr_rows()):
            c = provider.cost_value(basis[row])
            if c is not None:
                objective += b[row] * c
        return -objective

    @classmethod
    def from_basis(cls, BI, basis: List[int], provider) -> "Carry":
        """carry/mod.rs:428-463 (warm start)."""
        columns = [provider.column(j) for j in basis]
        bi = BI.invert(columns)
        rhs = [(i, v) for i, v in enumerate(provider.right_hand_side()) if v != 0]
        b = [ZERO] * provider.nr_rows()
        for (i, v) in bi.generate_column(rhs).column:
            b[i] = v
        mo = cls.create_minus_obj_from_artificial(provider, basis, b)
        mp = cls.create_minus_pi_from_artificial(bi, provider, basis)
        return cls(mo, mp, b, list(basis), bi)

    @classmethod
    def from_basis_pivots(cls, BI, basis_columns: List[Tuple[int, int]], provider) -> "Carry":
        """carry/mod.rs:465-482."""
        elements = sorted(basis_columns, key=lambda t: t[0])
        return cls.from_basis(BI, [c for (_r, c) in elements], provider)

    @classmethod
    def from_artificial(cls, artificial: "Carry", provider, nr_artificial: int) -> "Carry":
        """carry/mod.rs:484-510."""
        basis = [usize_sub(j, nr_artificial) for j in artificial.basis_indices]
        mp = cls.create_minus_pi_from_artificial(artificial.basis_inverse, provider, basis)
        mo = cls.create_minus_obj_from_artificial(provider, basis, artificial.b)
        return cls(mo, mp, artificial.b, basis, artificial.basis_inverse)

    @classmethod
    def from_artificial_remove_rows(cls, BI, artificial: "Carry", rows_removed: RemoveRows,
                                    nr_artificial: int) -> "Carry":
        """carry/mod.rs:512-547 (generic: re-invert) and :650-689 (``RemoveBasisPart``)."""
        basis = list(artificial.basis_indices)
        remove_indices(basis, rows_removed.filtered_rows())
        basis = [usize_sub(j, nr_artificial) for j in basis]
        if hasattr(artificial.basis_inverse, "remove_basis_part"):
            bi = artificial.basis_inverse
            bi.remove_basis_part(rows_removed.filtered_rows())
        else:
            bi = BI.invert([rows_removed.column(j) for j in basis])
        mp = cls.create_minus_pi_from_artificial(bi, rows_removed, basis)
        b = list(artificial.b)
        remove_indices(b, rows_removed.filtered_rows())
        mo = cls.create_minus_obj_from_artificial(rows_removed, basis, b)
        return cls(mo, mp, b, basis, bi)

    # -- per-pivot operations -------------------------------------------------------------
    def update_b(self, r: int, column: SparseVec) -> None:
        """carry/mod.rs:283-313."""
        pivot_value = sparse_get(column, r)
        assert pivot_value is not None, "Pivot value can't be zero."
        self.b[r] /= pivot_value
        br = self.b[r]
        for (i, v) in column:
            if i != r:
                self.b[i] -= v * br

    def update_minus_pi_and_obj(self, r: int, relative_cost: Fraction) -> None:
        """carry/mod.rs:326-333: uses the POST-update row r of B^-1."""
        for (j, value) in self.basis_inverse.basis_inverse_row(r):
            self.minus_pi[j] -= relative_cost * value
        self.minus_objective -= relative_cost * self.b[r]

    def change_basis(self, r: int, q: int, column, relative_cost: Fraction) -> int:
        """carry/mod.rs:549-570: b first, basis inverse second, -pi third."""
        self.update_b(r, column.column)
        self.basis_inverse.change_basis(r, column)
        self.update_minus_pi_and_obj(r, relative_cost)
        leaving = self.basis_indices[r]
        self.basis_indices[r] = q
        return leaving

    def cost_difference(self, original_column: SparseVec) -> Fraction:
        """carry/mod.rs:572-577 + vector/dense.rs:81-92."""
        total = ZERO
        for (i, v) in original_column:
            total += self.minus_pi[i] * v
        return total

    def generate_column(self, original_column: SparseVec):
        return self.basis_inverse.generate_column(original_column)

    def generate_element(self, i: int, original_column: SparseVec):
        return self.basis_inverse.generate_element(i, original_column)

    def after_basis_change(self, kind) -> None:
        """carry/mod.rs:602-614: full re-inversion from original columns when asked."""
        if self.basis_inverse.should_refactor():
            columns = [kind.original_column(j) for j in self.basis_indices]
            self.basis_inverse = type(self.basis_inverse).invert(columns)

    def current_bfs(self) -> SparseVec:
        """carry/mod.rs:616-625."""
        t = [(self.basis_indices[i], v) for i, v in enumerate(self.b) if v != 0]
        t.sort(key=lambda x: x[0])
        return t

    def get_objective_function_value(self) -> Fraction:
        return -self.minus_objective


# --------------------------------------------------------------------------------------------
# Tableau  (tableau/mod.rs)
# --------------------------------------------------------------------------------------------
class Tableau:
    """tableau/mod.rs:24-38."""

    def __init__(self, inverse_maintainer: Carry, basis_columns, kind):
        self.im = inverse_maintainer
        self.basis_columns = set(basis_columns)
        self.kind = kind

    # constructors --------------------------------------------------------------------------
    @classmethod
    def new_fully_artificial(cls, BI, provider) -> "Tableau":
        """kind/artificial/fully.rs:88-98."""
        m = provider.nr_rows()
        return cls(Carry.create_for_fully_artificial(BI, provider.right_hand_side()), range(m), Fully(provider))

    @classmethod
    def new_partially_artificial(cls, BI, provider) -> "Tableau":
        """kind/artificial/partially.rs:125-206."""
        m = provider.nr_rows()
        real = provider.pivot_element_indices()
        nr_real = len(real)
        nr_artificial = m - nr_real
        artificial: List[int] = []
        i = 0
        for ith in range(nr_artificial):
            while i < nr_real and ith + i == real[i][0]:
                i += 1
            artificial.append(ith + i)
        basis_indices: List[int] = []
        ac = 0
        for row in range(m):
            can_a = ac < nr_artificial
            can_r = row - ac < nr_real
            if can_a and can_r:
                if artificial[ac] < real[row - ac][0]:
                    basis_indices.append(ac)
                    ac += 1
                else:
                    basis_indices.append(nr_artificial + real[row - ac][1])
            elif can_a:
                basis_indices.append(ac)
                ac += 1
            elif can_r:
                basis_indices.append(nr_artificial + real[row - ac][1])
            else:
                raise AssertionError("unreachable (partially.rs:183)")
        im = Carry.create_for_partially_artificial(BI, artificial, real, provider.right_hand_side(), basis_indices)
        return cls(im, basis_indices, Partially(provider, artificial))

    @classmethod
    def from_artificial(cls, im: Carry, nr_artificial: int, basis, provider) -> "Tableau":
        """kind/non_artificial.rs:151-172."""
        return cls(Carry.from_artificial(im, provider, nr_artificial),
                   [usize_sub(c, nr_artificial) for c in basis], NonArtificial(provider))

    @classmethod
    def from_artificial_removing_rows(cls, BI, im: Carry, nr_artificial: int, basis, provider: RemoveRows):
        """kind/non_artificial.rs:191-220."""
        basis = set(basis)
        for row in provider.filtered_rows():
            basis.remove(im.basis_indices[row])
        cols = [usize_sub(j, nr_artificial) for j in basis]
        return cls(Carry.from_artificial_remove_rows(BI, im, provider, nr_artificial), cols, NonArtificial(provider))

    # queries -------------------------------------------------------------------------------
    def nr_rows(self) -> int:
        return self.kind.nr_rows()

    def nr_columns(self) -> int:
        return self.kind.nr_columns()

    def is_in_basis(self, j: int) -> bool:
        return j in self.basis_columns

    def relative_cost(self, j: int) -> Fraction:
        """tableau/mod.rs:102-108."""
        return self.im.cost_difference(self.kind.original_column(j)) + self.kind.initial_cost_value(j)

    def generate_column(self, j: int):
        """tableau/mod.rs:122-126."""
        return self.im.generate_column(self.kind.original_column(j))

    def generate_element(self, i: int, j: int):
        return self.im.generate_element(i, self.kind.original_column(j))

    def objective_function_value(self) -> Fraction:
        return self.im.get_objective_function_value()

    def current_bfs(self) -> SparseVec:
        return self.im.current_bfs()

    def select_primal_pivot_row(self, column: SparseVec) -> Optional[int]:
        """RATIO TEST, tableau/mod.rs:221-247: strict minimum; exact ties -> smaller leaving column."""
        best = None
        for (row, xij) in column:
            if xij > 0:
                ratio = self.im.b[row] / xij
                leaving = self.im.basis_indices[row]
                if best is not None:
                    if ratio == best[1] and leaving < best[2]:
                        best = (row, best[1], leaving)
                    elif ratio < best[1]:
                        best = (row, ratio, leaving)
                else:
                    best = (row, ratio, leaving)
        return None if best is None else best[0]

    def bring_into_basis(self, q: int, r: int, column, cost: Fraction) -> int:
        """tableau/mod.rs:47-60."""
        leaving = self.im.change_basis(r, q, column, cost)
        self.basis_columns.remove(leaving)
        self.basis_columns.add(q)
        self.im.after_basis_change(self.kind)
        return leaving

    # artificial-only (kind/artificial/mod.rs:64-80) ------------------------------------------
    def has_artificial_in_basis(self) -> bool:
        na = self.kind.nr_artificial_variables()
        return any(c < na for c in self.basis_columns)

    def artificial_basis_columns(self):
        na = self.kind.nr_artificial_variables()
        return {c for c in self.basis_columns if c < na}


def is_in_basic_feasible_solution_state(t: Tableau) -> bool:
    """Debug invariant checker, tableau/mod.rs:253-289."""
    if len(t.basis_columns) != t.nr_rows():
        return False
    for i in range(t.nr_rows()):
        j = t.im.basis_indices[i]
        if t.generate_column(j).column != [(i, ONE)]:
            return False
        if t.relative_cost(j) != 0:
            return False
        if t.im.b[i] < 0:
            return False
    return True


# --------------------------------------------------------------------------------------------
# Pivot rules  (strategy/pivot_rule.rs)
# --------------------------------------------------------------------------------------------
class FirstProfitable:
    """pivot_rule.rs:38-57."""
    name = "FirstProfitable"

    def select_primal_pivot_column(self, t: Tableau):
        for j in range(t.nr_columns()):
            if not t.is_in_basis(j):
                c = t.relative_cost(j)
                if c < 0:
                    return j, c
        return None


class FirstProfitableWithMemory:
    """pivot_rule.rs:62-93: search restarts AT the last selected index (now basic, so skipped)."""
    name = "FirstProfitableWithMemory"

    def __init__(self):
        self.last_selected: Optional[int] = None

    def select_primal_pivot_column(self, t: Tableau):
        def find(lo, hi):
            for j in range(lo, hi):
                if not t.is_in_basis(j):
                    c = t.relative_cost(j)
                    if c < 0:
                        return j, c
            return None
        n = t.nr_columns()
        if self.last_selected is None:
            potential = find(0, n)
        else:
            potential = find(self.last_selected, n) or find(0, self.last_selected)
        self.last_selected = None if potential is None else potential[0]
        return potential


class SteepestDescent:
    """pivot_rule.rs:97-126: Dantzig (most negative d_j); strict ``<`` so the lowest j wins ties."""
    name = "SteepestDescent"

    def select_primal_pivot_column(self, t: Tableau):
        smallest = None
        for j in range(t.nr_columns()):
            if t.is_in_basis(j):
                continue
            c = t.relative_cost(j)
            if c < 0 and (smallest is None or c < smallest[1]):
                smallest = (j, c)
        return smallest


# --------------------------------------------------------------------------------------------
# Driver loops  (phase_one.rs, phase_two.rs, two_phase/mod.rs)
# --------------------------------------------------------------------------------------------
TraceFn = Optional[Callable[[dict], None]]


def _pivot(t: Tableau, rule, phase: int, trace: TraceFn, check: bool):
    """One iteration of the loop body shared by phase_one.rs:135-146 and phase_two.rs:32-50.
    Returns 'pivoted' | 'no_candidate' | 'no_row'."""
    if check:
        assert is_in_basic_feasible_solution_state(t)
    sel = rule.select_primal_pivot_column(t)
    if sel is None:
        return "no_candidate"
    q, cost = sel
    column = t.generate_column(q)
    r = t.select_primal_pivot_row(column.column)
    if r is None:
        return "no_row"
    ratio = t.im.b[r] / sparse_get(column.column, r)
    leaving = t.bring_into_basis(q, r, column, cost)
    if trace is not None:
        trace({"phase": phase, "entering": q, "row": r, "leaving": leaving, "d_q": cost,
               "ratio": ratio, "objective": t.objective_function_value()})
    return "pivoted"


def remove_artificial_basis_variables(t: Tableau, trace: TraceFn = None) -> List[int]:
    """phase_one.rs:223-260.  NB: pushes the *artificial index*, not the row (:252)."""
    arts = sorted(t.artificial_basis_columns())
    rows_to_remove: List[int] = []
    na = t.kind.nr_artificial_variables()
    for artificial in arts:
        pivot_row = t.kind.pivot_row_from_artificial(artificial)
        found = None
        for j in range(na, t.nr_columns()):
            if t.is_in_basis(j):
                continue
            cost = t.relative_cost(j)
            if cost != 0:
                continue
            if t.generate_element(pivot_row, j) is not None:
                found = (j, cost)
                break
        if found is not None:
            q, cost = found
            column = t.generate_column(q)
            leaving = t.bring_into_basis(q, pivot_row, column, cost)
            if trace is not None:
                trace({"phase": 1, "entering": q, "row": pivot_row, "leaving": leaving, "d_q": cost,
                       "ratio": ZERO, "objective": t.objective_function_value(), "zero_level": True})
        else:
            rows_to_remove.append(artificial)
    return rows_to_remove


def phase_one_primal(t: Tableau, rule=None, trace: TraceFn = None, check: bool = False,
                     max_iterations: Optional[int] = None):
    """phase_one.rs:125-170.  Returns ('infeasible',) or ('feasible', rank_rows, nr_artificial, carry, basis)."""
    rule = rule or FirstProfitableWithMemory()
    it = 0
    while True:
        if max_iterations is not None and it >= max_iterations:
            return ("iteration_limit",)
        status = _pivot(t, rule, 1, trace, check)
        it += 1
        if status == "no_row":
            raise RuntimeError("Artificial cost can not be unbounded.")  # phase_one.rs:143
        if status == "no_candidate":
            if t.objective_function_value() == 0:
                rows = remove_artificial_basis_variables(t, trace) if t.has_artificial_in_basis() else []
                return ("feasible", rows, t.kind.nr_artificial_variables(), t.im, set(t.basis_columns))
            return ("infeasible",)


def phase_two_primal(t: Tableau, rule=None, trace: TraceFn = None, check: bool = False,
                     max_iterations: Optional[int] = None):
    """phase_two.rs:22-51.  Returns ('optimal', bfs) | ('unbounded',) | ('iteration_limit',)."""
    rule = rule or SteepestDescent()
    it = 0
    while True:
        if max_iterations is not None and it >= max_iterations:
            return ("iteration_limit",)
        status = _pivot(t, rule, 2, trace, check)
        it += 1
        if status == "no_candidate":
            return ("optimal", t.current_bfs())
        if status == "no_row":
            return ("unbounded",)


def solve_relaxation(provider: MatrixData, BI=BasisInverseRows, trace: TraceFn = None, check: bool = False,
                     phase_one_rule=None, phase_two_rule=None, full_initial_basis: bool = False):
    """two_phase/mod.rs:25-76 (and :82-113 when ``full_initial_basis``).

    Returns a dict: status in {'optimal','unbounded','infeasible'}, plus 'bfs', 'objective',
    'tableau' when a phase-2 tableau exists.
    """
    if full_initial_basis:
        pivots = provider.pivot_element_indices()
        im = Carry.from_basis_pivots(BI, pivots, provider)
        t2 = Tableau(im, [c for (_r, c) in pivots], NonArtificial(provider))
    else:
        t1 = Tableau.new_partially_artificial(BI, provider)
        res = phase_one_primal(t1, phase_one_rule, trace, check)
        if res[0] == "infeasible":
            return {"status": "infeasible"}
        _tag, rows, nr_a, im, basis = res
        if rows:
            rr = RemoveRows(provider, rows)
            t2 = Tableau.from_artificial_removing_rows(BI, im, nr_a, basis, rr)
        else:
            t2 = Tableau.from_artificial(im, nr_a, basis, provider)
    out = phase_two_primal(t2, phase_two_rule, trace, check)
    if out[0] == "optimal":
        return {"status": "optimal", "bfs": out[1], "objective": t2.objective_function_value(), "tableau": t2}
    return {"status": out[0], "tableau": t2}
