"""ctypes binding of the C (f64) CPU oracle ``oracle/librelp_oracle.so`` (built by oracle/Makefile).

TEST INFRASTRUCTURE ONLY (see oracle/relp_oracle.h).  ``problem`` arguments are duck-typed: any
object with the MatrixData array attributes (nr_normal, nr_eq, nr_range, nr_le, nr_ge, col_ptr,
row_idx, values, b, ranges, cost, upper_bound).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librelp_oracle.so")

RULE_FIRST_PROFITABLE, RULE_FIRST_PROFITABLE_WITH_MEMORY, RULE_STEEPEST_DESCENT = 0, 1, 2
STATUS = {0: "running", 1: "optimal", 2: "unbounded", 3: "infeasible", 4: "iteration_limit",
          5: "phase_one_done", -1: "error"}


class _MatrixData(C.Structure):
    _fields_ = [("nr_normal", C.c_int32), ("nr_eq", C.c_int32), ("nr_range", C.c_int32),
                ("nr_le", C.c_int32), ("nr_ge", C.c_int32),
                ("col_ptr", C.c_void_p), ("row_idx", C.c_void_p), ("values", C.c_void_p),
                ("b", C.c_void_p), ("ranges", C.c_void_p), ("cost", C.c_void_p), ("upper_bound", C.c_void_p)]


class _Config(C.Structure):
    _fields_ = [("tol_cost", C.c_double), ("tol_pivot", C.c_double), ("tol_zero", C.c_double),
                ("tol_tie", C.c_double), ("tol_feas", C.c_double),
                ("phase_one_rule", C.c_int32), ("phase_two_rule", C.c_int32),
                ("ratio_rule", C.c_int32), ("artificial_removal", C.c_int32),
                ("basis_inverse", C.c_int32), ("refactor_after", C.c_int32), ("lu_threshold", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "relp_f64.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(_MatrixData), C.POINTER(_Config)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_run.restype = C.c_int
        L.oracle_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.POINTER(C.c_int64)]
        for name in ("oracle_m", "oracle_n", "oracle_phase", "oracle_nr_artificial", "oracle_nr_filtered_rows",
                     "oracle_nr_zero_level_pivots", "oracle_nr_position_exchanges"):
            getattr(L, name).restype = C.c_int32
            getattr(L, name).argtypes = [C.c_void_p]
        L.oracle_objective.restype = C.c_double
        L.oracle_objective.argtypes = [C.c_void_p]
        for name in ("oracle_get_b", "oracle_get_minus_pi", "oracle_get_basis", "oracle_get_basis_inverse",
                     "oracle_get_filtered_rows"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_basis_inverse_nnz.restype = C.c_int64
        L.oracle_basis_inverse_nnz.argtypes = [C.c_void_p]
        L.oracle_lu_stats.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


DEFAULT_TOLERANCES = dict(tol_cost=1e-7, tol_pivot=1e-5, tol_zero=1e-11, tol_tie=1e-9, tol_feas=1e-7)


class OracleF64:
    """One f64 CPU solve state (`Tableau<Carry<f64, BasisInverseRows<f64>>, _>`)."""

    def __init__(self, problem, phase_one_rule=RULE_FIRST_PROFITABLE_WITH_MEMORY,
                 phase_two_rule=RULE_STEEPEST_DESCENT, ratio_rule=0, artificial_removal=0, basis_inverse=0, refactor_after=10,
                 lu_threshold=0.0, **tolerances):
        """basis_inverse: 0 = `BasisInverseRows`, 1 = `LUDecomposition` with its update file (what src/bin/main.rs:52 runs),
        re-inverted from the basis columns when more than `refactor_after` updates are pending (lower_upper/mod.rs:199-202: 10)."""
        tol = dict(DEFAULT_TOLERANCES)
        tol.update(tolerances)
        self._keep = [np.ascontiguousarray(problem.col_ptr, dtype=np.int64),
                      np.ascontiguousarray(problem.row_idx, dtype=np.int32),
                      np.ascontiguousarray(problem.values, dtype=np.float64),
                      np.ascontiguousarray(problem.b, dtype=np.float64),
                      np.ascontiguousarray(problem.ranges, dtype=np.float64),
                      np.ascontiguousarray(problem.cost, dtype=np.float64),
                      np.ascontiguousarray(problem.upper_bound, dtype=np.float64)]
        md = _MatrixData(problem.nr_normal, problem.nr_eq, problem.nr_range, problem.nr_le, problem.nr_ge,
                         *[a.ctypes.data for a in self._keep])
        cfg = _Config(tol["tol_cost"], tol["tol_pivot"], tol["tol_zero"], tol["tol_tie"], tol["tol_feas"],
                      phase_one_rule, phase_two_rule, ratio_rule, artificial_removal, basis_inverse, refactor_after, lu_threshold)
        self._h = lib().oracle_create(C.byref(md), C.byref(cfg))
        self.trace = []          # (phase, entering, row, leaving)

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, max_iters: int = 1 << 40, through_phases: bool = True, record: bool = True) -> str:
        cap = int(min(max_iters, 1 << 22)) if record else 0
        arrs = [np.zeros(cap, dtype=np.int32) for _ in range(4)] if record else [None] * 4
        n_done = C.c_int64(0)
        ptrs = [a.ctypes.data if a is not None else None for a in arrs]
        st = lib().oracle_run(self._h, int(max_iters), int(through_phases), *ptrs, cap, C.byref(n_done))
        self.last_n_done = n_done.value
        if record:
            k = min(n_done.value, cap)
            self.trace.extend(zip(*(a[:k].tolist() for a in arrs)))
        return STATUS[st]

    m = property(lambda self: lib().oracle_m(self._h))
    n = property(lambda self: lib().oracle_n(self._h))
    phase = property(lambda self: lib().oracle_phase(self._h))
    nr_artificial = property(lambda self: lib().oracle_nr_artificial(self._h))
    nr_zero_level_pivots = property(lambda self: lib().oracle_nr_zero_level_pivots(self._h))
    nr_position_exchanges = property(lambda self: lib().oracle_nr_position_exchanges(self._h))

    def filtered_rows(self):
        """Rows removed as redundant at the phase switch."""
        k = lib().oracle_nr_filtered_rows(self._h)
        out = np.zeros(max(k, 1), dtype=np.int32)
        lib().oracle_get_filtered_rows(self._h, out.ctypes.data)
        return out[:k].tolist()
    objective = property(lambda self: lib().oracle_objective(self._h))
    basis_inverse_nnz = property(lambda self: lib().oracle_basis_inverse_nnz(self._h))

    def lu_stats(self):
        out = np.zeros(4, dtype=np.int64)
        lib().oracle_lu_stats(self._h, out.ctypes.data)
        return dict(zip(("refactorisations", "updates_pending", "nnz_l", "nnz_u"), (int(v) for v in out)))

    def _vec(self, fn, dtype, n):
        out = np.zeros(n, dtype=dtype)
        getattr(lib(), fn)(self._h, out.ctypes.data)
        return out

    def b(self):
        return self._vec("oracle_get_b", np.float64, self.m)

    def minus_pi(self):
        return self._vec("oracle_get_minus_pi", np.float64, self.m)

    def basis(self):
        return self._vec("oracle_get_basis", np.int32, self.m)

    def basis_inverse(self):
        m = self.m
        return self._vec("oracle_get_basis_inverse", np.float64, m * m).reshape(m, m)


class LUF64:
    """`LUDecomposition` of the C oracle by itself (oracle/relp_f64_lu.h), for the reference's own LU known answers."""

    def __init__(self, handle, m):
        self._h, self.m = handle, m

    @staticmethod
    def _csc(cols, m):
        cols = list(cols) + [[] for _ in range(m - len(cols))]
        ptr = np.zeros(m + 1, dtype=np.int64)
        for j, c in enumerate(cols):
            ptr[j + 1] = ptr[j] + len(c)
        idx = np.ascontiguousarray([i for c in cols for i, _ in c] or [0], dtype=np.int32)
        val = np.ascontiguousarray([float(v) for c in cols for _, v in c] or [0.0], dtype=np.float64)
        return ptr, idx, val

    @classmethod
    def _setup(cls):
        L = lib()
        if not getattr(L, "_lu_ready", False):
            L.oracle_lu_from_triangles.restype = C.c_void_p
            L.oracle_lu_from_triangles.argtypes = [C.c_int32] + [C.c_void_p] * 6
            L.oracle_lu_invert.restype = C.c_void_p
            L.oracle_lu_invert.argtypes = [C.c_int32] + [C.c_void_p] * 3
            L.oracle_lu_destroy.argtypes = [C.c_void_p]
            L.oracle_lu_generate_column.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
            L.oracle_lu_change_basis.restype = C.c_int
            L.oracle_lu_change_basis.argtypes = [C.c_void_p, C.c_int32]
            L.oracle_lu_basis_inverse_row.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
            L.oracle_lu_nr_updates.restype = C.c_int32
            L.oracle_lu_nr_updates.argtypes = [C.c_void_p]
            L.oracle_lu_get_update.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
            L.oracle_lu_get_factor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            L._lu_ready = True
        return L

    @classmethod
    def from_triangles(cls, m, lower_columns, upper_columns):
        L = cls._setup()
        lp, li, lv = cls._csc(lower_columns, m)
        up, ui, uv = cls._csc(upper_columns, m)
        return cls(L.oracle_lu_from_triangles(m, lp.ctypes.data, li.ctypes.data, lv.ctypes.data, up.ctypes.data, ui.ctypes.data, uv.ctypes.data), m)

    @classmethod
    def identity(cls, m):
        return cls.from_triangles(m, [[] for _ in range(m)], [[(i, 1.0)] for i in range(m)])

    @classmethod
    def invert(cls, columns):
        L = cls._setup()
        m = len(columns)
        p, i, v = cls._csc(columns, m)
        h = L.oracle_lu_invert(m, p.ctypes.data, i.ctypes.data, v.ctypes.data)
        return cls(h, m) if h else None

    def close(self):
        if self._h:
            lib().oracle_lu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def generate_column(self, column):
        idx = np.ascontiguousarray([i for i, _ in column] or [0], dtype=np.int32)
        val = np.ascontiguousarray([float(v) for _, v in column] or [0.0], dtype=np.float64)
        out = np.zeros(self.m)
        lib().oracle_lu_generate_column(self._h, len(column), idx.ctypes.data, val.ctypes.data, out.ctypes.data)
        return out

    def change_basis(self, pivot_row_index):
        return bool(lib().oracle_lu_change_basis(self._h, pivot_row_index))

    def basis_inverse_row(self, row):
        out = np.zeros(self.m)
        lib().oracle_lu_basis_inverse_row(self._h, row, out.ctypes.data)
        return out

    def updates(self):
        out = []
        for k in range(lib().oracle_lu_nr_updates(self._h)):
            pivot, vals = C.c_int32(), np.zeros(self.m)
            lib().oracle_lu_get_update(self._h, k, C.byref(pivot), vals.ctypes.data)
            out.append((pivot.value, vals))
        return out

    def factor(self, which):
        """(dense m x m factor, row permutation forward, column permutation forward); which = "lower" / "upper"."""
        a = np.zeros((self.m, self.m))
        rf, cf = np.zeros(self.m, dtype=np.int32), np.zeros(self.m, dtype=np.int32)
        lib().oracle_lu_get_factor(self._h, 1 if which == "upper" else 0, a.ctypes.data, rf.ctypes.data, cf.ctypes.data)
        return a, rf, cf
