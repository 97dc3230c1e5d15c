"""ctypes binding of the C ABI (`include/relp_engine.h`, built into `librelp_engine.so`) and the
host-side mirror of the reference interface for the pivot path.

Names follow the reference (file:line under /root/reference/src/algorithm/two_phase/):
  ``Tableau``            tableau/mod.rs:24 -- relative_cost(s), generate_column, generate_element,
                         select_primal_pivot_row, bring_into_basis, current_bfs, objective_function_value
  ``PivotRule`` values   strategy/pivot_rule.rs:38,62,97
  ``phase_one_primal`` / ``phase_two_primal`` / ``solve_relaxation``
                         phase_one.rs:125, phase_two.rs:22, two_phase/mod.rs:30

The HIP library is mandatory.  If it is missing or fails to load this module raises: there is no
CPU fallback (the CPU restatements live in ``oracle/`` and are test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

from .matrix_data import MatrixData

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librelp_engine.so")

# relp_pivot_rule_t
FIRST_PROFITABLE, FIRST_PROFITABLE_WITH_MEMORY, STEEPEST_DESCENT = 0, 1, 2
# relp_outcome_t
RUNNING, OPTIMAL, UNBOUNDED, INFEASIBLE, PHASE_ONE_DONE, NO_ROW_PHASE_ONE = range(6)
OUTCOME_NAMES = {RUNNING: "running", OPTIMAL: "optimal", UNBOUNDED: "unbounded", INFEASIBLE: "infeasible",
                 PHASE_ONE_DONE: "phase_one_done", NO_ROW_PHASE_ONE: "no_row_phase_one"}
# relp_kernel_id_t
(K_PRICE, K_SELECT_COLUMN, K_BUILD_COLUMN, K_FTRAN, K_RATIO, K_UPDATE_VECTORS, K_UPDATE_INVERSE, K_APPLY_W,
 K_UPDATE_W, K_FLUSH, K_FT_RUN) = range(11)
KERNEL_NAMES = ["price", "select_column", "build_column", "ftran", "ratio", "update_vectors", "update_inverse",
                "apply_w", "update_w", "flush", "ft_run"]
ENGINE_REVISED, ENGINE_TABLEAU, ENGINE_LU, ENGINE_AUTO = 0, 1, 2, 3   # relp_engine_kind_t
RATIO_REFERENCE, RATIO_LARGEST_PIVOT = 0, 1           # relp_ratio_rule_t (f64 safeguard; 0 = tableau/mod.rs:229-239)
ARTIFICIAL_REFERENCE, ARTIFICIAL_TEXTBOOK = 0, 1      # relp_artificial_removal_t (0 = phase_one.rs:223-260 literally)
# relp_status_t
E_ARG, E_HIP, E_ZERO_PIVOT, E_SINGULAR, E_STATE, E_UNSUPPORTED, E_ALLOC = -1, -2, -3, -4, -5, -6, -7
FORMAT_CSC, FORMAT_DENSE = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1


class RelpError(RuntimeError):
    pass


class _MatrixData(C.Structure):
    _fields_ = [("nr_normal", C.c_int32), ("nr_eq", C.c_int32), ("nr_range", C.c_int32), ("nr_le", C.c_int32),
                ("nr_ge", C.c_int32), ("format", C.c_int32), ("matrix_memory", C.c_int32),
                ("col_ptr", C.c_void_p), ("row_idx", C.c_void_p), ("values", C.c_void_p),
                ("dense", C.c_void_p), ("dense_ld", C.c_int64),
                ("b", C.c_void_p), ("ranges", C.c_void_p), ("cost", C.c_void_p), ("upper_bound", C.c_void_p)]


class Config(C.Structure):
    """relp_config_t"""
    _fields_ = [("device", C.c_int32), ("phase_one_rule", C.c_int32), ("phase_two_rule", C.c_int32),
                ("tol_cost", C.c_double), ("tol_pivot", C.c_double), ("tol_zero", C.c_double),
                ("tol_tie", C.c_double), ("tol_feas", C.c_double),
                ("poll_interval", C.c_int32), ("trace_capacity", C.c_int32),
                ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("update_block", C.c_int32), ("engine", C.c_int32),
                ("ratio_rule", C.c_int32), ("artificial_removal", C.c_int32),
                ("pivot_rescue", C.c_int32), ("auto_reinversion", C.c_int32)]


# every symbol include/relp_engine.h declares (tests/test_abi.py checks the export list against the header)
_SIGNATURES = {
    "relp_default_config": (None, [C.POINTER(Config)]),
    "relp_robust_config": (None, [C.POINTER(Config)]),
    "relp_engine_kind": (C.c_int32, [C.c_void_p]),
    "relp_robust_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "relp_last_error": (C.c_char_p, [C.c_void_p]),
    "relp_version": (C.c_char_p, []),
    "relp_engine_create": (C.c_int, [C.POINTER(_MatrixData), C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "relp_engine_destroy": (None, [C.c_void_p]),
    "relp_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_select_primal_pivot_column": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                  C.POINTER(C.c_double)]),
    "relp_relative_costs": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_generate_column": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "relp_generate_element": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "relp_select_primal_pivot_row": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "relp_select_primal_pivot_row_of": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "relp_bring_into_basis": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_int32)]),
    "relp_run": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "relp_solve_relaxation": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "relp_from_basis": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_flush": (C.c_int, [C.c_void_p]),
    "relp_update_block": (C.c_int32, [C.c_void_p]),
    "relp_lu_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "relp_lu_lookahead_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "relp_lu_kernel_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "relp_lu_set_device_factorisation": (C.c_int, [C.c_void_p, C.c_int32]),
    "relp_lu_device_factorisation_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "relp_lu_factor_residual": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "relp_lu_phase_cycles": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "relp_basis_inverse_row": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "relp_should_refactor": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "relp_generate_column_of": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "relp_cost_difference_of": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "relp_lu_change_basis": (C.c_int, [C.c_void_p, C.c_int32]),
    "relp_lu_set_factors": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "relp_lu_updates": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "relp_lu_get_update": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "relp_lu_get_upper": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "relp_shard_flush_begin": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "relp_shard_flush_end": (C.c_int, [C.c_void_p]),
    "relp_nr_rows": (C.c_int32, [C.c_void_p]),
    "relp_nr_columns": (C.c_int32, [C.c_void_p]),
    "relp_phase": (C.c_int32, [C.c_void_p]),
    "relp_nr_artificial": (C.c_int32, [C.c_void_p]),
    "relp_get_objective": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "relp_get_b": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_get_minus_pi": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_get_basis_indices": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_get_basis_inverse": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_current_bfs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "relp_get_iterations": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "relp_get_degenerate_pivots": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "relp_get_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.POINTER(C.c_int64)]),
    "relp_check_basis": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "relp_profile_enable": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32]),
    "relp_profile_read": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "relp_synth_fill_dense": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_int64, C.c_void_p]),
    "relp_device_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64]),
    "relp_device_free": (C.c_int, [C.c_void_p]),
    "relp_shard_ranges": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 5),
    "relp_shard_column_range": (None, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "relp_shard_candidate_len": (C.c_int64, [C.c_void_p]),
    "relp_shard_rho_len": (C.c_int64, [C.c_void_p]),
    "relp_shard_price": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_shard_select_column": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "relp_shard_ftran": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_shard_ratio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "relp_shard_update": (C.c_int, [C.c_void_p, C.c_void_p]),
    "relp_poll": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "relp_shard_pivot": (C.c_int, [C.c_void_p]),
    "relp_set_reinversion_interval": (C.c_int, [C.c_void_p, C.c_int64]),
    "relp_reinversions": (C.c_int64, [C.c_void_p]),
    "relp_shard_set_collectives": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "relp_shard_run": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "relp_shard_inject_failure": (C.c_int, [C.c_void_p, C.c_int64]),
    "relp_rccl_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "relp_rccl_attach": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
    "relp_shard_plan": (C.c_int, [C.POINTER(_MatrixData), C.POINTER(Config), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_lib = None


def load_library():
    """Load ``librelp_engine.so``; raises if it is missing (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RelpError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def default_config(**overrides) -> Config:
    cfg = Config()
    load_library().relp_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown config field {k}")
        setattr(cfg, k, v)
    return cfg


def robust_config(**overrides) -> Config:
    """relp_robust_config: the f64 safeguards (largest-pivot ratio rule, textbook artificial removal, pivot rescue, adaptive
    re-inversion) and RELP_ENGINE_AUTO -- no per-file knobs."""
    cfg = Config()
    load_library().relp_robust_config(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown config field {k}")
        setattr(cfg, k, v)
    return cfg


def shard_column_range(nr_normal: int, rank: int, count: int) -> Tuple[int, int]:
    lo, hi = C.c_int32(), C.c_int32()
    load_library().relp_shard_column_range(nr_normal, rank, count, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_plan(provider: MatrixData, config: Config) -> Tuple[int, int]:
    """Structural columns [lo, hi) the rank ``config.shard_rank`` must supply (relp_shard_plan)."""
    md = _MatrixData()
    md.nr_normal, md.nr_eq, md.nr_range = provider.nr_normal, provider.nr_eq, provider.nr_range
    md.nr_le, md.nr_ge = provider.nr_le, provider.nr_ge
    ub = np.ascontiguousarray(provider.upper_bound, dtype=np.float64)
    md.upper_bound = ub.ctypes.data
    lo, hi = C.c_int32(), C.c_int32()
    st = load_library().relp_shard_plan(C.byref(md), C.byref(config), C.byref(lo), C.byref(hi))
    if st != 0:
        raise RelpError(f"relp_shard_plan failed ({st})")
    return lo.value, hi.value


class Tableau:
    """Device-resident ``Tableau<Carry<f64, BasisInverseRows<f64>>, K>`` (tableau/mod.rs:24-38)."""

    def __init__(self, provider: MatrixData, config: Optional[Config] = None, *, device_dense_ptr: Optional[int] = None,
                 device_dense_ld: Optional[int] = None, **config_overrides):
        """``Tableau::<_, Partially<_>>::new(provider)`` (kind/artificial/partially.rs:125).

        ``device_dense_ptr``: address of a column-major dense matrix already in HBM (e.g. a torch
        tensor's ``data_ptr()``), holding this shard's structural columns.
        """
        self._lib = load_library()
        cfg = config if config is not None else default_config()
        for k, v in config_overrides.items():
            setattr(cfg, k, v)
        self.config = cfg
        keep = []

        def ptr(a, dtype):
            arr = np.ascontiguousarray(a, dtype=dtype)
            keep.append(arr)
            return arr.ctypes.data

        md = _MatrixData()
        md.nr_normal, md.nr_eq, md.nr_range = provider.nr_normal, provider.nr_eq, provider.nr_range
        md.nr_le, md.nr_ge = provider.nr_le, provider.nr_ge
        if device_dense_ptr is not None:
            md.format, md.matrix_memory = FORMAT_DENSE, MEM_DEVICE
            md.dense = device_dense_ptr
            md.dense_ld = device_dense_ld or provider.nr_constraints
        elif provider.dense is not None:
            md.format, md.matrix_memory = FORMAT_DENSE, MEM_HOST
            dense = np.asfortranarray(provider.dense, dtype=np.float64)
            keep.append(dense)
            md.dense = dense.ctypes.data
            md.dense_ld = dense.shape[0]
        else:
            md.format, md.matrix_memory = FORMAT_CSC, MEM_HOST
            md.col_ptr = ptr(provider.col_ptr, np.int64)
            md.row_idx = ptr(provider.row_idx, np.int32)
            md.values = ptr(provider.values, np.float64)
        md.b = ptr(provider.b, np.float64)
        md.ranges = ptr(provider.ranges, np.float64)
        md.cost = ptr(provider.cost, np.float64)
        md.upper_bound = ptr(provider.upper_bound, np.float64)
        h = C.c_void_p()
        st = self._lib.relp_engine_create(C.byref(md), C.byref(cfg), C.byref(h))
        self._h = h
        if st != 0:
            msg = self._lib.relp_last_error(h).decode() if h else "allocation failed"
            if h:
                self._lib.relp_engine_destroy(h)
            self._h = None
            raise RelpError(f"relp_engine_create failed ({st}): {msg}")
        self.provider = provider
        del keep

    # -- plumbing ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.relp_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != 0:
            raise RelpError(f"relp call failed ({st}): {self._lib.relp_last_error(self._h).decode()}")

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream: int):
        self._ck(self._lib.relp_set_stream(self._h, hip_stream))

    # -- sizes / kind -----------------------------------------------------------------------
    def nr_rows(self) -> int:
        return self._lib.relp_nr_rows(self._h)

    def nr_columns(self) -> int:
        return self._lib.relp_nr_columns(self._h)

    def nr_artificial_variables(self) -> int:
        return self._lib.relp_nr_artificial(self._h)

    @property
    def phase(self) -> int:
        return self._lib.relp_phase(self._h)

    # -- one pivot (tableau/mod.rs:47-247, pivot_rule.rs:25) ---------------------------------
    def select_primal_pivot_column(self, rule: int) -> Optional[Tuple[int, float]]:
        found, col, cost = C.c_int32(), C.c_int32(), C.c_double()
        self._ck(self._lib.relp_select_primal_pivot_column(self._h, rule, C.byref(found), C.byref(col), C.byref(cost)))
        return (col.value, cost.value) if found.value else None

    def relative_costs(self) -> np.ndarray:
        out = np.zeros(self.nr_columns())
        self._ck(self._lib.relp_relative_costs(self._h, out.ctypes.data))
        return out

    def relative_cost(self, j: int) -> float:
        return float(self.relative_costs()[j])

    def generate_column(self, j: int) -> np.ndarray:
        out = np.zeros(self.nr_rows())
        self._ck(self._lib.relp_generate_column(self._h, j, out.ctypes.data))
        return out

    def generate_element(self, i: int, j: int) -> float:
        v = C.c_double()
        self._ck(self._lib.relp_generate_element(self._h, i, j, C.byref(v)))
        return v.value

    def set_reinversion_interval(self, pivots: int) -> None:
        """Revised engine: rebuild B^-1, b, -pi from the basis columns every `pivots` basis changes (0 = never)."""
        self._ck(self._lib.relp_set_reinversion_interval(self._h, int(pivots)))

    def reinversions(self) -> int:
        return int(self._lib.relp_reinversions(self._h))

    def select_primal_pivot_row(self, column=None) -> Optional[int]:
        """tableau/mod.rs:221-247.  Without an argument: on the last generated column (device resident);
        with a dense column of m entries: the reference's signature."""
        found, row = C.c_int32(), C.c_int32()
        if column is None:
            self._ck(self._lib.relp_select_primal_pivot_row(self._h, C.byref(found), C.byref(row)))
        else:
            col = np.ascontiguousarray(column, dtype=np.float64)
            if col.shape != (self.nr_rows(),):
                raise ValueError("column must have nr_rows() entries")
            self._ck(self._lib.relp_select_primal_pivot_row_of(self._h, col.ctypes.data_as(C.POINTER(C.c_double)),
                                                              C.byref(found), C.byref(row)))
        return row.value if found.value else None

    def bring_into_basis(self, column: int, row: int, cost: float) -> int:
        leaving = C.c_int32()
        self._ck(self._lib.relp_bring_into_basis(self._h, column, row, cost, C.byref(leaving)))
        return leaving.value

    # -- loops ------------------------------------------------------------------------------
    def run(self, max_iters: int) -> Tuple[int, int]:
        """Up to ``max_iters`` basis changes of the current phase; returns (iterations, outcome)."""
        done, oc = C.c_int64(), C.c_int32()
        self._ck(self._lib.relp_run(self._h, max_iters, C.byref(done), C.byref(oc)))
        return done.value, oc.value

    def solve_relaxation(self, max_iters: int = 1 << 40) -> int:
        oc = C.c_int32()
        self._ck(self._lib.relp_solve_relaxation(self._h, max_iters, C.byref(oc)))
        return oc.value

    def flush(self) -> None:
        """Fold pending deferred updates into the explicit inverse."""
        self._ck(self._lib.relp_flush(self._h))

    def update_block(self) -> int:
        return self._lib.relp_update_block(self._h)

    def engine_kind(self) -> int:
        """The engine in use (RELP_ENGINE_AUTO resolved at create)."""
        return int(self._lib.relp_engine_kind(self._h))

    def robust_stats(self) -> dict:
        out = (C.c_int64 * 4)()
        self._ck(self._lib.relp_robust_stats(self._h, out))
        return dict(zip(("small_pivots", "columns_barred", "confirmations", "reinversion_interval"), (int(v) for v in out)))

    def lu_stats(self) -> dict:
        """Factor statistics of the LU engine (relp_lu_stats)."""
        out = (C.c_int64 * 8)()
        self._ck(self._lib.relp_lu_stats(self._h, out))
        keys = ("refactorisations", "m", "nnz_l", "nnz_u", "levels_l", "levels_u", "levels_ut", "levels_lt")
        stats = dict(zip(keys, (int(v) for v in out)))
        la = (C.c_int64 * 4)()
        self._ck(self._lib.relp_lu_lookahead_stats(self._h, la))
        stats.update(lookahead_installs=int(la[0]), replayed_changes=int(la[1]), lookahead=int(la[2]), fuse_lanes=int(la[3]))
        return stats

    def lu_kernel_layout(self) -> dict:
        """How the LU engine's pivot loop runs (relp_lu_kernel_layout): persistent kernel or product-form fallback, the
        kernel's layout (0 all in LDS, 1 big, 2 nothing per row in LDS), dense-tail slots, grid PRICE."""
        out = (C.c_int32 * 4)()
        self._ck(self._lib.relp_lu_kernel_layout(self._h, out))
        return {"persistent_kernel": bool(out[0]), "layout": int(out[1]), "tail_slots": int(out[2]), "grid_price": bool(out[3])}

    def lu_set_device_factorisation(self, on: bool) -> None:
        """Refactorise on the device from now on (relp_lu_set_device_factorisation; LUDecomposition::invert)."""
        self._ck(self._lib.relp_lu_set_device_factorisation(self._h, 1 if on else 0))

    def lu_factor_residual(self) -> float:
        """max |P B Q - L U| for the factors in use (relp_lu_factor_residual)."""
        out = C.c_double()
        self._ck(self._lib.relp_lu_factor_residual(self._h, C.byref(out)))
        return out.value

    def lu_device_factorisation_stats(self) -> dict:
        out = (C.c_int64 * 6)()
        self._ck(self._lib.relp_lu_device_factorisation_stats(self._h, out))
        keys = ("enabled", "device_factorisations", "host_fallbacks", "kernel_us", "last_bump", "last_peeled")
        return dict(zip(keys, (int(v) for v in out)))

    # -- BasisInverse surface (carry/mod.rs:68-157) ---------------------------------------------
    def basis_inverse_row(self, row: int) -> np.ndarray:
        out = np.zeros(self.nr_rows())
        self._ck(self._lib.relp_basis_inverse_row(self._h, row, out.ctypes.data))
        return out

    def should_refactor(self) -> bool:
        v = C.c_int32()
        self._ck(self._lib.relp_should_refactor(self._h, C.byref(v)))
        return bool(v.value)

    @staticmethod
    def _sparse(column):
        idx = np.ascontiguousarray([i for i, _ in column], dtype=np.int32)
        val = np.ascontiguousarray([v for _, v in column], dtype=np.float64)
        return idx, val

    def generate_column_of(self, column) -> np.ndarray:
        """`BasisInverse::generate_column(original_column)` for sorted (row, value) pairs."""
        idx, val = self._sparse(column)
        out = np.zeros(self.nr_rows())
        self._ck(self._lib.relp_generate_column_of(self._h, idx.ctypes.data, val.ctypes.data, len(idx), out.ctypes.data))
        return out

    def cost_difference_of(self, column) -> float:
        idx, val = self._sparse(column)
        v = C.c_double()
        self._ck(self._lib.relp_cost_difference_of(self._h, idx.ctypes.data, val.ctypes.data, len(idx), C.byref(v)))
        return v.value

    def lu_change_basis(self, pivot_row_index: int) -> None:
        self._ck(self._lib.relp_lu_change_basis(self._h, pivot_row_index))

    def lu_set_factors(self, lower_columns, upper_columns) -> None:
        """`LUDecomposition { lower_triangular, upper_triangular, .. }` literally (P = Q = I): lists of columns of
        (row, value) pairs; L without its unit diagonal (m - 1 or m columns), U with its diagonal."""
        m = self.nr_rows()

        def csc(cols):
            cols = list(cols) + [[] for _ in range(m - len(cols))]
            ptr = np.zeros(m + 1, dtype=np.int64)
            for j, c in enumerate(cols):
                ptr[j + 1] = ptr[j] + len(c)
            idx = np.ascontiguousarray([i for c in cols for i, _ in c] or [0], dtype=np.int32)
            val = np.ascontiguousarray([v for c in cols for _, v in c] or [0.0], dtype=np.float64)
            return ptr, idx, val
        lp, li, lv = csc(lower_columns)
        up, ui, uv = csc(upper_columns)
        self._ck(self._lib.relp_lu_set_factors(self._h, lp.ctypes.data, li.ctypes.data, lv.ctypes.data, up.ctypes.data,
                                               ui.ctypes.data, uv.ctypes.data))

    def lu_updates(self):
        """[(pivot, [(position, value), ...]), ...]: the reference's `updates` (EtaFile values + RotateToBack index)."""
        n = C.c_int32()
        self._ck(self._lib.relp_lu_updates(self._h, C.byref(n)))
        m = self.nr_rows()
        out = []
        for k in range(n.value):
            pivot, nnz = C.c_int32(), C.c_int32()
            idx, val = np.zeros(m, dtype=np.int32), np.zeros(m)
            self._ck(self._lib.relp_lu_get_update(self._h, k, C.byref(pivot), idx.ctypes.data, val.ctypes.data, m, C.byref(nnz)))
            out.append((pivot.value, list(zip(idx[:nnz.value].tolist(), val[:nnz.value].tolist()))))
        return out

    def lu_upper(self):
        """The reference's `upper_triangular`: list of columns of (row, value), diagonal last."""
        m = self.nr_rows()
        cap = m * (m + 1) // 2 + m
        ptr, idx, val, nnz = np.zeros(m + 1, dtype=np.int64), np.zeros(cap, dtype=np.int32), np.zeros(cap), C.c_int64()
        self._ck(self._lib.relp_lu_get_upper(self._h, ptr.ctypes.data, idx.ctypes.data, val.ctypes.data, cap, C.byref(nnz)))
        return [list(zip(idx[ptr[j]:ptr[j + 1]].tolist(), val[ptr[j]:ptr[j + 1]].tolist())) for j in range(m)]

    PHASE_NAMES = ("price", "scatter", "l_solve", "eta_forward", "spike_push", "u_solve", "ratio", "b_update", "u_bar",
                   "ut_solve", "compact", "eta_backward", "lt_solve", "vectors", "load_store", "stage_and_level0")

    def lu_phase_cycles(self) -> dict:
        out = (C.c_int64 * 16)()
        self._ck(self._lib.relp_lu_phase_cycles(self._h, out))
        return dict(zip(self.PHASE_NAMES, (int(v) for v in out)))

    def from_basis(self, basis_columns) -> None:
        arr = np.ascontiguousarray(basis_columns, dtype=np.int32)
        self._ck(self._lib.relp_from_basis(self._h, arr.ctypes.data))

    # -- state ------------------------------------------------------------------------------
    def objective_function_value(self) -> float:
        v = C.c_double()
        self._ck(self._lib.relp_get_objective(self._h, C.byref(v)))
        return v.value

    def b(self) -> np.ndarray:
        out = np.zeros(self.nr_rows())
        self._ck(self._lib.relp_get_b(self._h, out.ctypes.data))
        return out

    def minus_pi(self) -> np.ndarray:
        out = np.zeros(self.nr_rows())
        self._ck(self._lib.relp_get_minus_pi(self._h, out.ctypes.data))
        return out

    def basis_indices(self) -> np.ndarray:
        out = np.zeros(self.nr_rows(), dtype=np.int32)
        self._ck(self._lib.relp_get_basis_indices(self._h, out.ctypes.data))
        return out

    def basis_inverse(self) -> np.ndarray:
        m = self.nr_rows()
        out = np.zeros((m, m))
        self._ck(self._lib.relp_get_basis_inverse(self._h, out.ctypes.data))
        return out

    def current_bfs(self) -> List[Tuple[int, float]]:
        m = self.nr_rows()
        cols, vals, cnt = np.zeros(m, dtype=np.int32), np.zeros(m), C.c_int32()
        self._ck(self._lib.relp_current_bfs(self._h, cols.ctypes.data, vals.ctypes.data, m, C.byref(cnt)))
        return list(zip(cols[:cnt.value].tolist(), vals[:cnt.value].tolist()))

    def degenerate_pivots(self) -> int:
        """Pivots so far with ratio exactly 0 (the basis changed, b did not)."""
        out = C.c_int64()
        self._ck(self._lib.relp_get_degenerate_pivots(self._h, C.byref(out)))
        return out.value

    def iterations(self) -> int:
        v = C.c_int64()
        self._ck(self._lib.relp_get_iterations(self._h, C.byref(v)))
        return v.value

    def trace(self) -> List[Tuple[int, int, int, int]]:
        cap = int(self.config.trace_capacity)
        arrs = [np.zeros(max(cap, 1), dtype=np.int32) for _ in range(4)]
        cnt = C.c_int64()
        self._ck(self._lib.relp_get_trace(self._h, *[a.ctypes.data for a in arrs], cap, C.byref(cnt)))
        k = min(cnt.value, cap)
        return list(zip(*(a[:k].tolist() for a in arrs)))

    def check_basis(self) -> Tuple[float, float, float]:
        """``is_in_basic_feasible_solution_state`` (tableau/mod.rs:253-289) as three residuals."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._ck(self._lib.relp_check_basis(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- profiling --------------------------------------------------------------------------
    def profile_enable(self, enable: bool, max_launches: int = 0, sample_every: int = 1):
        self._ck(self._lib.relp_profile_enable(self._h, int(enable), max_launches, sample_every))

    def profile_read(self):
        out = {}
        for kid, name in enumerate(KERNEL_NAMES):
            n, ms = C.c_int64(), C.c_double()
            self._ck(self._lib.relp_profile_read(self._h, kid, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out


def phase_one_primal(tableau: Tableau, max_iters: int = 1 << 40) -> int:
    """phase_one::primal (phase_one.rs:125-170)."""
    if tableau.phase != 1:
        raise RelpError("tableau is not artificial")
    return tableau.run(max_iters)[1]


def phase_two_primal(tableau: Tableau, max_iters: int = 1 << 40) -> int:
    """phase_two::primal (phase_two.rs:22-51)."""
    if tableau.phase != 2:
        raise RelpError("tableau still has artificial variables")
    return tableau.run(max_iters)[1]


def solve_relaxation(provider: MatrixData, **config_overrides):
    """SolveRelaxation::solve_relaxation (two_phase/mod.rs:30-76).  Returns (outcome, tableau)."""
    t = Tableau(provider, **config_overrides)
    return t.solve_relaxation(), t


# ---- beyond the reference: one answer, checked, out of several attempts -------------------------------------------------------
# The reference has ONE configuration, solves the data as read and does not check what it returns; in f64 its literal rules miss optima
# the safeguards of relp_robust_config reach, and the other way round, and the PILOT family needs scaling while six other files of its
# Netlib directory are LOST by scaling (absolute tolerances on scaled rows).  `solve_verified` is the union as a procedure: legs of
# (data as read | scaled by MatrixData.scaled, configuration, engine) in a fixed order, each with a pivot and a time budget, and an
# outcome stands only when it verifies -- `optimal` by the residuals of relp_check_basis (B^-1 B = I, basis columns are unit columns,
# b >= 0: the wrong optima of the sweeps all sit on a basis with some b_i <= -0.07), `infeasible` / `unbounded` only when every leg
# has run, none reached a verified optimum, and two engines said so on the data AS READ from a state that passes the first two checks
# (on scaled data two engines agreed on a wrong `infeasible` for WOODW; PILOT87 ends `infeasible` on an exploded tableau).
VERIFY_IDENTITY, VERIFY_BASIC, VERIFY_MIN_B = 1e-5, 1e-3, -1e-6
# ("robust-dantzig": the safeguards with PivotRule::SteepestDescent -- the largest coefficient -- in phase 1 as well, where the
# reference takes FirstProfitableWithMemory: TUFF, DEGEN3, CYCLE cycle for 600,000 pivots under the reference's phase-1 rule, under
# Bland's too, and end after 1,708 / 11,775 / 49,326 pivots with this one; PILOT87 reaches Netlib's 301.71072827 with it.)
VERIFIED_LEGS = (("read", "robust", ENGINE_LU), ("scaled", "robust", ENGINE_LU), ("scaled", "robust-dantzig", ENGINE_LU),
                 ("scaled", "robust-dantzig", ENGINE_TABLEAU), ("read", "robust-dantzig", ENGINE_LU), ("scaled", "robust", ENGINE_REVISED),
                 ("read", "robust", ENGINE_REVISED), ("read", "robust", ENGINE_TABLEAU), ("scaled", "robust", ENGINE_TABLEAU),
                 ("read", "default", ENGINE_LU), ("read", "default", ENGINE_REVISED), ("read", "default", ENGINE_TABLEAU),
                 # (the long leg: twelve times the budgets.  DFL001, 5,934 x 12,092: phase 1 takes 920,000 pivots and the optimum
                 # 1,990,000 -- 230 s on the tableau engine -- where every other leg is cut off)
                 ("read", "robust-dantzig", ENGINE_TABLEAU, 12))
_ENGINE_NAMES = {ENGINE_REVISED: "revised", ENGINE_TABLEAU: "tableau", ENGINE_LU: "lu"}


def solve_verified(provider: MatrixData, pivots_per_leg: Optional[int] = None, seconds_per_leg: float = 60.0, legs=VERIFIED_LEGS):
    """Returns (outcome, tableau or None, report).  `report["verified"]` says whether the outcome passed its check; `report["legs"]`
    lists what every leg did.  The tableau of the leg that was accepted is returned open (close it); the others are closed.  When
    the accepted leg ran on scaled data (`report["scaled"]`) the objective value is still the provider's; a solution is brought
    back by `provider.unscale_bfs(t.current_bfs(), report["row_scale"], report["column_scale"])`."""
    import time
    budget = pivots_per_leg if pivots_per_leg is not None else 30 * (int(provider.nr_rows) + int(provider.nr_columns))
    report = {"legs": [], "verified": False, "scaled": False}
    scaled = None
    said = {}                                           # infeasible / unbounded -> engines that said so on the data as read
    last = RUNNING
    for entry in legs:
        data, cfg_name, kind = entry[:3]
        times = entry[3] if len(entry) > 3 else 1                # (a long leg: `times` the pivot and time budgets)
        leg = {"data": data, "config": cfg_name, "engine": _ENGINE_NAMES[kind]}
        report["legs"].append(leg)
        t0 = time.perf_counter()
        if data == "scaled" and scaled is None:
            scaled = provider.scaled()
        if cfg_name == "robust":
            cfg = robust_config(engine=kind)
        elif cfg_name == "robust-dantzig":
            cfg = robust_config(engine=kind, phase_one_rule=STEEPEST_DESCENT, phase_two_rule=STEEPEST_DESCENT)
        else:
            cfg = default_config(engine=kind)
        try:
            t = Tableau(scaled[0] if data == "scaled" else provider, config=cfg)
        except RelpError as e:
            leg["outcome"] = "create_failed: " + str(e)[:120]
            continue
        try:
            oc, total = RUNNING, 0
            while total < times * budget and time.perf_counter() - t0 < times * seconds_per_leg:
                done, oc = t.run(min(20000, times * budget - total))
                total += done
                if oc not in (RUNNING, PHASE_ONE_DONE):
                    break
            leg["pivots"] = total
            leg["outcome"] = OUTCOME_NAMES.get(oc, str(oc)) if oc not in (RUNNING, PHASE_ONE_DONE) else "limit"
            last = oc
            if oc == OPTIMAL:
                ident, basic, min_b = t.check_basis()
                leg["check_basis"] = [ident, basic, min_b]
                if ident <= VERIFY_IDENTITY and basic <= VERIFY_BASIC and min_b >= VERIFY_MIN_B:
                    report["verified"] = True
                    report["scaled"] = data == "scaled"
                    if data == "scaled":
                        report["row_scale"], report["column_scale"] = scaled[1], scaled[2]
                    return OPTIMAL, t, report
            elif oc in (INFEASIBLE, UNBOUNDED) and data == "read":
                # (a claim counts only from a state that is still a basis: PILOT87 on the tableau engine ends `infeasible` with
                # a phase-1 objective of 1e302 and then inf)
                ident, basic, min_b = t.check_basis()
                leg["check_basis"] = [ident, basic, min_b]
                if ident <= VERIFY_IDENTITY and basic <= VERIFY_BASIC and np.isfinite(t.objective_function_value()):
                    said.setdefault(oc, set()).add(kind)
        except RelpError as e:
            leg["outcome"] = "error: " + str(e)[:120]
        t.close()
    for oc, engines in said.items():
        if len(engines) >= 2:
            report["verified"] = True
            return oc, None, report
    return last, None, report
