"""`GeneralForm` -> standard form -> `MatrixData`, and solution reconstruction (SURVEY.md section
8f rows 1 and 3).  One-shot, serial CPU code in exact ``Fraction`` arithmetic; it fixes the row and
column order and the values the pivot engine sees.

Restates (file:line under /root/reference/src/):
  io/mps/convert.rs:50-72, 144-304        MPS -> GeneralForm (bounds: UP keeps the implied 0 lower bound
                                          only if no other lower bound was given; MI: upper := 0; PL: lower := 0)
  io/mps/convert.rs:313-500               ranges / merged right-hand sides
  data/linear_program/general_form/mod.rs:307-314   standardize = presolve; transform_variables;
                                                    make_b_non_negative; make_minimization_problem
  general_form/mod.rs:488-569             split free variables, flip upper-only variables, shift lower bounds to 0
  general_form/mod.rs:574-613             b >= 0 (Range(r): b := r - b), maximisation -> negated costs
  general_form/mod.rs:633-698             stable partition of the rows into [== | range | <= | >=]
  general_form/mod.rs:259-285             MatrixData::new(...)
  general_form/mod.rs:817-942             reshift, un-flip, recombine x+ - x-, objective = sum c_j x_j + fixed cost
                                          (NOT negated back for maximisation inputs, :861-863)

Presolve (general_form/mod.rs:333-478, presolve/**) is restated in `presolve.py`; `standardize()` runs it
first like the reference (`standardize(presolve=False)` keeps the un-presolved problem, which has the
same optimum but another `MatrixData`).
"""
from __future__ import annotations

from dataclasses import dataclass
from fractions import Fraction
from typing import Dict, List, Optional, Tuple

import numpy as np

from .matrix_data import MatrixData
from .mps import MPS, MPSError
from . import presolve as _presolve

ZERO = Fraction(0)
ONE = Fraction(1)


@dataclass
class Variable:
    """general_form `Variable` (cost, bounds, shift, flipped)."""
    cost: Fraction
    lower_bound: Optional[Fraction]
    upper_bound: Optional[Fraction]
    shift: Fraction = ZERO
    flipped: bool = False
    integer: bool = False


class Solved(Exception):
    """Presolve determined every variable: `LinearProgramType::FiniteOptimum(solution)`
    (general_form/mod.rs:354-356).  ``objective`` is the fixed cost, ``values`` maps names to values."""

    def __init__(self, objective, values):
        super().__init__("solved by presolve")
        self.objective, self.values = objective, values


class GeneralForm:
    """general_form/mod.rs:39-73."""

    def __init__(self, maximize: bool, columns: List[List[Tuple[int, Fraction]]], constraint_types: List[tuple],
                 b: List[Fraction], variables: List[Variable], names: List[str], fixed_cost: Fraction = ZERO):
        self.maximize = maximize
        self.columns = [list(c) for c in columns]            # column major, sorted by row
        self.constraint_types = list(constraint_types)       # ('E',) ('L',) ('G',) ('R', r)
        self.b = list(b)
        self.variables = variables
        self.fixed_cost = Fraction(fixed_cost)
        # original variable bookkeeping: ('active', j) or ('free', j_plus, j_minus)
        self.original = [(n, ("active", j)) for j, n in enumerate(names)]
        self.from_active_to_original = list(range(len(names)))
        self.counts = None

    # ---- MPS -> GeneralForm (io/mps/convert.rs) ---------------------------------------------
    @classmethod
    def from_mps(cls, mps: MPS) -> "GeneralForm":
        nvars = len(mps.columns)
        cost = [ZERO] * nvars
        for (j, v) in mps.cost_values:
            cost[j] = v
        variables = [Variable(cost[j], None, None, ZERO, False, mps.columns[j][1]) for j in range(nvars)]
        needs_default_lower = [True] * nvars
        is_free = [False] * nvars

        def tighten(var, which, value, greater):
            cur = getattr(var, which)
            if cur is None or (value > cur if greater else value < cur):
                setattr(var, which, value)
        for (_bname, vals) in mps.bounds:                     # process_bounds, convert.rs:144-187
            for (j, (btype, value)) in vals:
                var = variables[j]
                nd, fr = False, False
                if btype in ("LO", "LI"):
                    tighten(var, "lower_bound", value, True)
                elif btype in ("UP", "UI"):
                    tighten(var, "upper_bound", value, False)
                    nd = True                                  # implied 0 lower bound stays possible (GLPK)
                elif btype == "FX":
                    tighten(var, "lower_bound", value, True)
                    tighten(var, "upper_bound", value, False)
                elif btype == "FR":
                    if var.lower_bound is not None or var.upper_bound is not None:
                        raise MPSError("Variable can't be bounded and free")
                    fr = True
                elif btype == "MI":
                    tighten(var, "upper_bound", ZERO, False)
                elif btype == "PL":
                    tighten(var, "lower_bound", ZERO, True)
                elif btype == "BV":
                    tighten(var, "lower_bound", ZERO, True)
                    tighten(var, "upper_bound", ONE, False)
                    var.integer = True
                if btype in ("LI", "UI"):
                    var.integer = True
                is_free[j] = is_free[j] or fr
                needs_default_lower[j] = needs_default_lower[j] and nd
        for j, var in enumerate(variables):
            if is_free[j] and (var.lower_bound is not None or var.upper_bound is not None):
                raise MPSError("A variable is both free and bounded.")
        for j, var in enumerate(variables):                    # fill_in_default_lower_bounds, convert.rs:289-304
            if needs_default_lower[j]:
                var.lower_bound = ZERO
        # constraints (convert.rs:313-500)
        nrows = len(mps.rows)
        range_rows = sorted(((i, v) for (_g, vals) in mps.ranges for (i, v) in vals), key=lambda t: t[0])
        range_of = dict(range_rows)
        ctypes: List[tuple] = []
        for i, (_name, kind) in enumerate(mps.rows):
            if i in range_of:
                ctypes.append(("E",) if range_of[i] == 0 else ("R", range_of[i]))
            else:
                ctypes.append((kind,))
        bvals: List[Optional[Fraction]] = [None] * nrows
        for (_g, vals) in mps.rhss:
            for (i, value) in vals:
                kind = mps.rows[i][1]
                if bvals[i] is None:
                    if ctypes[i][0] == "R":
                        r = ctypes[i][1]
                        sign = (r > 0) - (r < 0)
                        r = abs(r)
                        ctypes[i] = ("R", r)
                        if kind == "G":
                            bvals[i] = value + r
                        elif kind == "L":
                            bvals[i] = value
                        else:
                            bvals[i] = value + r if sign >= 0 else value
                    else:
                        bvals[i] = value
                else:
                    if kind == "E":
                        if value != bvals[i]:
                            raise MPSError("Trivial infeasibility: a constraint can't equal two values")
                    elif kind == "G":
                        bvals[i] = max(bvals[i], value)
                    else:
                        bvals[i] = min(bvals[i], value)
        b = [ZERO if v is None else v for v in bvals]
        columns = [list(c[2]) for c in mps.columns]
        return cls(mps.maximize, columns, ctypes, b, variables, [c[0] for c in mps.columns], ZERO)

    # ---- standardisation (general_form/mod.rs:488-613) ----------------------------------------
    def transform_variables(self) -> None:
        # split_free_variables, :536-569
        free = [j for j, v in enumerate(self.variables) if v.lower_bound is None and v.upper_bound is None]
        for j in free:
            self.columns.append([(i, -v) for (i, v) in self.columns[j]])
            orig = self.from_active_to_original[j]
            self.original[orig] = (self.original[orig][0], ("free", j, len(self.from_active_to_original)))
            self.from_active_to_original.append(orig)
            self.variables.append(Variable(-self.variables[j].cost, ZERO, None, ZERO, False, self.variables[j].integer))
            self.variables[j].lower_bound = ZERO
        for j, var in enumerate(self.variables):
            if var.lower_bound is None and var.upper_bound is not None:       # flip, :498-509
                var.flipped = not var.flipped
                var.shift = -var.shift
                var.cost = -var.cost
                var.lower_bound = -var.upper_bound
                var.upper_bound = None
                self.columns[j] = [(i, -v) for (i, v) in self.columns[j]]
            if var.lower_bound is not None:                                    # shift, :512-524
                lower = var.lower_bound
                var.shift -= lower
                if var.upper_bound is not None:
                    var.upper_bound -= lower
                self.fixed_cost += lower * var.cost
                for (i, coefficient) in self.columns[j]:
                    self.b[i] -= coefficient * lower
                var.lower_bound = ZERO

    def make_b_non_negative(self) -> None:
        negate = {i for i, v in enumerate(self.b) if v < 0}
        if not negate:
            return
        for j, col in enumerate(self.columns):
            self.columns[j] = [(i, -v if i in negate else v) for (i, v) in col]
        for i in negate:
            kind = self.constraint_types[i][0]
            if kind == "L":
                self.constraint_types[i] = ("G",)
                self.b[i] = -self.b[i]
            elif kind == "E":
                self.b[i] = -self.b[i]
            elif kind == "G":
                self.constraint_types[i] = ("L",)
                self.b[i] = -self.b[i]
            else:
                self.b[i] = self.constraint_types[i][1] - self.b[i]

    def make_minimization_problem(self) -> None:
        if self.maximize:
            self.maximize = False
            for v in self.variables:
                v.cost = -v.cost

    # ---- presolve (general_form/mod.rs:333-478, 727-800) -----------------------------------------
    def presolve(self) -> None:
        """Apply the reductions of `presolve.compute_presolve_changes`.  Raises `presolve.Infeasible` /
        `presolve.Unbounded`, or `Solved` when every variable was determined (the reference returns
        `Err(FiniteOptimum(solution))` and the simplex never runs)."""
        ch = _presolve.compute_presolve_changes(self)
        # update_values_that_remain, :405-437 (indices relative to the problem before any removal)
        for i, v in ch["b"].items():
            self.b[i] = v
        for i, t in ch["constraints"].items():
            self.constraint_types[i] = t
        self.fixed_cost += ch["fixed_cost"]
        for (j, sol) in ch["removed_variables"]:
            orig = self.from_active_to_original[j]
            self.original[orig] = (self.original[orig][0], ("removed", sol))
        for (j, direction), value in ch["bounds"].items():
            if direction == _presolve.LOWER:
                self.variables[j].lower_bound = value
            else:
                self.variables[j].upper_bound = value
        # remove_rows_and_columns, :439-478 (survivors keep their relative order)
        gone_cols = {j for (j, _) in ch["removed_variables"]}
        gone_rows = set(ch["constraints_marked_removed"])
        if gone_cols:
            keep = [j for j in range(len(self.variables)) if j not in gone_cols]
            self.columns = [self.columns[j] for j in keep]
            self.variables = [self.variables[j] for j in keep]
            self.from_active_to_original = [self.from_active_to_original[j] for j in keep]
            for new_index, orig in enumerate(self.from_active_to_original):
                self.original[orig] = (self.original[orig][0], ("active", new_index))
        if gone_rows:
            new_row = {}
            for i in range(len(self.b)):
                if i not in gone_rows:
                    new_row[i] = len(new_row)
            self.columns = [[(new_row[i], v) for (i, v) in col if i not in gone_rows] for col in self.columns]
            self.constraint_types = [t for i, t in enumerate(self.constraint_types) if i not in gone_rows]
            self.b = [v for i, v in enumerate(self.b) if i not in gone_rows]
        # compute_solution_where_possible, :727-750: functions of solved variables become values
        memo: Dict[int, Optional[Fraction]] = {}

        def value_of(k):
            ref = self.original[k][1]
            if ref[0] != "removed":
                return None
            if ref[1][0] == "solved":
                return ref[1][1]
            if k not in memo:
                memo[k] = None
                total = ZERO
                for (other, coefficient) in ref[1][2]:
                    v = value_of(other)
                    if v is None:
                        total = None
                        break
                    total += coefficient * v
                memo[k] = None if total is None else ref[1][1] - total
            return memo[k]
        for k in range(len(self.original)):
            ref = self.original[k][1]
            if ref[0] == "removed" and ref[1][0] == "function":
                v = value_of(k)
                if v is not None:
                    self.original[k] = (self.original[k][0], ("removed", ("solved", v)))
        if all(ref[0] == "removed" and ref[1][0] == "solved" for (_, ref) in self.original):
            raise Solved(self.fixed_cost, {name: ref[1][1] for (name, ref) in self.original})

    def standardize(self, presolve: bool = True) -> None:
        """general_form/mod.rs:307-314"""
        if presolve:
            self.presolve()
        self.transform_variables()
        self.make_b_non_negative()
        self.make_minimization_problem()

    def reorder_constraints_by_type(self) -> Tuple[int, int, int, int]:
        order = {"E": 0, "R": 1, "L": 2, "G": 3}
        idx = sorted(range(len(self.b)), key=lambda i: order[self.constraint_types[i][0]])   # stable
        dest = {src: d for d, src in enumerate(idx)}
        self.constraint_types = [self.constraint_types[i] for i in idx]
        self.b = [self.b[i] for i in idx]
        self.columns = [sorted(((dest[i], v) for (i, v) in col), key=lambda t: t[0]) for col in self.columns]
        counts = [sum(1 for t in self.constraint_types if t[0] == k) for k in ("E", "R", "L", "G")]
        self.counts = tuple(counts)
        return self.counts

    # ---- MatrixData ---------------------------------------------------------------------------
    def derive_matrix_data_exact(self, presolve: bool = True):
        """(columns, b, ranges, counts, costs, upper bounds) as exact data for oracle/relp_exact.MatrixData."""
        self.standardize(presolve=presolve)
        ne, nr, nl, ng = self.reorder_constraints_by_type()
        ranges = [t[1] for t in self.constraint_types[ne:ne + nr]]
        costs = [v.cost for v in self.variables]
        ubs = [v.upper_bound for v in self.variables]
        return self.columns, self.b, ranges, (ne, nr, nl, ng), costs, ubs

    def to_matrix_data(self, exact) -> MatrixData:
        """f64 `MatrixData` for the engine from the tuple returned by ``derive_matrix_data_exact``."""
        columns, b, ranges, (ne, nr, nl, ng), costs, ubs = exact
        col_ptr = [0]
        rows, vals = [], []
        for col in columns:
            for (i, v) in col:
                rows.append(i)
                vals.append(float(v))
            col_ptr.append(len(rows))
        return MatrixData(nr_normal=len(columns), nr_eq=ne, nr_range=nr, nr_le=nl, nr_ge=ng,
                          b=np.array([float(v) for v in b]), cost=np.array([float(c) for c in costs]),
                          upper_bound=np.array([np.inf if u is None else float(u) for u in ubs]),
                          ranges=np.array([float(r) for r in ranges]),
                          col_ptr=np.array(col_ptr, dtype=np.int64), row_idx=np.array(rows, dtype=np.int32),
                          values=np.array(vals, dtype=np.float64))

    # ---- solution reconstruction (general_form/mod.rs:817-942) --------------------------------
    def compute_full_solution(self, reduced: Dict[int, object]):
        """``reduced``: value per active (standardised) variable index, zeros omitted.
        Returns (objective, {name: value})."""
        zero = ZERO if all(isinstance(v, Fraction) for v in reduced.values()) else 0.0
        cost = sum((v * (self.variables[j].cost if isinstance(v, Fraction) else float(self.variables[j].cost))
                    for j, v in reduced.items()), zero)
        cost = cost + (self.fixed_cost if isinstance(cost, Fraction) else float(self.fixed_cost))
        values = {}
        for j, var in enumerate(self.variables):                # reshift_solution, :817-836
            v = reduced.get(j, zero)
            v = v - (var.shift if isinstance(v, Fraction) else float(var.shift))
            if var.flipped:
                v = -v
            values[j] = v
        # compute_solution_value_with_bfs, :889-942
        solved: Dict[int, object] = {}

        def value_of(k):
            if k in solved:
                return solved[k]
            ref = self.original[k][1]
            if ref[0] == "active":
                v = values[ref[1]]
            elif ref[0] == "free":
                v = values[ref[1]] - values[ref[2]]
            elif ref[1][0] == "solved":
                v = ref[1][1] if isinstance(zero, Fraction) else float(ref[1][1])
            else:
                _, constant, coefficients = ref[1]
                exact = isinstance(zero, Fraction)
                total = zero
                for (other, coefficient) in coefficients:
                    total = total + value_of(other) * (coefficient if exact else float(coefficient))
                v = (constant if exact else float(constant)) - total
            solved[k] = v
            return v
        out = {name: value_of(k) for k, (name, _) in enumerate(self.original)}
        return cost, out
