"""Presolve of a `GeneralForm`: the reduction rules the reference applies
before standardisation, restated in exact ``Fraction`` arithmetic.  One-shot, serial, branchy CPU code;
it decides which rows / columns (and which bounds, right-hand sides, constraint types) the pivot engine
is given, so the `MatrixData` - and with it the pivot sequence - equals the reference's.

Restates (file:line under /root/reference/src/data/linear_program/general_form/):
  mod.rs:333-391            presolve driver, termination heuristic
  mod.rs:405-478            update_values_that_remain, remove_rows_and_columns
  mod.rs:727-800            compute_solution_where_possible, get_solution
  presolve/mod.rs:26-418    Index: presolve_step (rule priority), after_bound_change,
                            update_activity_bounds / _counters, remove_constraint_values,
                            queue_variable_by_counter, queue_constraint_by_counter, is_empty_constraint_feasible
  presolve/counters.rs      row / column counts, missing-bound counts per activity side, active iteration
  presolve/queues.rs        four work lists (three stacks, one FIFO with a membership guard), initial contents
  presolve/updates.rs       overlay of b, constraint types, bounds, activity-derived bounds; into_changes
  presolve/rule/fixed_variable.rs, bound_constraint.rs, slack.rs, domain_propagation.rs

Determinism: the reference's only hash containers are maps keyed by index (iteration order never
matters: every key is written independently) and the FIFO's membership guard.
"""
from __future__ import annotations

from collections import deque
from fractions import Fraction
from typing import Dict, List, Optional, Tuple

ZERO = Fraction(0)
LOWER, UPPER = 0, 1                       # BoundDirection
MEANINGFUL, NOT_MEANINGFUL, NONE = 2, 1, 0   # Change (presolve/mod.rs:47-51)


class Infeasible(Exception):
    """LinearProgramType::Infeasible"""


class Unbounded(Exception):
    """LinearProgramType::Unbounded"""


def _flip(direction: int) -> int:
    return 1 - direction


def _times_sign(direction: int, coefficient: Fraction) -> int:
    """`BoundDirection * NonZeroSign` (elements.rs): a negative coefficient swaps the sides."""
    return direction if coefficient > 0 else 1 - direction


def is_empty_constraint_feasible(rhs: Fraction, ctype: tuple) -> bool:
    """presolve/mod.rs:396-418"""
    kind = ctype[0]
    if kind == "E":
        return rhs == 0
    if kind == "R":
        return rhs >= 0 and rhs - ctype[1] <= 0
    if kind == "L":
        return rhs >= 0
    return rhs <= 0


def optimize_independent_column(maximize: bool, cost: Fraction, lower, upper) -> Fraction:
    """presolve/updates.rs:540-560"""
    assert cost != 0
    wants_lower = (cost > 0) != maximize              # (Minimize, +) or (Maximize, -)
    bound = lower if wants_lower else upper
    if bound is None:
        raise Unbounded()
    return bound


class Index:
    """presolve/mod.rs `Index` with its `Counters`, `Queues` and `Updates`."""

    def __init__(self, gf):
        self.gf = gf
        m, n = len(gf.b), len(gf.variables)
        # ---- Counters::new (counters.rs:48-80) ----
        self.rows: List[List[Tuple[int, Fraction]]] = [[] for _ in range(m)]
        for j, col in enumerate(gf.columns):
            for (i, v) in col:
                self.rows[i].append((j, v))
        self.c_constraint = [len(r) for r in self.rows]
        self.c_variable = [len(c) for c in gf.columns]
        self.c_activity = []
        for row in self.rows:
            lo = up = 0
            for (j, coefficient) in row:
                var = gf.variables[j]
                lower, upper = (var.lower_bound, var.upper_bound) if coefficient > 0 else (var.upper_bound, var.lower_bound)
                lo += lower is None
                up += upper is None
            self.c_activity.append([lo, up])
        # ---- Updates::new (updates.rs:105-159) ----
        self.u_b: Dict[int, Fraction] = {}
        self.u_constraints: Dict[int, tuple] = {}
        self.u_fixed_cost = ZERO
        self.u_bounds: Dict[Tuple[int, int], Fraction] = {}
        self.u_activity_bounds: Dict[Tuple[int, int], Fraction] = {}
        self.removed_variables: List[Tuple[int, tuple]] = []     # (j, ('solved', v) | ('function', constant, [(orig, coef)]))
        self.constraints_marked_removed: List[int] = []
        for j in range(n):
            if self.c_variable[j] != 0:
                continue
            var = gf.variables[j]
            if var.cost == 0:
                value = var.upper_bound if var.upper_bound is not None else (var.lower_bound if var.lower_bound is not None else ZERO)
            else:
                value = optimize_independent_column(gf.maximize, var.cost, var.lower_bound, var.upper_bound)
                self.u_fixed_cost += var.cost * value
            self.removed_variables.append((j, ("solved", value)))
        for i in range(m):
            if self.c_constraint[i] == 0:
                if not is_empty_constraint_feasible(gf.b[i], gf.constraint_types[i]):
                    raise Infeasible()
                self.constraints_marked_removed.append(i)
        # ---- Queues::new (queues.rs:137-171) ----
        self.q_bound = [i for i in range(m) if self.c_constraint[i] == 1]
        self.q_activity = deque()
        self.q_activity_set = set()
        for i in range(m):
            if self.c_constraint[i] > 1:
                if self.c_activity[i][0] <= 1:
                    self.activity_insert(i, LOWER)
                if self.c_activity[i][1] <= 1:
                    self.activity_insert(i, UPPER)
        self.q_slack = [j for j in range(n) if self.c_variable[j] == 1 and gf.variables[j].cost == 0]
        self.q_substitution = [j for j in range(n) if self.c_variable[j] > 0 and
                               gf.variables[j].lower_bound is not None and
                               gf.variables[j].lower_bound == gf.variables[j].upper_bound]
        self.activity_bounds: List[List[Optional[Fraction]]] = [[None, None] for _ in range(m)]

    # ---- queues ----
    def activity_insert(self, constraint: int, direction: int) -> None:
        key = (constraint, direction)
        if key not in self.q_activity_set:
            self.q_activity_set.add(key)
            self.q_activity.append(key)

    def activity_pop(self):
        if not self.q_activity:
            return None
        key = self.q_activity.popleft()
        self.q_activity_set.discard(key)
        return key

    def are_queues_empty(self) -> bool:
        return not (self.q_activity or self.q_slack or self.q_bound or self.q_substitution)

    # ---- counters ----
    def is_constraint_still_active(self, i: int) -> bool:
        return self.c_constraint[i] > 0

    def is_variable_still_active(self, j: int) -> bool:
        return self.c_variable[j] > 0

    def iter_active_column(self, j: int):
        return [(i, v) for (i, v) in self.gf.columns[j] if self.c_constraint[i] > 0]

    def iter_active_row(self, i: int):
        return [(j, v) for (j, v) in self.rows[i] if self.c_variable[j] > 0]

    # ---- updates (latest version of the problem) ----
    def b(self, i: int) -> Fraction:
        return self.u_b.get(i, self.gf.b[i])

    def change_b(self, i: int, change: Fraction) -> None:
        self.u_b[i] = self.b(i) + change

    def constraint_type(self, i: int) -> tuple:
        return self.u_constraints.get(i, self.gf.constraint_types[i])

    def variable_bound(self, j: int, direction: int) -> Optional[Fraction]:
        """updates.rs:221-243: activity-derived, then certain, then original"""
        key = (j, direction)
        if key in self.u_activity_bounds:
            return self.u_activity_bounds[key]
        if key in self.u_bounds:
            return self.u_bounds[key]
        var = self.gf.variables[j]
        return var.lower_bound if direction == LOWER else var.upper_bound

    def is_variable_fixed(self, j: int) -> Optional[Fraction]:
        lower, upper = self.variable_bound(j, LOWER), self.variable_bound(j, UPPER)
        return lower if (lower is not None and upper is not None and lower == upper) else None

    def variable_feasible_value(self, j: int) -> Optional[Fraction]:
        """updates.rs:197-218 (prefers the upper bound)"""
        lower, upper = self.variable_bound(j, LOWER), self.variable_bound(j, UPPER)
        if lower is None and upper is None:
            return ZERO
        if lower is None:
            return upper
        if upper is None:
            return lower
        return upper if lower <= upper else None

    @staticmethod
    def _compare_and_update(key, new, existing, bounds):
        """bound_compare_and_update, updates.rs:498-520: ('none',) | ('shift', difference)"""
        tighter = new > existing if key[1] == LOWER else new < existing
        if tighter:
            bounds[key] = new
            return ("shift", new - existing)
        return ("none",)

    def update_bound(self, j: int, direction: int, new: Fraction):
        """updates.rs:260-299: a bound that is exported for certain"""
        key = (j, direction)
        if key in self.u_bounds:
            compare_with = self.u_bounds[key]
        elif key in self.u_activity_bounds:
            compare_with = self.u_activity_bounds.pop(key)
            self.u_bounds[key] = compare_with
        else:
            var = self.gf.variables[j]
            original = var.lower_bound if direction == LOWER else var.upper_bound
            if original is None:
                self.u_bounds[key] = new
                return ("new",)
            compare_with = original
        return self._compare_and_update(key, new, compare_with, self.u_bounds)

    def update_activity_variable_bound(self, j: int, direction: int, new: Fraction):
        """updates.rs:316-360: a bound derived from a row's activity (exported only if it turns out useful)"""
        key = (j, direction)
        if key in self.u_activity_bounds:
            return self._compare_and_update(key, new, self.u_activity_bounds[key], self.u_activity_bounds)
        if key in self.u_bounds:
            return self._compare_and_update(key, new, self.u_bounds[key], self.u_bounds)
        var = self.gf.variables[j]
        original = var.lower_bound if direction == LOWER else var.upper_bound
        if original is None:
            self.u_activity_bounds[key] = new
            return ("new",)
        return self._compare_and_update(key, new, original, self.u_activity_bounds)

    def optimize_column_independently(self, j: int):
        var = self.gf.variables[j]
        value = optimize_independent_column(self.gf.maximize, var.cost, self.variable_bound(j, LOWER),
                                            self.variable_bound(j, UPPER))
        self.u_fixed_cost += var.cost * value
        return ("solved", value)

    def nr_variables_remaining(self) -> int:
        return len(self.gf.variables) - len(self.removed_variables)

    def nr_constraints_remaining(self) -> int:
        return len(self.gf.b) - len(self.constraints_marked_removed)

    # ---- presolve/mod.rs ----
    def presolve_step(self) -> int:
        """mod.rs:122-157: the first applicable rule, in priority order"""
        if self.q_substitution:
            self.presolve_fixed_variable(self.q_substitution.pop())
            return MEANINGFUL
        while self.q_bound:
            constraint = self.q_bound.pop()
            if self.is_constraint_still_active(constraint):
                self.presolve_bound_constraint(constraint)
                return MEANINGFUL
        while self.q_slack:
            variable = self.q_slack.pop()
            if self.is_variable_still_active(variable):
                self.presolve_slack(variable)
                return MEANINGFUL
        while True:
            item = self.activity_pop()
            if item is None:
                break
            constraint, direction = item
            if self.is_constraint_still_active(constraint):
                return self.presolve_domain_propagation(constraint, direction)
        return NOT_MEANINGFUL

    def after_bound_change(self, variable: int, direction: int, change: Optional[Fraction]) -> None:
        """mod.rs:172-193"""
        if self.is_variable_fixed(variable) is not None and self.is_variable_still_active(variable):
            self.q_substitution.append(variable)
        if change is not None:
            self.update_activity_bounds(variable, direction, change)
        else:
            self.update_activity_counters(variable, direction)

    def update_activity_bounds(self, variable: int, direction: int, by_how_much: Fraction) -> None:
        """mod.rs:208-240"""
        for (row, coefficient) in self.iter_active_column(variable):
            if not self.is_constraint_still_active(row):
                continue
            side = _times_sign(direction, coefficient)
            if self.activity_bounds[row][side] is not None:
                self.activity_bounds[row][side] += by_how_much * coefficient
                self.activity_insert(row, side)

    def update_activity_counters(self, variable: int, direction: int) -> None:
        """mod.rs:248-268"""
        for (constraint, coefficient) in self.iter_active_column(variable):
            side = _times_sign(direction, coefficient)
            self.c_activity[constraint][side] -= 1
            if self.c_activity[constraint][side] <= 1:
                self.activity_insert(constraint, side)

    def remove_constraint_values(self, constraint: int) -> None:
        """mod.rs:279-295"""
        for (variable, _) in self.iter_active_row(constraint):
            self.c_constraint[constraint] -= 1
            self.c_variable[variable] -= 1
            self.queue_variable_by_counter(variable)

    def queue_variable_by_counter(self, variable: int) -> None:
        """mod.rs:300-324"""
        count = self.c_variable[variable]
        if count == 0:
            if self.gf.variables[variable].cost == 0:
                value = ("solved", self.variable_feasible_value(variable))
            else:
                value = self.optimize_column_independently(variable)
            self.remove_variable(variable, value)
        elif count == 1:
            if self.gf.variables[variable].cost == 0:
                self.q_slack.append(variable)

    def queue_constraint_by_counter(self, constraint: int) -> int:
        """mod.rs:345-364"""
        count = self.c_constraint[constraint]
        if count == 0:
            if is_empty_constraint_feasible(self.b(constraint), self.constraint_type(constraint)):
                self.remove_constraint(constraint)
                return MEANINGFUL
            raise Infeasible()
        if count == 1:
            self.q_bound.append(constraint)
        return NONE

    def remove_constraint(self, constraint: int) -> None:
        self.constraints_marked_removed.append(constraint)

    def remove_variable(self, variable: int, solution: tuple) -> None:
        self.removed_variables.append((variable, solution))

    # ---- rule/fixed_variable.rs:21-48 ----
    def presolve_fixed_variable(self, variable: int) -> None:
        value = self.is_variable_fixed(variable)
        column = self.iter_active_column(variable)
        for (constraint, coefficient) in column:
            self.change_b(constraint, -coefficient * value)
        self.u_fixed_cost += self.gf.variables[variable].cost * value
        for (constraint, _) in column:
            self.c_variable[variable] -= 1
            self.c_constraint[constraint] -= 1
            self.queue_constraint_by_counter(constraint)
        self.remove_variable(variable, ("solved", value))

    # ---- rule/bound_constraint.rs:27-85 ----
    def presolve_bound_constraint(self, constraint: int) -> None:
        (variable, coefficient), = self.iter_active_row(constraint)
        bound_value = self.b(constraint) / coefficient
        ctype = self.constraint_type(constraint)
        kind, positive = ctype[0], coefficient > 0
        changes = []
        if (kind == "G" and positive) or (kind == "L" and not positive):
            changes.append((LOWER, bound_value))
        elif (kind == "L" and positive) or (kind == "G" and not positive):
            changes.append((UPPER, bound_value))
        elif kind == "E":
            changes.append((LOWER, bound_value))
            changes.append((UPPER, bound_value))
        else:
            bound1 = (self.b(constraint) - ctype[1]) / coefficient
            if positive:
                changes += [(LOWER, bound1), (UPPER, bound_value)]
            else:
                changes += [(LOWER, bound_value), (UPPER, bound1)]
        self.c_variable[variable] -= 1
        self.c_constraint[constraint] -= 1
        self.remove_constraint(constraint)
        for (direction, value) in changes:
            change = self.update_bound(variable, direction, value)
            if change[0] == "new":
                self.after_bound_change(variable, direction, None)
            elif change[0] == "shift":
                self.after_bound_change(variable, direction, change[1])
        if self.variable_feasible_value(variable) is None:
            raise Infeasible()
        self.queue_variable_by_counter(variable)

    # ---- rule/slack.rs:63-214 ----
    def compute_removed_variable_solution(self, constraint: int, variable: int, coefficient: Fraction) -> tuple:
        constant = self.b(constraint) / coefficient
        coefficients = [(self.gf.from_active_to_original[j], other / coefficient)
                        for (j, other) in self.iter_active_row(constraint) if j != variable]
        return ("function", constant, coefficients)

    def update_activity_queues_if_needed(self, constraint: int, lower_none: bool, upper_none: bool, positive: bool) -> None:
        if (lower_none and positive) or (upper_none and not positive):
            self.c_activity[constraint][0] -= 1
            if self.c_activity[constraint][0] <= 1:
                self.activity_insert(constraint, LOWER)
        if (upper_none and positive) or (lower_none and not positive):
            self.c_activity[constraint][1] -= 1
            if self.c_activity[constraint][1] <= 1:
                self.activity_insert(constraint, UPPER)

    def presolve_slack(self, variable: int) -> None:
        (constraint, coefficient), = self.iter_active_column(variable)
        ctype = self.constraint_type(constraint)
        kind = ctype[0]
        lower, upper = self.variable_bound(variable, LOWER), self.variable_bound(variable, UPPER)
        has_l, has_u = lower is not None, upper is not None
        positive = coefficient > 0

        remove_both = ((not has_l and not has_u) or
                       (kind == "G" and has_l and not has_u and positive) or
                       (kind == "L" and not has_l and has_u and positive) or
                       (kind == "L" and has_l and not has_u and not positive) or
                       (kind == "G" and not has_l and has_u and not positive))
        if remove_both:
            solution = self.compute_removed_variable_solution(constraint, variable, coefficient)
            for (other, _) in self.iter_active_row(constraint):
                self.c_constraint[constraint] -= 1
                self.c_variable[other] -= 1
                if other != variable:
                    self.queue_variable_by_counter(other)
            self.remove_variable(variable, solution)
            self.remove_constraint(constraint)
            return

        # the cases that remove the column only (table of slack.rs:38-58)
        if kind == "E" and has_l and has_u:
            if positive:
                new_type, bound = ("R", coefficient * (upper - lower)), lower
            else:
                new_type, bound = ("R", coefficient * (lower - upper)), upper
        elif kind == "R" and has_l and has_u:
            if positive:
                new_type, bound = ("R", ctype[1] + coefficient * (upper - lower)), lower
            else:
                new_type, bound = ("R", ctype[1] + coefficient * (lower - upper)), upper
        elif positive and ((kind in "LER" and has_l and not has_u) or (kind == "L" and has_l and has_u)):
            new_type, bound = ("L",), lower            # <a, x> <= b - c l
        elif positive and ((kind in "EGR" and not has_l and has_u) or (kind == "G" and has_l and has_u)):
            new_type, bound = ("G",), upper            # <a, x> >= b - c u
        elif not positive and ((kind in "EGR" and has_l and not has_u) or (kind == "G" and has_l and has_u)):
            new_type, bound = ("G",), lower            # <a, x> >= b - c l
        elif not positive and ((kind in "LER" and not has_l and has_u) or (kind == "L" and has_l and has_u)):
            new_type, bound = ("L",), upper            # <a, x> <= b - c u
        else:
            raise AssertionError("slack case table is exhaustive")

        change = -coefficient * bound
        if kind in "ER":
            removed = self.compute_removed_variable_solution(constraint, variable, coefficient)
        else:
            removed = ("solved", bound)
        self.c_variable[variable] -= 1
        self.remove_variable(variable, removed)
        self.update_activity_queues_if_needed(constraint, not has_l, not has_u, positive)
        self.c_constraint[constraint] -= 1
        self.queue_constraint_by_counter(constraint)
        self.change_b(constraint, change)
        self.u_constraints[constraint] = new_type

    # ---- rule/domain_propagation.rs ----
    def presolve_domain_propagation(self, constraint: int, direction: int) -> int:
        counter = self.c_activity[constraint][direction]
        if counter == 0:
            return self.for_entire_constraint(constraint, direction)
        if counter == 1:
            return self.create_variable_bound(constraint, direction)
        raise AssertionError("constraint queued for activity with more than one bound missing")

    def compute_activity_bound_if_needed(self, constraint: int, direction: int) -> Fraction:
        if self.activity_bounds[constraint][direction] is None:
            total = ZERO
            for (variable, coefficient) in self.iter_active_row(constraint):
                total += coefficient * self.variable_bound(variable, _times_sign(direction, coefficient))
            self.activity_bounds[constraint][direction] = total
        return self.activity_bounds[constraint][direction]

    def constraint_update(self, constraint: int, bound_value: Fraction, direction: int):
        """domain_propagation.rs:243-319: None | ('remove',) | ('replace', kind, rhs shift) | ('set',)"""
        rhs = self.b(constraint)
        ctype = self.constraint_type(constraint)
        kind = ctype[0]
        cmp = (rhs > bound_value) - (rhs < bound_value)          # rhs.cmp(bound_value)
        if direction == LOWER:
            if kind in "ERL" and cmp < 0:
                raise Infeasible()
            if kind in "EL" and cmp == 0:
                return ("set",)
            if kind == "G" and cmp <= 0:
                return ("remove",)
            if kind == "R" and cmp > 0:
                lower_bound = rhs - ctype[1]
                return None if bound_value < lower_bound else ("replace", "L", ZERO)
            if kind == "R" and cmp == 0:
                raise AssertionError("a range of width zero should have been an equality")
            return None                                          # (== <= >=) b > l
        # direction == UPPER
        if kind in "EG" and cmp > 0:
            raise Infeasible()
        if kind in "EG" and cmp == 0:
            return ("set",)
        if kind == "L" and cmp >= 0:
            return ("remove",)
        if kind == "R" and cmp == 0:
            return ("replace", "G", -ctype[1])
        if kind == "R" and cmp > 0:
            lower_bound = rhs - ctype[1]
            if bound_value < lower_bound:
                raise Infeasible()
            if bound_value == lower_bound:
                return ("set",)
            return ("replace", "G", -ctype[1])
        return None                                              # b < u

    def can_variable_rule_be_applied(self, constraint: int, direction: int) -> Optional[Fraction]:
        rhs = self.b(constraint)
        ctype = self.constraint_type(constraint)
        kind = ctype[0]
        if kind == "E":
            return rhs
        if kind == "R":
            return rhs if direction == LOWER else rhs - ctype[1]
        if kind == "L":
            return rhs if direction == LOWER else None
        return None if direction == LOWER else rhs

    def for_entire_constraint(self, constraint: int, direction: int) -> int:
        most = [NONE]
        activity_bound = self.compute_activity_bound_if_needed(constraint, direction)
        remove, apply_variable_part = self.constraint_part(constraint, activity_bound, direction, most)
        if apply_variable_part:
            rhs = self.can_variable_rule_be_applied(constraint, direction)
            if rhs is not None:
                self.variable_part(constraint, rhs, activity_bound, direction, most)
        if remove:
            self.remove_constraint_values(constraint)
            self.remove_constraint(constraint)
        return most[0]

    def constraint_part(self, constraint: int, bound: Fraction, direction: int, most) -> Tuple[bool, bool]:
        update = self.constraint_update(constraint, bound, direction)
        if update is None:
            return (False, True)
        if update[0] == "remove":
            result = (True, True)
        elif update[0] == "replace":
            self.u_constraints[constraint] = (update[1],)
            self.change_b(constraint, update[2])
            result = (False, True)
        else:
            to_update = []
            for (variable, coefficient) in self.iter_active_row(constraint):
                vdir = _times_sign(direction, coefficient)
                value = self.variable_bound(variable, vdir)
                if (variable, vdir) in self.u_activity_bounds:
                    self.u_bounds[(variable, vdir)] = self.u_activity_bounds.pop((variable, vdir))
                change = self.update_bound(variable, _flip(vdir), value)
                if change[0] == "new":
                    to_update.append((variable, _flip(vdir)))
                self.q_substitution.append(variable)
            for (variable, vdir) in to_update:
                self.update_activity_counters(variable, vdir)
            result = (True, False)
        most[0] = MEANINGFUL
        return result

    def variable_part(self, constraint: int, rhs: Fraction, activity_bound: Fraction, activity_direction: int, most) -> None:
        for (variable, coefficient) in self.iter_active_row(constraint):
            new_direction = _times_sign(_flip(activity_direction), coefficient)
            bound_value = self.variable_bound(variable, _times_sign(activity_direction, coefficient))
            residual = activity_bound - coefficient * bound_value
            new_value = (rhs - residual) / coefficient
            change = self.update_activity_variable_bound(variable, new_direction, new_value)
            if change[0] == "new":
                self.after_bound_change(variable, new_direction, None)
                most[0] = MEANINGFUL
            elif change[0] == "shift":
                self.after_bound_change(variable, new_direction, change[1])
                if most[0] != MEANINGFUL:
                    most[0] = NOT_MEANINGFUL

    def create_variable_bound(self, constraint: int, activity_direction: int) -> int:
        rhs = self.can_variable_rule_be_applied(constraint, activity_direction)
        if rhs is None:
            return NONE
        total = ZERO
        target = None
        for (variable, coefficient) in self.iter_active_row(constraint):
            bound = self.variable_bound(variable, _times_sign(activity_direction, coefficient))
            if bound is None:
                if target is None:
                    target = (variable, coefficient)
            else:
                total += coefficient * bound
        target_column, target_coefficient = target
        value = (rhs - total) / target_coefficient
        bound_direction = _times_sign(_flip(activity_direction), target_coefficient)
        change = self.update_activity_variable_bound(target_column, bound_direction, value)
        if change[0] == "new":
            self.after_bound_change(target_column, bound_direction, None)
            return MEANINGFUL
        if change[0] == "shift":
            self.after_bound_change(target_column, bound_direction, change[1])
            return NOT_MEANINGFUL
        return NONE

    # ---- Updates::into_changes (updates.rs:397-452) ----
    def into_changes(self):
        removed_rows = set(self.constraints_marked_removed)
        for i in removed_rows:
            self.u_b.pop(i, None)
            self.u_constraints.pop(i, None)
        for (j, _) in self.removed_variables:
            for d in (LOWER, UPPER):
                self.u_bounds.pop((j, d), None)
                self.u_activity_bounds.pop((j, d), None)
        # derived bounds are kept only where they remove the need to split a free variable
        free_to_be_restricted = set()
        for (j, _d) in self.u_activity_bounds:
            var = self.gf.variables[j]
            if var.lower_bound is None and var.upper_bound is None and \
                    (j, LOWER) not in self.u_bounds and (j, UPPER) not in self.u_bounds:
                free_to_be_restricted.add(j)
        for (j, d), value in self.u_activity_bounds.items():
            if j in free_to_be_restricted:
                self.u_bounds[(j, d)] = value
        b = {i: v for i, v in self.u_b.items() if v != self.gf.b[i]}
        constraints = {i: t for i, t in self.u_constraints.items() if t != self.gf.constraint_types[i]}
        return {"b": b, "constraints": constraints, "fixed_cost": self.u_fixed_cost, "bounds": dict(self.u_bounds),
                "removed_variables": sorted(self.removed_variables, key=lambda t: t[0]),
                "constraints_marked_removed": sorted(self.constraints_marked_removed)}


def compute_presolve_changes(gf) -> dict:
    """general_form/mod.rs:368-391"""
    index = Index(gf)
    without_meaningful_change = 0
    while not index.are_queues_empty() and \
            without_meaningful_change < index.nr_variables_remaining() + index.nr_constraints_remaining():
        change = index.presolve_step()
        if change == MEANINGFUL:
            without_meaningful_change = 0
        elif change == NOT_MEANINGFUL:
            without_meaningful_change += 1
    return index.into_changes()
