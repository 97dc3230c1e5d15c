"""Small hand-written LPs for the corners of the path: 1 x 1 problems, empty columns, fully degenerate
right-hand sides, zero costs, rank deficiency, contradictory rows, a range row, bounds only."""
import numpy as np

from rust_lp_amd import MatrixData

INF = np.inf


def md(rows_eq, rows_le, rows_ge, b, c, ub, rows_range=(), ranges=()):
    rows = [*rows_eq, *rows_range, *rows_le, *rows_ge]
    n = len(c)
    A = np.asfortranarray(np.array(rows, dtype=np.float64).reshape(len(rows), n))
    return MatrixData(nr_normal=n, nr_eq=len(rows_eq), nr_range=len(rows_range), nr_le=len(rows_le), nr_ge=len(rows_ge),
                      b=np.array(b, dtype=np.float64), cost=np.array(c, dtype=np.float64), upper_bound=np.array(ub, dtype=np.float64),
                      ranges=np.array(ranges, dtype=np.float64), dense=A)


CASES = {
    "1x1 le": md([], [[1.0]], [], [2.0], [-1.0], [INF]),
    "1x1 eq": md([[2.0]], [], [], [4.0], [1.0], [INF]),
    "1x1 ge unbounded": md([], [], [[1.0]], [1.0], [-1.0], [INF]),
    "1x1 ge bounded by ub": md([], [], [[1.0]], [1.0], [-1.0], [5.0]),
    "empty column": md([], [[1.0, 0.0], [2.0, 0.0]], [], [2.0, 3.0], [-1.0, 1.0], [INF, INF]),
    "empty column negative cost (unbounded)": md([], [[1.0, 0.0]], [], [2.0], [-1.0, -1.0], [INF, INF]),
    "all-zero rhs": md([], [[1.0, -1.0], [-1.0, 2.0], [1.0, 1.0]], [], [0.0, 0.0, 0.0], [-1.0, -1.0], [INF, INF]),
    "zero costs": md([[1.0, 1.0]], [[1.0, 0.0]], [], [2.0, 1.0], [0.0, 0.0], [INF, INF]),
    "duplicate equalities (rank deficient)": md([[1.0, 1.0], [2.0, 2.0]], [], [], [2.0, 4.0], [1.0, 2.0], [INF, INF]),
    "contradictory equalities": md([[1.0, 1.0], [1.0, 1.0]], [], [], [2.0, 3.0], [1.0, 2.0], [INF, INF]),
    "range row": md([], [[1.0, 1.0]], [], [10.0, 4.0], [-1.0, -2.0], [INF, 3.0], rows_range=[[1.0, 2.0]], ranges=[2.0]),
    "only bounds matter": md([], [[1.0, 1.0]], [], [100.0], [-1.0, -1.0], [2.0, 3.0]),
    "zero ub": md([], [[1.0, 1.0]], [], [5.0], [-1.0, -2.0], [0.0, INF]),
}
