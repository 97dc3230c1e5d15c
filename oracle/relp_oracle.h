/*
 * relp_oracle.h -- C (f64) CPU restatement of RELP's revised-simplex pivot path.
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the reported CPU baseline.  The product
 * (rust-lp_amd/) never links or loads it.
 *
 * It restates, single-threaded and with the reference's own data structures (sorted sparse
 * (index, value) vectors with exact zeros removed, dense b and -pi, per-column clone in PRICE):
 *   two_phase/phase_one.rs, phase_two.rs, strategy/pivot_rule.rs, tableau/mod.rs,
 *   tableau/kind/{artificial/partially.rs, non_artificial.rs},
 *   inverse_maintenance/carry/{mod.rs, basis_inverse_rows.rs}, matrix_provider/matrix_data.rs.
 * Field = double.  The reference has NO float field (README.md:27; ops.rs:43-46 needs Ord+Eq), so
 * this is the build's f64 extension of the same algorithm; its results are pinned by comparing
 * its pivot traces with the exact oracle (oracle/relp_exact.py) in tests/.
 *
 * Tolerances (all 0 => literally the reference's exact comparisons):
 *   tol_cost  : column j is a candidate iff d_j < -tol_cost          (pivot_rule.rs:56,81,117)
 *   tol_pivot : row i takes part in the ratio test iff alpha_i > tol_pivot   (tableau/mod.rs:227)
 *   tol_zero  : |b_i| <= tol_zero is read as b_i = 0 in the ratio            (f64 only)
 *   tol_tie   : rows with ratio <= min + tol_tie*max(1,|min|) tie; smallest leaving column wins
 *               (tableau/mod.rs:229-239, Bland).  SteepestDescent uses the same band on d_j: columns
 *               with d_j <= min + tol_tie*max(1,|min|) tie and the lowest index wins (pivot_rule.rs:118)
 *   tol_feas  : phase 1 is feasible iff |objective| <= tol_feas*max(1, initial objective)
 */
#ifndef RELP_ORACLE_H
#define RELP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t nr_normal;                       /* structural columns */
    int32_t nr_eq, nr_range, nr_le, nr_ge;   /* constraint rows, stored in this order */
    const int64_t *col_ptr;                  /* CSC over constraint rows: nr_normal + 1 */
    const int32_t *row_idx;                  /* sorted inside each column */
    const double  *values;
    const double  *b;                        /* nr_constraints */
    const double  *ranges;                   /* nr_range */
    const double  *cost;                     /* nr_normal */
    const double  *upper_bound;              /* nr_normal; +inf = no upper bound */
} oracle_matrix_data_t;

typedef struct {
    double tol_cost, tol_pivot, tol_zero, tol_tie, tol_feas;
    int32_t phase_one_rule;                  /* 0 FirstProfitable, 1 FirstProfitableWithMemory, 2 SteepestDescent */
    int32_t phase_two_rule;
    /* Extensions the exact reference has no need for (0 = the reference's rule, the default everywhere):
     * ratio_rule 1: among the rows inside the tie band the largest pivot element wins (compared at float precision), then the
     *   lowest leaving column (tableau/mod.rs:229-239 has only the latter);
     * artificial_removal 1: a basic artificial is pivoted out in the row it is basic IN (phase_one.rs:236 takes the row it
     *   started in, so one that re-entered elsewhere survives into phase 2 as a free column) on ANY non-basic column with a
     *   non-zero element in that row (phase_one.rs:239-244 tries only columns with zero reduced cost), and where there is
     *   none the artificial's OWN constraint is removed together with the basis position it sits in (the two positions are
     *   exchanged first; phase_one.rs:252 pushes the artificial's index, which is another row once `<=` rows lie in front). */
    int32_t ratio_rule;
    int32_t artificial_removal;
    /* basis_inverse 0: `BasisInverseRows` (carry/basis_inverse_rows.rs: explicit inverse as sparse rows, never refactorised);
     * 1: `LUDecomposition` (carry/lower_upper: P B Q = L U with Markowitz pivoting, Forrest-Tomlin-style update file),
     * what src/bin/main.rs:52 runs.  refactor_after: the basis is re-inverted from its columns when MORE than this many updates
     * are pending (lower_upper/mod.rs:199-202 has 10); <= 0 means 10. */
    int32_t basis_inverse;
    int32_t refactor_after;
    /* LU back-end, f64 only: a Markowitz pivot must be at least this fraction of its column's largest active entry (0 = the
     * reference's search over every entry, pivoting.rs:45-81) */
    double lu_threshold;
} oracle_config_t;

enum { ORACLE_RULE_FIRST_PROFITABLE = 0, ORACLE_RULE_FIRST_PROFITABLE_WITH_MEMORY = 1, ORACLE_RULE_STEEPEST_DESCENT = 2 };
enum { ORACLE_RUNNING = 0, ORACLE_OPTIMAL = 1, ORACLE_UNBOUNDED = 2, ORACLE_INFEASIBLE = 3,
       ORACLE_ITERATION_LIMIT = 4, ORACLE_PHASE_ONE_DONE = 5, ORACLE_ERROR = -1 };

typedef struct oracle_engine oracle_engine_t;

/* Copies the problem.  Builds the partially-artificial phase-1 tableau (partially.rs:125-206). */
oracle_engine_t *oracle_create(const oracle_matrix_data_t *md, const oracle_config_t *cfg);
void oracle_destroy(oracle_engine_t *e);

/* Run up to max_iters basis changes of the current phase (phase-1 -> phase-2 switch included when
 * `through_phases` != 0).  Appends to the trace arrays (each of capacity trace_cap, may be NULL).
 * Returns the status enum. */
int oracle_run(oracle_engine_t *e, int64_t max_iters, int through_phases,
               int32_t *tr_phase, int32_t *tr_entering, int32_t *tr_row, int32_t *tr_leaving,
               int64_t trace_cap, int64_t *n_done);

int32_t oracle_m(const oracle_engine_t *e);           /* rows of the current tableau */
int32_t oracle_n(const oracle_engine_t *e);           /* columns of the current tableau (incl. artificials) */
int32_t oracle_phase(const oracle_engine_t *e);       /* 1 or 2 */
int32_t oracle_nr_artificial(const oracle_engine_t *e);
/* rows removed as redundant at the phase switch (indices as the reference pushes them, phase_one.rs:252) */
int32_t oracle_nr_filtered_rows(const oracle_engine_t *e);
/* pivots made at zero level to drive basic artificial variables out at the end of phase 1 (phase_one.rs:236-250) */
int32_t oracle_nr_zero_level_pivots(const oracle_engine_t *e);
/* artificial_removal 1: stuck artificial variables that sat in a foreign basis position and were moved into their own row */
int32_t oracle_nr_position_exchanges(const oracle_engine_t *e);
void oracle_get_filtered_rows(const oracle_engine_t *e, int32_t *out);
double  oracle_objective(const oracle_engine_t *e);
void    oracle_get_b(const oracle_engine_t *e, double *out);
void    oracle_get_minus_pi(const oracle_engine_t *e, double *out);
void    oracle_get_basis(const oracle_engine_t *e, int32_t *out);
/* dense row-major copy of B^-1 (m*m) */
void    oracle_get_basis_inverse(const oracle_engine_t *e, double *out);
int64_t oracle_basis_inverse_nnz(const oracle_engine_t *e);
/* LU back-end: { refactorisations so far, updates pending, entries of L, entries of U (with the diagonal) } */
void    oracle_lu_stats(const oracle_engine_t *e, int64_t *out4);

/* ---- the LU back-end by itself, for the reference's own known answers (lower_upper/mod.rs:488-867) ---- */
typedef struct oracle_lu oracle_lu_t;
oracle_lu_t *oracle_lu_from_triangles(int32_t m, const int64_t *lptr, const int32_t *lidx, const double *lval,
                                      const int64_t *uptr, const int32_t *uidx, const double *uval);
oracle_lu_t *oracle_lu_invert(int32_t m, const int64_t *cptr, const int32_t *cidx, const double *cval);
void oracle_lu_destroy(oracle_lu_t *h);
void oracle_lu_generate_column(oracle_lu_t *h, int32_t n, const int32_t *idx, const double *val, double *out_m);
int  oracle_lu_change_basis(oracle_lu_t *h, int32_t pivot_row_index);      /* with the spike of the last generate_column */
void oracle_lu_basis_inverse_row(oracle_lu_t *h, int32_t row, double *out_m);
int32_t oracle_lu_nr_updates(const oracle_lu_t *h);
void oracle_lu_get_update(const oracle_lu_t *h, int32_t k, int32_t *pivot, double *values_m);
void oracle_lu_get_factor(const oracle_lu_t *h, int which, double *out_mm, int32_t *row_forward, int32_t *column_forward);

#ifdef __cplusplus
}
#endif
#endif
