/*
 * relp_engine.h -- C ABI of the MI355X-native revised-simplex pivot engine.
 *
 * This is the drop-in boundary for RELP's hot path (vandenheuvel/rust-lp, `relp` 0.0.5).  The
 * reference has no FFI; the unit replaced is `Tableau<Carry<F, BI>, K>` + `PivotRule` driven by
 * `phase_one::primal` / `phase_two::primal`.  Each entry point names the reference interface it
 * replaces (file:line under /root/reference/src/algorithm/two_phase/).  A Rust maintainer binds
 * these with `extern "C"` and implements `InverseMaintener` / `PivotRule` on top.
 *
 * Conventions
 *   - plain C types only; every function returns a relp_status_t (0 = ok, < 0 = error); no panics
 *     or aborts cross the boundary (the reference panics on a zero pivot, carry/mod.rs:291).
 *   - a handle is not thread-safe: one host thread per handle; each handle owns one HIP stream
 *     (relp_set_stream lets the caller supply its own, e.g. the stream RCCL collectives run on).
 *   - the problem is copied at create time (the reference borrows the provider, `&'a MP`); all
 *     getters copy into caller-provided host buffers.
 *   - field = f64 (the reference has no float field, README.md:27: this is the build's extension);
 *     comparisons use the tolerances of relp_config_t, all-zero tolerances = the reference's exact
 *     `<`, `>`, `==`.
 *   - column indices are tableau indices: in phase 1 the `nr_artificial` artificial columns come
 *     first (kind/artificial/partially.rs:72-80), in phase 2 they are provider indices.
 */
#ifndef RELP_ENGINE_H
#define RELP_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct relp_engine relp_engine_t;

typedef enum {
    RELP_OK = 0,
    RELP_E_ARG = -1,          /* bad argument */
    RELP_E_HIP = -2,          /* HIP runtime error (relp_last_error gives the text) */
    RELP_E_ZERO_PIVOT = -3,   /* "Pivot value can't be zero." carry/mod.rs:291, basis_inverse_rows.rs:47 */
    RELP_E_SINGULAR = -4,     /* basis could not be inverted (from_basis) */
    RELP_E_STATE = -5,        /* call not valid in the current phase / state */
    RELP_E_UNSUPPORTED = -6,
    RELP_E_ALLOC = -7
} relp_status_t;

/* Outcome of a loop (algorithm/mod.rs:46-50 `OptimizationResult`, phase_one.rs:177-194). */
typedef enum {
    RELP_RUNNING = 0,          /* iteration limit hit, more pivots possible */
    RELP_OPTIMAL = 1,          /* phase 2: no candidate column (FiniteOptimum) */
    RELP_UNBOUNDED = 2,        /* phase 2: no pivot row (Unbounded) */
    RELP_INFEASIBLE = 3,       /* phase 1 ended with objective != 0 */
    RELP_PHASE_ONE_DONE = 4,   /* phase 1 ended feasible; the tableau is now NonArtificial */
    RELP_NO_ROW_PHASE_ONE = 5  /* "Artificial cost can not be unbounded." phase_one.rs:143 */
} relp_outcome_t;

/* strategy/pivot_rule.rs:38,62,97 */
typedef enum {
    RELP_RULE_FIRST_PROFITABLE = 0,
    RELP_RULE_FIRST_PROFITABLE_WITH_MEMORY = 1,   /* phase-1 default, phase_one.rs:55,97 */
    RELP_RULE_STEEPEST_DESCENT = 2                /* = most negative d_j; phase-2 default, two_phase/mod.rs:44 */
} relp_pivot_rule_t;

typedef enum { RELP_FORMAT_CSC = 0, RELP_FORMAT_DENSE = 1 } relp_format_t;
typedef enum { RELP_MEM_HOST = 0, RELP_MEM_DEVICE = 1 } relp_memory_t;

/* matrix_provider/matrix_data.rs:54-90.  Constraint rows are ordered [== | range | <= | >=]; the
 * slack, bound-slack columns and bound rows are virtual (matrix_data.rs:37-52, 308-348). */
typedef struct {
    int32_t nr_normal;                        /* structural columns */
    int32_t nr_eq, nr_range, nr_le, nr_ge;    /* constraint rows */
    int32_t format;                           /* relp_format_t */
    int32_t matrix_memory;                    /* relp_memory_t: where col_ptr/row_idx/values/dense live */
    const int64_t *col_ptr;                   /* CSC: nr_normal + 1 */
    const int32_t *row_idx;                   /* CSC: sorted inside a column */
    const double  *values;                    /* CSC */
    const double  *dense;                     /* DENSE: column-major nr_constraints x nr_normal */
    int64_t        dense_ld;                  /* DENSE: leading dimension (>= nr_constraints) */
    const double  *b;                         /* host, nr_constraints (all >= 0, general_form/mod.rs:574) */
    const double  *ranges;                    /* host, nr_range */
    const double  *cost;                      /* host, nr_normal */
    const double  *upper_bound;               /* host, nr_normal; +inf = none (lower bounds are 0) */
} relp_matrix_data_t;

typedef struct {
    int32_t device;               /* HIP device ordinal, -1 = current */
    int32_t phase_one_rule;       /* relp_pivot_rule_t */
    int32_t phase_two_rule;
    double  tol_cost;             /* candidate iff d_j < -tol_cost          (pivot_rule.rs:56,81,117) */
    double  tol_pivot;            /* ratio test iff alpha_i > tol_pivot      (tableau/mod.rs:227).  Defaults:
                                   * tol_cost = tol_feas = 1e-7, tol_pivot = 1e-5, tol_tie = 1e-9, tol_zero = 1e-11.
                                   * tol_pivot is larger than the others because an alpha_i that is exactly 0 in the
                                   * reference's rational field carries the rounding noise of B^-1 in f64 (1e-7 .. 1e-6 on
                                   * unscaled Netlib data such as 25FV47, whose entries span 1e-3 .. 1e4): a pivot on
                                   * such an element destroys the basis inverse */
    double  tol_zero;             /* |b_i| <= tol_zero reads as 0 in the ratio */
    double  tol_tie;              /* ratio ties: <= min + tol_tie*max(1,|min|), smallest leaving column wins
                                     (tableau/mod.rs:229-239) */
    double  tol_feas;             /* phase 1 feasible iff |obj| <= tol_feas*max(1, initial obj) (phase_one.rs:146) */
    int32_t poll_interval;        /* relp_run: iterations enqueued between host polls of the outcome */
    int32_t trace_capacity;       /* pivots recorded on the device (0 = no trace) */
    /* column / row sharding for multi-GPU pricing; 0,1 = everything local */
    int32_t shard_rank, shard_count;
    /* basis-inverse maintenance: 0 = rank-1 update of the explicit inverse at every pivot
     * (basis_inverse_rows.rs:131-142 literally); K > 0 = deferred: the explicit inverse is kept as
     * (I + W S') B0inv and the K most recent pivots are folded in by one m x K x m GEMM ("flush");
     * -1 = automatic (64 when m >= 4096, else 0; the dense tableau engine, which always works in blocks: 64, and 96 for
     * tableaus of 40,000 columns or more with m >= 4096).  Results are the same up to f64 rounding. */
    int32_t update_block;
    /* relp_engine_kind_t: which device representation maintains the basis inverse */
    int32_t engine;
    /* ---- f64 safeguards the exact reference has no need for; 0 (relp_default_config) = the reference's rule ----
     * relp_ratio_rule_t.  RELP_RATIO_LARGEST_PIVOT: among the rows inside the tie band of the minimum ratio the largest pivot
     * element wins (compared at float precision), then the lowest leaving column; tableau/mod.rs:229-239 has only the latter.
     * On degenerate LPs with badly scaled columns (Netlib GREENBEA / GREENBEB, which the reference `#[ignore]`s) the
     * reference's choice among dozens of ratio-0 rows regularly lands on a pivot of 1e-4 beside candidates of 1e+2; exact
     * arithmetic does not care, an f64 basis inverse is destroyed within ~1,000 pivots. */
    int32_t ratio_rule;
    /* relp_artificial_removal_t.  RELP_ARTIFICIAL_TEXTBOOK: at the end of phase 1 a basic artificial variable is pivoted out
     * in the row it is basic in (phase_one.rs:236 takes the row it started in: an artificial that re-entered the basis in
     * another row survives into phase 2 as a free column on that row -- Netlib GREENBEA ends 73,482 below its optimum that
     * way), on the first non-basic column with a non-zero element in that row, whatever that column's reduced cost (the
     * pivot is at zero level either way; phase_one.rs:239-244 only tries columns whose reduced cost is exactly zero), and
     * where there is no such column the artificial's OWN constraint is removed together with the basis position it sits in
     * (the two positions are exchanged first -- rows of B^-1 / of the tableau, b, the basis array -- so that one index names
     * both: that pair always leaves a basis of the filtered problem; phase_one.rs:252 pushes the artificial's INDEX, which
     * names another row as soon as `<=` rows lie in front of the artificial's row: the reference then deletes a
     * non-redundant row and optimises a relaxation -- Netlib 80BAU3B ends at 964,593.50 instead of 987,224.19 that way). */
    int32_t artificial_removal;
    /* RELP_PIVOT_RESCUE (1): a pivot loop that ends without a pivot row is looked at before it is believed.  The reference
     * takes every alpha_i > 0 (tableau/mod.rs:227); f64 needs `tol_pivot`, and an ABSOLUTE one is wrong both ways on unscaled
     * data: Netlib SIERRA (entries up to 1e5) has legitimate columns whose largest entry is 1e-5 -- all of it below the
     * tolerance, phase 1 "unbounded" after 620 pivots on every engine -- while SCSD6 prices a column whose reduced cost
     * (-2.2e-7) and positive entries (2.2e-7 beside entries of -5) are both artefacts of its six-digit data.  With the
     * rescue the engine reads alpha back: positive entries of at least tol_pivot * max|alpha| are pivoted on (one step with the
     * tolerance lowered to that), otherwise the column is barred from pricing until the basis has changed, and an outcome
     * reached with barred columns is confirmed once more without them.  Rare events, handled on the host. */
    int32_t pivot_rescue;
    /* 1: the explicit inverse / the tableau is rebuilt from the basis columns at an interval the engine adapts itself -- halved
     * (down to 32 pivots) when a rebuild moves b by more than 1e-7 relative, doubled (up to 1,024) when by less than 1e-10 --
     * instead of the hand-set relp_set_reinversion_interval (the LU engine refactorises every update_block pivots anyway). */
    int32_t auto_reinversion;
} relp_config_t;

typedef enum { RELP_RATIO_REFERENCE = 0, RELP_RATIO_LARGEST_PIVOT = 1 } relp_ratio_rule_t;
typedef enum { RELP_ARTIFICIAL_REFERENCE = 0, RELP_ARTIFICIAL_TEXTBOOK = 1 } relp_artificial_removal_t;

typedef enum {
    /* revised simplex with the explicit dense inverse (`Carry<_, BasisInverseRows<_>>`): PRICE and FTRAN
     * are HBM streams over A and B^-1 at every pivot */
    RELP_ENGINE_REVISED = 0,
    /* dense tableau T = B^-1 [A | slacks] kept as (I + W S') T0: PRICE is one tableau row, FTRAN one
     * tableau column per pivot, and T0 is updated by an m x K x n GEMM on the f64 matrix cores every
     * update_block pivots (needs 8 m n bytes; same pivots as the revised engine up to f64 rounding).
     * relp_from_basis re-tabulates T0 = B^-1 [A | I] from a factorisation of the given basis.  Column-sharded across GPUs
     * through relp_shard_* / relp_shard_run: both
     * phases, artificial removal and redundant-row removal included (the revised engine shards only LPs with a full
     * slack basis). */
    RELP_ENGINE_TABLEAU = 1,
    RELP_ENGINE_LU = 2,       /* sparse LU of the basis + pending updates (Carry<_, LUDecomposition<_>>) */
    /* resolved at create (relp_engine_kind tells which): the dense tableau while it fits comfortably -- 8 m n <= 64 GB and
     * m <= 50,000 rows, where it is the fastest engine on every file of the reference -- the LU engine beyond,
     * where its O(nnz) memory and the persistent kernel's third layout win (5.3b) */
    RELP_ENGINE_AUTO = 3
} relp_engine_kind_t;

/* relp_default_config: the reference's rules literally (every safeguard 0, engine RELP_ENGINE_REVISED).
 * relp_robust_config: what an f64 caller who wants the optimum rather than the reference's pivot sequence should take --
 * ratio_rule = RELP_RATIO_LARGEST_PIVOT, artificial_removal = RELP_ARTIFICIAL_TEXTBOOK, pivot_rescue, auto_reinversion,
 * engine = RELP_ENGINE_AUTO; no per-file knobs (tests/test_gpu_big_pins.py, tests/test_gpu_corpus.py). */
void relp_default_config(relp_config_t *cfg);
void relp_robust_config(relp_config_t *cfg);
/* the engine in use (RELP_ENGINE_AUTO resolved) */
int32_t relp_engine_kind(const relp_engine_t *h);
/* RELP_PIVOT_RESCUE / auto_reinversion at work: out[4] = { pivots made with a lowered tolerance, columns barred, outcomes
 * confirmed without barred columns, the re-inversion interval in effect } */
relp_status_t relp_robust_stats(const relp_engine_t *h, int64_t *out4);
const char *relp_last_error(const relp_engine_t *h);
const char *relp_version(void);

/* ---- construction ----------------------------------------------------------------------------
 * Tableau::<_, Partially<_>>::new (kind/artificial/partially.rs:125-206) +
 * InverseMaintener::create_for_partially_artificial (carry/mod.rs:381-426): B^-1 = I, b = rhs,
 * -pi_i = -1 on artificial rows, -obj = -sum of b over artificial rows. */
relp_status_t relp_engine_create(const relp_matrix_data_t *md, const relp_config_t *cfg, relp_engine_t **out);
void          relp_engine_destroy(relp_engine_t *h);
relp_status_t relp_set_stream(relp_engine_t *h, void *hip_stream);

/* ---- one pivot, step by step (the body of phase_one.rs:135-146 / phase_two.rs:32-50) ---------- */
/* PivotRule::select_primal_pivot_column (strategy/pivot_rule.rs:25-33): full PRICE on the device. */
relp_status_t relp_select_primal_pivot_column(relp_engine_t *h, int32_t rule, int32_t *found,
                                              int32_t *column, double *relative_cost);
/* Tableau::relative_cost (tableau/mod.rs:102-108) for every column (basic columns included). */
relp_status_t relp_relative_costs(relp_engine_t *h, double *out_n);
/* Tableau::generate_column (tableau/mod.rs:122-126) = BasisInverse::generate_column
 * (basis_inverse_rows.rs:144-155): FTRAN; result stays on the device until the next call;
 * `out_m` (may be NULL) receives the dense column. */
relp_status_t relp_generate_column(relp_engine_t *h, int32_t column, double *out_m);
/* Tableau::generate_element (tableau/mod.rs:129-134). */
relp_status_t relp_generate_element(relp_engine_t *h, int32_t row, int32_t column, double *out);
/* Tableau::select_primal_pivot_row (tableau/mod.rs:221-247) on the last generated column. */
relp_status_t relp_select_primal_pivot_row(relp_engine_t *h, int32_t *found, int32_t *row);
/* The same ratio test on a column the caller supplies (dense, m entries, host memory), the reference's signature
 * `select_primal_pivot_row(&self, column: &SparseVector)`; leaves the last generated column untouched. */
relp_status_t relp_select_primal_pivot_row_of(relp_engine_t *h, const double *column_m, int32_t *found, int32_t *row);
/* Tableau::bring_into_basis (tableau/mod.rs:47-60) -> Carry::change_basis (carry/mod.rs:549-570):
 * consumes the last generated column; returns the leaving column. */
relp_status_t relp_bring_into_basis(relp_engine_t *h, int32_t column, int32_t row, double relative_cost,
                                    int32_t *leaving_column);

/* ---- whole loops on the device ------------------------------------------------------------------
 * relp_run: up to max_iters basis changes of the current phase with no per-pivot host sync.
 * phase_one::primal (phase_one.rs:125-170) / phase_two::primal (phase_two.rs:22-51). */
relp_status_t relp_run(relp_engine_t *h, int64_t max_iters, int64_t *iterations_done, int32_t *outcome);
/* SolveRelaxation::solve_relaxation (two_phase/mod.rs:30-76): phase 1, artificial removal
 * (phase_one.rs:223-260), phase switch (kind/non_artificial.rs:151-220), phase 2. */
relp_status_t relp_solve_relaxation(relp_engine_t *h, int64_t max_iters, int32_t *outcome);
/* InverseMaintener::from_basis (carry/mod.rs:428-463): warm start from provider column indices,
 * one per row; switches to phase 2.  RELP_ENGINE_LU: any basis (factorise, b = FTRAN(rhs), -pi = BTRAN(-c_B));
 * RELP_ENGINE_REVISED: any basis (basis_inverse_rows.rs:103-129: LU - factorised on the host like every
 * refactorisation - then m unit solves, b and -pi on the device; slack bases are a signed permutation and take a
 * shortcut); RELP_ENGINE_TABLEAU (unsharded): any basis -- the tableau B^-1 [A | I] is re-tabulated column by column from the
 * factors, then b, the reduced costs and -obj (tests/cpp/test_tableau.cpp, tests/test_gpu_parity.py::test_from_basis_*). */
relp_status_t relp_from_basis(relp_engine_t *h, const int32_t *basis_columns_m);
/* RELP_ENGINE_REVISED and RELP_ENGINE_TABLEAU (unsharded), an f64 matter (the exact reference never needs it,
 * `should_refactor() = false`, basis_inverse_rows.rs:175-179): every `pivots` basis changes inside relp_run the state is
 * rebuilt from the columns of the current basis -- host LU, then on the device B^-1 row by row (revised) or the tableau
 * B^-1 [A | I] column by column (tableau), b = B^-1 rhs, -pi / the reduced costs and -obj -- which bounds the error of a
 * representation that is otherwise only ever updated.  Default: 1,000 for sparse input (CSC, at most 10 % nonzeros)
 * with at most 4,096 rows -- where the host factorisation of a basis is cheap -- else 0 = never.
 * relp_reinversions: how many have been done. */
relp_status_t relp_set_reinversion_interval(relp_engine_t *h, int64_t pivots);
int64_t       relp_reinversions(const relp_engine_t *h);
/* Fold every pending deferred update into the stored representation: B0^-1 += W (S' B0^-1) (revised), T0 += W R0
 * (tableau), refactorisation (LU).  No-op when update_block = 0. */
relp_status_t relp_flush(relp_engine_t *h);
/* The block size K in effect (0 = explicit rank-1 updates; LU engine: pivots between refactorisations). */
int32_t       relp_update_block(const relp_engine_t *h);
/* RELP_ENGINE_LU only: statistics of the current factorisation, out[8] = { refactorisations so far, m,
 * nnz(L) (off-diagonal), nnz(U) (with diagonal), levels of the four solve schedules L, U (FTRAN) and
 * U', L' (BTRAN) }.  Levels bound the length of the dependent chain of a triangular solve. */
relp_status_t relp_lu_stats(const relp_engine_t *h, int64_t *out8);
/* RELP_ENGINE_LU: the look-ahead refactorisation (a refactorisation prepared on the host while the pivot kernel keeps
 * going on the old factors; carry/mod.rs:602-614 refactorises synchronously): out[4] = { look-ahead factorisations
 * installed, basis changes replayed onto them, the look-ahead length in effect (RELP_LU_LOOKAHEAD, 0 = off), the lane
 * budget of a fused group of levels (RELP_FUSE_LANES) }; both switches are read when the engine is created. */
relp_status_t relp_lu_lookahead_stats(const relp_engine_t *h, int64_t *out4);
/* RELP_ENGINE_LU: how the pivot loop runs.  out[4] = { persistent pivot kernel in use (0: the product-form fallback, one launch
 * per step), its layout (0: work vectors, eta file and permutations in one CU's LDS; 1: x, -pi and the slot tables in LDS, the
 * rest in L2; 2: nothing per row in LDS, any m), slots of the dense tail of U (= the longest refactorisation interval), PRICE as
 * a grid launch per pivot (Dantzig's rule over >= 32,768 columns in layout 2; RELP_FT_GRID_PRICE) }.  RELP_FT_BIG = 0 / 1 / 2
 * forces that layout at create: when the forced layout does not fit, no other layout is tried and the engine runs the
 * product-form fallback (out[0] = 0) -- bench.py's `lu_product_form_fallback` leg relies on exactly that. */
relp_status_t relp_lu_kernel_layout(const relp_engine_t *h, int32_t *out4);
/* RELP_ENGINE_LU: refactorise ON THE DEVICE (LUDecomposition::invert -> decomposition/mod.rs:27-138 with the Markowitz
 * pivoting of decomposition/pivoting.rs:45-81): singleton rows / columns peeled in parallel rounds, the bump eliminated on
 * a dense working copy by one workgroup, L and U read off afterwards -- no basis download, no search on the host.  Off by
 * default (the host factorisation behind the look-ahead is faster at Netlib sizes); RELP_LU_DEVICE_FACTOR=1
 * switches it on at create.  A bump beyond the working copy (RELP_LUF_BUMP_CAP, 2,048) is factorised on the host.
 * stats: out[6] = { enabled, device factorisations, fallbacks to the host, microseconds spent in the kernel so far,
 * bump size and number of peeled singleton pivots of the last factorisation }. */
relp_status_t relp_lu_set_device_factorisation(relp_engine_t *h, int32_t on);
relp_status_t relp_lu_device_factorisation_stats(const relp_engine_t *h, int64_t *out6);
/* max |P B Q - L U| over all entries for the factors in use and the current basis (decomposition/mod.rs:301-491 asserts
 * the factors themselves; pivot orders differ, the identity is what they have in common).  Dense arithmetic on the host:
 * m <= 1,024, else *out = -1.  The factors describe the basis of the last refactorisation: with Forrest-Tomlin updates
 * pending the comparison would be against another basis, and *out = -1 as well (call it right after relp_from_basis /
 * a refactorisation). */
relp_status_t relp_lu_factor_residual(relp_engine_t *h, double *out);

/* RELP_ENGINE_LU in Forrest-Tomlin mode: shader clocks spent per phase of the pivot inside the persistent kernel since create
 * (thread 0): out[16] = { PRICE, entering-column scatter, L solve, eta file forward, spike push, U solve, ratio test, b update,
 * u_bar extraction, U' solve for r, compaction of r and the spike, eta file backward, L' solve, -pi / basis, state load + store,
 * 0 }.  Measurement aid for profiles/ and DESIGN.md. */
relp_status_t relp_lu_phase_cycles(relp_engine_t *h, int64_t *out16);

/* ---- the `BasisInverse` surface (carry/mod.rs:68-157) beside the tableau-level calls above ------------------------
 * BasisInverse::basis_inverse_row(row) (carry/mod.rs:145-150): row `row` of B^-1, dense, m entries.  LU engine: one BTRAN
 * of e_row (lower_upper/mod.rs:204-222); revised engine: a copy of the stored row (basis_inverse_rows.rs:181-183);
 * tableau engine: row `row` of the tableau over the columns that were the identity. */
relp_status_t relp_basis_inverse_row(relp_engine_t *h, int32_t row, double *out_m);
/* BasisInverse::should_refactor (carry/mod.rs:138-143).  LU engine: 1 when as many updates are pending as
 * relp_config_t.update_block allows (lower_upper/mod.rs:199-202 refactors when updates.len() > 10: update_block = 11) or the
 * eta pool is exhausted; relp_run / relp_bring_into_basis refactor by themselves (Carry::after_basis_change,
 * carry/mod.rs:602-614).  Revised and tableau engine: always 0 (basis_inverse_rows.rs:175-179). */
relp_status_t relp_should_refactor(relp_engine_t *h, int32_t *out);
/* BasisInverse::generate_column(original_column) (carry/mod.rs:108-118) for a column the caller supplies as sorted
 * (row, value) pairs over the m tableau rows -- the reference's signature, where relp_generate_column takes a column
 * index of the provider.  The result also becomes the "last generated column" (relp_select_primal_pivot_row,
 * relp_lu_change_basis).  Tableau engine: computed on the host from B^-1 read off the tableau. */
relp_status_t relp_generate_column_of(relp_engine_t *h, const int32_t *row_idx, const double *values, int32_t nnz, double *out_m);
/* InverseMaintener::cost_difference(column) (carry/mod.rs:572-577): (-pi) . column */
relp_status_t relp_cost_difference_of(relp_engine_t *h, const int32_t *row_idx, const double *values, int32_t nnz, double *out);
/* BasisInverse::change_basis(pivot_row_index, column) on the inverse alone (lower_upper/mod.rs:92-155): the Forrest-Tomlin
 * update with the column (and spike, mod.rs:378-381) of the last relp_generate_column / relp_generate_column_of.  b, -pi and
 * the basis indices are NOT touched (that is Carry::change_basis = relp_bring_into_basis).  LU engine only. */
relp_status_t relp_lu_change_basis(relp_engine_t *h, int32_t pivot_row_index);
/* `LUDecomposition { row_permutation: identity, column_permutation: identity, lower_triangular, upper_triangular, updates: [] }`
 * given literally (lower_upper/mod.rs:33-57), the way the reference's tests construct their starting points: L column-major
 * with the unit diagonal implied, U column-major with its diagonal; CSC arrays of m columns each. */
relp_status_t relp_lu_set_factors(relp_engine_t *h, const int64_t *l_col_ptr, const int32_t *l_row_idx, const double *l_values,
                                  const int64_t *u_col_ptr, const int32_t *u_row_idx, const double *u_values);
/* The update file in the reference's coordinates: `updates.len()`; `updates[k]` = (EtaFile { values, pivot, len },
 * RotateToBack { index: pivot }) with positions as they were before that rotation (eta_file.rs:14-18, mod.rs:56);
 * `upper_triangular` after every update so far: column-major, rows sorted, diagonal last in its column (mod.rs:48-52). */
relp_status_t relp_lu_updates(relp_engine_t *h, int32_t *count);
relp_status_t relp_lu_get_update(relp_engine_t *h, int32_t k, int32_t *pivot, int32_t *idx, double *values, int32_t cap, int32_t *nnz);
relp_status_t relp_lu_get_upper(relp_engine_t *h, int64_t *col_ptr_m1, int32_t *row_idx, double *values, int64_t cap, int64_t *nnz);

/* ---- state getters (InverseMaintener accessors, inverse_maintenance/mod.rs:200-266) ----------- */
int32_t relp_nr_rows(const relp_engine_t *h);
int32_t relp_nr_columns(const relp_engine_t *h);
int32_t relp_phase(const relp_engine_t *h);                 /* 1 or 2 */
int32_t relp_nr_artificial(const relp_engine_t *h);
relp_status_t relp_get_objective(relp_engine_t *h, double *out);        /* get_objective_function_value */
relp_status_t relp_get_b(relp_engine_t *h, double *out_m);              /* b() */
relp_status_t relp_get_minus_pi(relp_engine_t *h, double *out_m);
relp_status_t relp_get_basis_indices(relp_engine_t *h, int32_t *out_m); /* basis_column_index_for_row */
relp_status_t relp_get_basis_inverse(relp_engine_t *h, double *out_mm); /* dense row-major B^-1 */
/* current_bfs (carry/mod.rs:616-625): (column, value) pairs sorted by column, zeros dropped. */
relp_status_t relp_current_bfs(relp_engine_t *h, int32_t *cols, double *vals, int32_t cap, int32_t *count);
relp_status_t relp_get_iterations(relp_engine_t *h, int64_t *out);
/* how many of those pivots were degenerate: ratio b_r / alpha_r exactly 0, the basis changed but b did not */
relp_status_t relp_get_degenerate_pivots(relp_engine_t *h, int64_t *out);
/* recorded pivots: phase, entering, row, leaving (each `cap` long); count = pivots so far */
relp_status_t relp_get_trace(relp_engine_t *h, int32_t *phase, int32_t *entering, int32_t *row,
                             int32_t *leaving, int64_t cap, int64_t *count);
/* Debug invariant checker (tableau/mod.rs:253-289): max |B^-1 B - I|, max |d_basic|, min b. */
relp_status_t relp_check_basis(relp_engine_t *h, double *max_identity_error, double *max_basic_cost,
                               double *min_b);

/* ---- per-kernel timing (bench.py roofline) ------------------------------------------------------ */
typedef enum {
    RELP_K_PRICE = 0, RELP_K_SELECT_COLUMN = 1, RELP_K_BUILD_COLUMN = 2, RELP_K_FTRAN = 3,
    RELP_K_RATIO = 4, RELP_K_UPDATE_VECTORS = 5, RELP_K_UPDATE_INVERSE = 6,
    RELP_K_APPLY_W = 7,          /* deferred update: alpha = v + W (S'v) */
    RELP_K_UPDATE_W = 8,         /* deferred update: W <- E W, pivot row rho */
    RELP_K_FLUSH = 9,            /* deferred update: B0inv += W (S' B0inv); LU engine: refactorisation */
    RELP_K_FT_RUN = 10,          /* LU engine: the persistent pivot kernel (many pivots per launch) */
    RELP_K_COUNT = 11
} relp_kernel_id_t;
/* When enabled, the kernel classes of every `sample_every`-th pivot inside relp_run are bracketed by
 * HIP events on the engine's stream (an event pair costs ~4 us of stream time, so bracketing every
 * launch would slow a 330 us pivot by 15 %); relp_profile_read sums them (this synchronises). */
relp_status_t relp_profile_enable(relp_engine_t *h, int32_t enable, int64_t max_launches, int32_t sample_every);
relp_status_t relp_profile_read(relp_engine_t *h, int32_t kernel_id, int64_t *launches, double *total_ms);

/* ---- synthetic workloads (bench / tests; rust-lp_amd/synthetic.py defines the numbers) ----------
 * Fills a column-major m x n matrix on the device with A[i,j] = (1 + x(0, j*m+i) % 999) / 1000. */
relp_status_t relp_synth_fill_dense(double *device_a, int64_t ld, int32_t m, int32_t n, uint64_t seed,
                                    int64_t first_column, void *hip_stream);
relp_status_t relp_device_alloc(void **out, int64_t bytes);
relp_status_t relp_device_free(void *p);

/* ---- multi-GPU shards: structural columns [col_lo, col_hi) and rows
 * [row_lo, row_hi) of B^-1 live on this rank; b, -pi, basis are replicated.  Buffers are device
 * pointers the caller exchanges with RCCL on the engine's stream; no host sync inside. ----------- */
relp_status_t relp_shard_ranges(const relp_engine_t *h, int32_t *col_lo, int32_t *col_hi,
                                int32_t *row_lo, int32_t *row_hi, int32_t *row_slice_stride);
/* columns [lo, hi) of the structural block owned by `rank` of `count` (what relp_engine_create
 * expects in `dense` when cfg.shard_count > 1) */
void          relp_shard_column_range(int32_t nr_normal, int32_t rank, int32_t count, int32_t *lo, int32_t *hi);
/* message length in doubles of one PRICE candidate: [key, j, d_j, a_j (m entries, tableau row space)]; the
 * tableau engine appends the minimum ratio b_i / alpha_i of every block of 256 rows (the receiver's ratio test
 * starts from those) */
int64_t       relp_shard_candidate_len(const relp_engine_t *h);
/* length in doubles of the rho buffer (m rounded up to the B^-1 row pitch) */
int64_t       relp_shard_rho_len(const relp_engine_t *h);
/* local PRICE over the owned columns (+ every virtual column) -> candidate message */
relp_status_t relp_shard_price(relp_engine_t *h, double *dev_candidate);
/* choose the global entering column from `count` gathered candidates (min d, then min j) */
relp_status_t relp_shard_select_column(relp_engine_t *h, const double *dev_candidates, int32_t count);
/* FTRAN slice: alpha[row_lo:row_hi) into dev_alpha_slice (row_slice_stride entries, zero padded) */
relp_status_t relp_shard_ftran(relp_engine_t *h, double *dev_alpha_slice);
/* ratio test on the gathered alpha (count slices of row_slice_stride); the owner of the pivot row
 * writes rho = row_r(B^-1)/alpha_r into dev_rho (m entries), every other rank writes zeros, so a
 * SUM all-reduce of dev_rho is the broadcast */
relp_status_t relp_shard_ratio(relp_engine_t *h, const double *dev_alpha_slices, int32_t count, double *dev_rho);
/* rank-1 update of the owned rows + replicated b, -pi, obj, basis */
relp_status_t relp_shard_update(relp_engine_t *h, const double *dev_rho);
/* structural columns [lo, hi) that rank cfg->shard_rank must supply in `dense` for this problem and
 * engine kind (only the counts and upper bounds of `md` are read).  The revised engine splits the
 * structural columns; the tableau engine splits its stored columns [artificial | structural | virtual]. */
relp_status_t relp_shard_plan(const relp_matrix_data_t *md, const relp_config_t *cfg, int32_t *col_lo, int32_t *col_hi);
/* Tableau engine only: the whole pivot after relp_shard_select_column (ratio test, row update of the
 * owned columns, W / b / basis update, local flush when due).  One all-gather per pivot in total. */
relp_status_t relp_shard_pivot(relp_engine_t *h);
/* deferred update in sharded mode, every relp_update_block() pivots: `begin` snapshots the rows
 * S' B0inv this rank owns (zeros elsewhere) and returns the buffer to SUM all-reduce in place
 * (*len_doubles = 0: nothing pending); `end` applies W (S' B0inv) to the owned rows. */
relp_status_t relp_shard_flush_begin(relp_engine_t *h, double **dev_snapshot, int64_t *len_doubles);
relp_status_t relp_shard_flush_end(relp_engine_t *h);
/* outcome poll (one small device->host copy) */
relp_status_t relp_poll(relp_engine_t *h, int32_t *outcome, int64_t *iterations);

/* ---- native multi-GPU loop: the same per-pivot sequence as above, enqueued by the library itself with the
 * collectives called through two hooks between its kernels, on the engine's stream (no Python and no host
 * sync inside a pivot; the host polls the outcome every cfg.poll_interval pivots like relp_run).  The hooks
 * return 0 on success. */
typedef int (*relp_allgather_fn)(void *ctx, const void *dev_send, void *dev_recv, int64_t bytes_per_rank, void *hip_stream);
typedef int (*relp_allreduce_sum_fn)(void *ctx, double *dev_buf, int64_t count, void *hip_stream);
relp_status_t relp_shard_set_collectives(relp_engine_t *h, relp_allgather_fn allgather, relp_allreduce_sum_fn allreduce_sum,
                                         void *ctx);
/* phase_one::primal / phase_two::primal across ranks: up to max_iters basis changes of the current phase.  Every
 * rank must call it with the same arguments; every rank takes the same decisions from the same gathered data,
 * so *done and *outcome agree on all ranks. */
relp_status_t relp_shard_run(relp_engine_t *h, int64_t max_iters, int64_t *done, int32_t *outcome);
/* Failure agreement: a rank whose step fails locally (anything but a collective) keeps the collectives of the current chunk
 * going and reports at the next poll, where one 16-byte all-gather of the statuses makes every rank return the same error:
 * nobody is left waiting inside a collective.  relp_shard_inject_failure is the test hook for exactly that: this engine's
 * relp_shard_run fails locally after `after_pivots` further pivots. */
relp_status_t relp_shard_inject_failure(relp_engine_t *h, int64_t after_pivots);
/* RCCL as the collectives: rank 0 makes a unique id (ncclGetUniqueId, 128 bytes), the caller hands it to every
 * rank (any channel; bench.py broadcasts it with torch.distributed), and each rank attaches a communicator of
 * cfg.shard_count ranks on the engine's device (ncclCommInitRank), which installs ncclAllGather / ncclAllReduce
 * as the hooks.  librccl.so.1 is taken from the process if already loaded (PyTorch ships one), else dlopen'ed.
 * RELP_E_UNSUPPORTED when no RCCL library can be loaded. */
#define RELP_RCCL_ID_BYTES 128
relp_status_t relp_rccl_unique_id(uint8_t id[RELP_RCCL_ID_BYTES]);
relp_status_t relp_rccl_attach(relp_engine_t *h, const uint8_t id[RELP_RCCL_ID_BYTES]);

#ifdef __cplusplus
}
#endif
#endif /* RELP_ENGINE_H */
