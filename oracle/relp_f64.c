/*
 * relp_f64.c -- single-threaded C (f64) restatement of RELP's pivot path with the explicit
 * row-major basis inverse (`Carry<F, BasisInverseRows<F>>`).  See relp_oracle.h for scope and the
 * "test infrastructure only" rule.  Citations are file:line under /root/reference/src/algorithm/two_phase/.
 *
 * Data structures follow the reference on purpose (this file doubles as the reported CPU
 * baseline): sparse vectors are sorted arrays of 16-byte (index, value) tuples with exact zeros
 * removed; b and -pi are dense; PRICE clones every column before the dot product
 * (matrix_provider/matrix_data.rs:318); row reduction allocates a fresh vector per edited row
 * (data/linear_algebra/vector/sparse.rs:213-248).
 */
#include "relp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { int64_t idx; double val; } tup;          /* Rust (usize, f64) */
typedef struct { tup *d; int64_t n, cap; } svec;

static void sv_init(svec *v) { v->d = NULL; v->n = 0; v->cap = 0; }
static void sv_free(svec *v) { free(v->d); v->d = NULL; v->n = v->cap = 0; }
static void sv_reserve(svec *v, int64_t cap) {
    if (cap > v->cap) {
        int64_t nc = v->cap ? v->cap * 2 : 8;
        if (nc < cap) nc = cap;
        v->d = (tup *)realloc(v->d, (size_t)nc * sizeof(tup));
        v->cap = nc;
    }
}
static void sv_push(svec *v, int64_t idx, double val) {
    sv_reserve(v, v->n + 1);
    v->d[v->n].idx = idx; v->d[v->n].val = val; v->n++;
}
static void sv_clear(svec *v) { v->n = 0; }
static const double *sv_get(const svec *v, int64_t idx) {
    int64_t lo = 0, hi = v->n;
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (v->d[mid].idx < idx) lo = mid + 1; else hi = mid; }
    return (lo < v->n && v->d[lo].idx == idx) ? &v->d[lo].val : NULL;
}

#include "relp_f64_lu.h"

struct oracle_engine {
    /* ---- MatrixData (matrix_provider/matrix_data.rs:54-90) ---- */
    int32_t nr_normal, nr_eq, nr_range, nr_le, nr_ge;
    int64_t *col_ptr; int32_t *row_idx; double *values;
    double *b0, *ranges, *cost, *upper;
    int32_t nr_bounds;               /* variables with an upper bound */
    int32_t *var_to_bound;           /* -1 = none (matrix_data.rs:168-177) */
    int32_t *bound_to_var;
    int32_t col_start[7], row_start[7];
    /* rank-deficient view (filter/generic_wrapper.rs:51): sorted rows deleted from the provider */
    int32_t nr_filtered; int32_t *filtered;
    int32_t nr_zero_level;                  /* pivots made by remove_artificial_basis_variables */
    int32_t nr_exchanges;                   /* stuck artificials moved into their own row (artificial_removal 1) */

    oracle_config_t cfg;

    /* ---- Kind ---- */
    int32_t phase;                   /* 1 = Partially artificial, 2 = NonArtificial */
    int32_t nr_artificial;
    int32_t *column_to_row;          /* partially.rs:21 */

    /* ---- Carry (carry/mod.rs:45-65) ---- */
    int32_t m;
    double minus_objective;
    double *minus_pi, *b;
    int32_t *basis_indices;
    svec *rows;                      /* BasisInverseRows (basis_inverse_rows.rs:20); NULL with the LU back-end */
    lu_t *lu;                        /* LUDecomposition (lower_upper/mod.rs:35-57), cfg.basis_inverse = 1 */
    svec spike;                      /* ColumnAndSpike::spike of the last generate_column (lower_upper/mod.rs:376-391) */
    int64_t lu_refactorisations; int lu_failed;
    int32_t wrapped_na;              /* artificial columns of phase 1 (a survivor keeps the wrapped index INT32_MAX - (na - 1 - a)) */

    /* ---- Tableau (tableau/mod.rs:24-38): basis membership set ---- */
    uint8_t *in_basis; int32_t n_flags;

    /* pivot-rule state (pivot_rule.rs:62-64) */
    int64_t last_selected;           /* -1 = None */
    double initial_phase1_objective;
    svec scratch_col, scratch_alpha;
};

/* ------------------------------------------------------------------------------------------ */
/* MatrixData                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static int32_t md_nr_constraints(const oracle_engine_t *e) { return e->nr_eq + e->nr_range + e->nr_le + e->nr_ge; }
static int32_t md_nr_rows_full(const oracle_engine_t *e) { return md_nr_constraints(e) + e->nr_bounds + e->nr_range; }
static int32_t md_nr_columns(const oracle_engine_t *e) {
    return e->nr_normal + e->nr_range + e->nr_le + e->nr_ge + e->nr_bounds + e->nr_range;
}

/* matrix_data.rs:198-222 */
static void md_column_type(const oracle_engine_t *e, int32_t j, int *group, int32_t *k) {
    int g = 0;
    while (g + 1 < 6 && j >= e->col_start[g + 1]) g++;
    *group = g; *k = j - e->col_start[g];
}

/* utilities.rs:47-69 applied to a freshly built column (generic_wrapper.rs:224-284) */
static void filter_column(const oracle_engine_t *e, svec *col) {
    if (e->nr_filtered == 0) return;
    int64_t out = 0; int32_t skipped = 0;
    for (int64_t k = 0; k < col->n; k++) {
        int64_t i = col->d[k].idx;
        while (skipped < e->nr_filtered && e->filtered[skipped] < i) skipped++;
        if (skipped < e->nr_filtered && e->filtered[skipped] == i) continue;
        col->d[out].idx = i - skipped; col->d[out].val = col->d[k].val; out++;
    }
    col->n = out;
}

/* matrix_data.rs:308-348: builds (clones) column j of the provider */
static void md_column(const oracle_engine_t *e, int32_t j, svec *out) {
    int g; int32_t k;
    md_column_type(e, j, &g, &k);
    sv_clear(out);
    switch (g) {
    case 0: {
        int64_t s = e->col_ptr[k], t = e->col_ptr[k + 1];
        sv_reserve(out, t - s + 1);
        for (int64_t p = s; p < t; p++) { out->d[out->n].idx = e->row_idx[p]; out->d[out->n].val = e->values[p]; out->n++; }
        if (e->var_to_bound[k] >= 0) sv_push(out, md_nr_constraints(e) + e->var_to_bound[k], 1.0);
        break; }
    case 1: sv_push(out, e->row_start[1] + k, 1.0); sv_push(out, e->row_start[5] + k, 1.0); break;
    case 2: sv_push(out, e->row_start[2] + k, 1.0); break;
    case 3: sv_push(out, e->row_start[3] + k, -1.0); break;
    case 4: sv_push(out, e->row_start[4] + k, 1.0); break;
    default: sv_push(out, e->row_start[5] + k, 1.0); break;
    }
    filter_column(e, out);
}

/* matrix_data.rs:350-357 (None == 0 for slacks) */
static double md_cost_value(const oracle_engine_t *e, int32_t j) {
    return j < e->nr_normal ? e->cost[j] : 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* Kind                                                                                        */
/* ------------------------------------------------------------------------------------------ */
static int32_t kind_nr_columns(const oracle_engine_t *e) {
    return (e->phase == 1 ? e->nr_artificial : 0) + md_nr_columns(e);
}
/* partially.rs:72-80, non_artificial.rs:58-62 */
static void kind_original_column(const oracle_engine_t *e, int32_t j, svec *out) {
    if (e->phase == 1) {
        if (j < e->nr_artificial) { sv_clear(out); sv_push(out, e->column_to_row[j], 1.0); return; }
        md_column(e, j - e->nr_artificial, out);
    } else {
        md_column(e, j, out);
    }
}
/* partially.rs:52-60, non_artificial.rs:42-46 */
static double kind_initial_cost(const oracle_engine_t *e, int32_t j) {
    if (e->phase == 1) return j < e->nr_artificial ? 1.0 : 0.0;
    return md_cost_value(e, j);
}

/* ------------------------------------------------------------------------------------------ */
/* Tableau / Carry                                                                             */
/* ------------------------------------------------------------------------------------------ */
/* tableau/mod.rs:102-108 + carry/mod.rs:572-577 + vector/dense.rs:81-92 */
static double relative_cost(oracle_engine_t *e, int32_t j) {
    svec *col = &e->scratch_col;
    kind_original_column(e, j, col);
    double total = 0.0;
    for (int64_t k = 0; k < col->n; k++) total += e->minus_pi[col->d[k].idx] * col->d[k].val;
    return total + kind_initial_cost(e, j);
}

/* vector/sparse.rs:82-106 */
static double sparse_inner(const svec *row, const svec *col) {
    double total = 0.0; int64_t i = 0;
    for (int64_t k = 0; k < col->n; k++) {
        int64_t index = col->d[k].idx;
        while (i < row->n && row->d[i].idx < index) i++;
        if (i < row->n && row->d[i].idx == index) { total += row->d[i].val * col->d[k].val; i++; }
    }
    return total;
}

/* FTRAN, basis_inverse_rows.rs:144-173: alpha_i = row_i . a_q, zeros dropped */
static void generate_column(oracle_engine_t *e, int32_t j, svec *alpha) {
    svec *col = &e->scratch_col;
    kind_original_column(e, j, col);
    sv_clear(alpha);
    if (e->lu) { lu_generate_column(e->lu, col, alpha, &e->spike); return; }     /* lower_upper/mod.rs:157-190 */
    for (int32_t i = 0; i < e->m; i++) {
        double v = sparse_inner(&e->rows[i], col);
        if (v != 0.0) sv_push(alpha, i, v);
    }
}

/* RATIO TEST, tableau/mod.rs:221-247 (two-pass statement, identical for zero tolerances) */
static int32_t select_primal_pivot_row(const oracle_engine_t *e, const svec *alpha) {
    const oracle_config_t *c = &e->cfg;
    double min_ratio = INFINITY; int any = 0;
    double tol_pivot = c->tol_pivot;
    if (getenv("ORACLE_PIVOT_REL")) { double mx = 0.0; for (int64_t k = 0; k < alpha->n; k++) mx = fmax(mx, fabs(alpha->d[k].val)); tol_pivot *= mx; }
    for (int64_t k = 0; k < alpha->n; k++) {
        double x = alpha->d[k].val;
        if (x > tol_pivot) {
            double bi = e->b[alpha->d[k].idx];
            if (bi <= c->tol_zero) bi = 0.0;      /* also clamps a b_i that rounding pushed below 0 */
            double ratio = bi / x;
            if (!any || ratio < min_ratio) { min_ratio = ratio; any = 1; }
        }
    }
    if (!any) return -1;
    double bound = min_ratio + c->tol_tie * fmax(1.0, fabs(min_ratio));
    int32_t best_row = -1, best_leaving = 0;
    uint32_t best_size = 0;     /* ratio_rule 1 (an f64 extension, not the reference's rule): largest pivot first */
    for (int64_t k = 0; k < alpha->n; k++) {
        double x = alpha->d[k].val;
        if (x > tol_pivot) {
            int32_t row = (int32_t)alpha->d[k].idx;
            double bi = e->b[row];
            if (bi <= c->tol_zero) bi = 0.0;      /* also clamps a b_i that rounding pushed below 0 */
            double ratio = bi / x;
            if (ratio <= bound) {
                int32_t leaving = e->basis_indices[row];
                /* the pivot's size at float precision, larger = smaller key (relp_device_common.h: tie_key) */
                uint32_t size = 0;
                if (c->ratio_rule) { float f = (float)x; memcpy(&size, &f, 4); size = 0x7fffffffu - size; }
                if (best_row < 0 || size < best_size || (size == best_size && leaving < best_leaving)) {
                    best_row = row; best_leaving = leaving; best_size = size;
                }
            }
        }
    }
    return best_row;
}

/* vector/sparse.rs:213-248: target += multiple * other, into a fresh allocation */
static void add_multiple_of_row(svec *target, double multiple, const svec *other) {
    svec nw; sv_init(&nw); sv_reserve(&nw, target->n + other->n);
    int64_t j = 0;
    for (int64_t k = 0; k < target->n; k++) {
        int64_t i = target->d[k].idx; double value = target->d[k].val;
        while (j < other->n && other->d[j].idx < i) { sv_push(&nw, other->d[j].idx, multiple * other->d[j].val); j++; }
        if (j < other->n && other->d[j].idx == i) {
            double nv = value + multiple * other->d[j].val;
            if (nv != 0.0) sv_push(&nw, i, nv);
            j++;
        } else {
            sv_push(&nw, i, value);
        }
    }
    for (; j < other->n; j++) sv_push(&nw, other->d[j].idx, multiple * other->d[j].val);
    free(target->d);
    *target = nw;
}

/* carry/mod.rs:549-570; returns the leaving column, or -1 on a zero pivot */
static int32_t change_basis(oracle_engine_t *e, int32_t r, int32_t q, const svec *alpha, double relative_cost_q) {
    const double *pv = sv_get(alpha, r);
    if (!pv) return -1;                                 /* "Pivot value can't be zero." carry/mod.rs:291 */
    double pivot_value = *pv;
    /* update_b, carry/mod.rs:283-313 */
    e->b[r] /= pivot_value;
    double br = e->b[r];
    for (int64_t k = 0; k < alpha->n; k++) {
        int64_t i = alpha->d[k].idx;
        if (i != r) e->b[i] -= alpha->d[k].val * br;
    }
    if (e->lu) {
        /* LUDecomposition::change_basis (lower_upper/mod.rs:92-155), then row r of the NEW inverse by BTRAN (:204-222) */
        if (!lu_change_basis(e->lu, r, &e->spike)) e->lu_failed = 1;
        svec rho; sv_init(&rho);
        lu_basis_inverse_row(e->lu, r, &rho);
        for (int64_t k = 0; k < rho.n; k++) e->minus_pi[rho.d[k].idx] -= relative_cost_q * rho.d[k].val;
        sv_free(&rho);
        e->minus_objective -= relative_cost_q * e->b[r];
        int32_t leaving_lu = e->basis_indices[r];
        e->basis_indices[r] = q;
        if (leaving_lu < e->n_flags) e->in_basis[leaving_lu] = 0;
        e->in_basis[q] = 1;
        return leaving_lu;
    }
    /* BasisInverseRows::change_basis, basis_inverse_rows.rs:131-142, 42-83 */
    svec *prow = &e->rows[r];
    int64_t out = 0;
    for (int64_t k = 0; k < prow->n; k++) {             /* element_wise_divide, sparse.rs:291-301 */
        double v = prow->d[k].val / pivot_value;
        if (v != 0.0) { prow->d[out].idx = prow->d[k].idx; prow->d[out].val = v; out++; }
    }
    prow->n = out;
    for (int64_t k = 0; k < alpha->n; k++) {
        int64_t i = alpha->d[k].idx;
        if (i != r) add_multiple_of_row(&e->rows[i], -alpha->d[k].val, prow);
    }
    /* update_minus_pi_and_obj, carry/mod.rs:326-333 (post-update row r) */
    for (int64_t k = 0; k < prow->n; k++) e->minus_pi[prow->d[k].idx] -= relative_cost_q * prow->d[k].val;
    e->minus_objective -= relative_cost_q * e->b[r];
    int32_t leaving = e->basis_indices[r];
    e->basis_indices[r] = q;
    /* tableau/mod.rs:72-84 */
    if (leaving < e->n_flags) e->in_basis[leaving] = 0;    /* a wrapped artificial has no flag */
    e->in_basis[q] = 1;
    return leaving;
}

/* ------------------------------------------------------------------------------------------ */
/* Pivot rules (strategy/pivot_rule.rs)                                                        */
/* ------------------------------------------------------------------------------------------ */
static int find_first(oracle_engine_t *e, int64_t lo, int64_t hi, int32_t *j_out, double *d_out) {
    for (int64_t j = lo; j < hi; j++) {
        if (e->in_basis[j]) continue;
        double d = relative_cost(e, (int32_t)j);
        if (d < -e->cfg.tol_cost) { *j_out = (int32_t)j; *d_out = d; return 1; }
    }
    return 0;
}

static int select_primal_pivot_column(oracle_engine_t *e, int rule, int32_t *j_out, double *d_out) {
    int32_t n = kind_nr_columns(e);
    if (rule == ORACLE_RULE_FIRST_PROFITABLE) return find_first(e, 0, n, j_out, d_out);     /* :38-57 */
    if (rule == ORACLE_RULE_FIRST_PROFITABLE_WITH_MEMORY) {                                  /* :62-93 */
        int found;
        if (e->last_selected < 0) found = find_first(e, 0, n, j_out, d_out);
        else {
            found = find_first(e, e->last_selected, n, j_out, d_out);
            if (!found) found = find_first(e, 0, e->last_selected, j_out, d_out);
        }
        e->last_selected = found ? *j_out : -1;
        return found;
    }
    /* SteepestDescent = most negative reduced cost, first index wins ties (:97-126).  Stated in two
     * passes like the ratio test: strict minimum, then the lowest index among the columns within
     * tol_tie of it (identical to the reference's one-pass loop for tol_tie = 0). */
    int any = 0; double dmin = 0.0;
    double *d = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int32_t j = 0; j < n; j++) {
        d[j] = 0.0;
        if (e->in_basis[j]) continue;
        d[j] = relative_cost(e, j);
        if (d[j] < -e->cfg.tol_cost) { if (!any || d[j] < dmin) { dmin = d[j]; any = 1; } }
    }
    if (any) {
        double bound = dmin + e->cfg.tol_tie * fmax(1.0, fabs(dmin));
        for (int32_t j = 0; j < n; j++) {
            if (e->in_basis[j]) continue;
            if (d[j] < -e->cfg.tol_cost && d[j] <= bound) { *j_out = j; *d_out = d[j]; break; }
        }
    }
    free(d);
    return any;
}

/* ------------------------------------------------------------------------------------------ */
/* Construction                                                                                */
/* ------------------------------------------------------------------------------------------ */
static void *dup_mem(const void *p, size_t bytes) {
    void *q = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(q, p, bytes);
    return q;
}

oracle_engine_t *oracle_create(const oracle_matrix_data_t *md, const oracle_config_t *cfg) {
    oracle_engine_t *e = (oracle_engine_t *)calloc(1, sizeof(*e));
    e->nr_normal = md->nr_normal; e->nr_eq = md->nr_eq; e->nr_range = md->nr_range;
    e->nr_le = md->nr_le; e->nr_ge = md->nr_ge;
    int32_t mc = md_nr_constraints(e);
    int64_t nnz = md->col_ptr[md->nr_normal];
    e->col_ptr = (int64_t *)dup_mem(md->col_ptr, sizeof(int64_t) * (size_t)(md->nr_normal + 1));
    e->row_idx = (int32_t *)dup_mem(md->row_idx, sizeof(int32_t) * (size_t)nnz);
    e->values = (double *)dup_mem(md->values, sizeof(double) * (size_t)nnz);
    e->b0 = (double *)dup_mem(md->b, sizeof(double) * (size_t)mc);
    e->ranges = (double *)dup_mem(md->ranges, sizeof(double) * (size_t)md->nr_range);
    e->cost = (double *)dup_mem(md->cost, sizeof(double) * (size_t)md->nr_normal);
    e->upper = (double *)dup_mem(md->upper_bound, sizeof(double) * (size_t)md->nr_normal);
    e->cfg = *cfg;
    e->var_to_bound = (int32_t *)malloc(sizeof(int32_t) * (size_t)(md->nr_normal + 1));
    e->bound_to_var = (int32_t *)malloc(sizeof(int32_t) * (size_t)(md->nr_normal + 1));
    e->nr_bounds = 0;
    for (int32_t j = 0; j < md->nr_normal; j++) {
        if (isfinite(e->upper[j])) { e->var_to_bound[j] = e->nr_bounds; e->bound_to_var[e->nr_bounds++] = j; }
        else e->var_to_bound[j] = -1;
    }
    int32_t camt[6] = { e->nr_normal, e->nr_range, e->nr_le, e->nr_ge, e->nr_bounds, e->nr_range };
    int32_t ramt[6] = { e->nr_eq, e->nr_range, e->nr_le, e->nr_ge, e->nr_bounds, e->nr_range };
    e->col_start[0] = e->row_start[0] = 0;
    for (int g = 0; g < 6; g++) { e->col_start[g + 1] = e->col_start[g] + camt[g]; e->row_start[g + 1] = e->row_start[g] + ramt[g]; }
    e->nr_filtered = 0; e->filtered = NULL;
    e->nr_zero_level = 0;
    e->nr_exchanges = 0;
    sv_init(&e->scratch_col); sv_init(&e->scratch_alpha);

    /* Tableau::<_, Partially<_>>::new, partially.rs:125-206 */
    int32_t m = md_nr_rows_full(e);
    e->m = m;
    int32_t nr_real = e->nr_le + e->nr_bounds + e->nr_range;          /* matrix_data.rs:432-452 */
    int32_t *real_row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr_real + 1));
    int32_t *real_col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr_real + 1));
    int32_t t = 0;
    for (int32_t j = 0; j < e->nr_le; j++, t++) { real_row[t] = e->row_start[2] + j; real_col[t] = e->col_start[2] + j; }
    for (int32_t j = 0; j < e->nr_bounds; j++, t++) { real_row[t] = e->row_start[4] + j; real_col[t] = e->col_start[4] + j; }
    for (int32_t j = 0; j < e->nr_range; j++, t++) { real_row[t] = e->row_start[5] + j; real_col[t] = e->col_start[5] + j; }
    int32_t na = m - nr_real;
    e->nr_artificial = na;
    e->column_to_row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(na + 1));
    int32_t i = 0;
    for (int32_t ith = 0; ith < na; ith++) {
        while (i < nr_real && ith + i == real_row[i]) i++;
        e->column_to_row[ith] = ith + i;
    }
    e->basis_indices = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    int32_t ac = 0;
    for (int32_t row = 0; row < m; row++) {
        int can_a = ac < na, can_r = (row - ac) < nr_real;
        if (can_a && can_r) {
            if (e->column_to_row[ac] < real_row[row - ac]) { e->basis_indices[row] = ac; ac++; }
            else e->basis_indices[row] = na + real_col[row - ac];
        } else if (can_a) { e->basis_indices[row] = ac; ac++; }
        else e->basis_indices[row] = na + real_col[row - ac];
    }
    free(real_row); free(real_col);

    /* Carry::create_for_partially_artificial, carry/mod.rs:381-426 */
    e->phase = 1;
    e->b = (double *)malloc(sizeof(double) * (size_t)m);
    for (int32_t r = 0; r < mc; r++) e->b[r] = e->b0[r];                 /* right_hand_side, matrix_data.rs:359-371 */
    for (int32_t k = 0; k < e->nr_bounds; k++) e->b[mc + k] = e->upper[e->bound_to_var[k]];
    for (int32_t k = 0; k < e->nr_range; k++) e->b[mc + e->nr_bounds + k] = e->ranges[k];
    e->minus_pi = (double *)calloc((size_t)m, sizeof(double));
    double objective = 0.0;
    for (int32_t k = 0; k < na; k++) { objective += e->b[e->column_to_row[k]]; e->minus_pi[e->column_to_row[k]] = -1.0; }
    e->minus_objective = -objective;
    e->initial_phase1_objective = objective;
    sv_init(&e->spike);
    if (e->cfg.basis_inverse == 1) {
        e->lu = lu_identity(m);                                /* lower_upper/mod.rs:66-74 */
    } else {
        e->rows = (svec *)malloc(sizeof(svec) * (size_t)m);
        for (int32_t r = 0; r < m; r++) { sv_init(&e->rows[r]); sv_push(&e->rows[r], r, 1.0); }
    }
    e->n_flags = na + md_nr_columns(e);
    e->in_basis = (uint8_t *)calloc((size_t)e->n_flags, 1);
    for (int32_t r = 0; r < m; r++) e->in_basis[e->basis_indices[r]] = 1;
    e->last_selected = -1;
    return e;
}

void oracle_destroy(oracle_engine_t *e) {
    if (!e) return;
    if (e->rows) for (int32_t r = 0; r < e->m; r++) sv_free(&e->rows[r]);
    lu_free(e->lu); sv_free(&e->spike);
    free(e->rows); free(e->in_basis); free(e->minus_pi); free(e->b); free(e->basis_indices);
    free(e->column_to_row); free(e->var_to_bound); free(e->bound_to_var);
    free(e->col_ptr); free(e->row_idx); free(e->values); free(e->b0); free(e->ranges); free(e->cost); free(e->upper);
    free(e->filtered);
    sv_free(&e->scratch_col); sv_free(&e->scratch_alpha);
    free(e);
}

/* ------------------------------------------------------------------------------------------ */
/* Phase boundary                                                                              */
/* ------------------------------------------------------------------------------------------ */
static void record(int64_t *n_done, int64_t cap, int32_t *tp, int32_t *te, int32_t *tr, int32_t *tl,
                   int32_t phase, int32_t q, int32_t r, int32_t leaving) {
    int64_t k = *n_done;
    if (k < cap) { if (tp) tp[k] = phase; if (te) te[k] = q; if (tr) tr[k] = r; if (tl) tl[k] = leaving; }
    (*n_done)++;
}

static int lu_refactor(oracle_engine_t *e);
/* phase_one.rs:223-260.  Returns number of redundant "rows" written to rows_to_remove. */
static int32_t remove_artificial_basis_variables(oracle_engine_t *e, int32_t *rows_to_remove,
                                                 int64_t *n_done, int64_t cap, int32_t *tp, int32_t *te,
                                                 int32_t *tr, int32_t *tl) {
    int32_t nrem = 0;
    int32_t n = kind_nr_columns(e);
    svec alpha; sv_init(&alpha);
    for (int32_t a = 0; a < e->nr_artificial; a++) {        /* sorted artificial indices */
        if (!e->in_basis[a]) continue;
        int32_t pivot_row = e->column_to_row[a];            /* phase_one.rs:236: the artificial's ORIGINAL row */
        if (e->cfg.artificial_removal)                      /* extension: the row it is basic in now */
            for (int32_t i = 0; i < e->m; i++) if (e->basis_indices[i] == a) { pivot_row = i; break; }
        int found = 0; int32_t q = -1; double cost = 0.0;
        for (int32_t j = e->nr_artificial; j < n && !found; j++) {
            if (e->in_basis[j]) continue;
            double d = relative_cost(e, j);
            if (!e->cfg.artificial_removal && fabs(d) > e->cfg.tol_cost) continue;   /* cost.is_zero() */
            kind_original_column(e, j, &e->scratch_col);
            double el;
            if (e->lu) {                                        /* lower_upper/mod.rs:192-197: a whole FTRAN */
                lu_generate_column(e->lu, &e->scratch_col, &alpha, &e->spike);
                const double *pv = sv_get(&alpha, pivot_row);
                el = pv ? *pv : 0.0;
            } else {
                el = sparse_inner(&e->rows[pivot_row], &e->scratch_col); /* generate_element */
            }
            if (fabs(el) > e->cfg.tol_pivot) { found = 1; q = j; cost = d; }
        }
        if (found) {
            generate_column(e, q, &alpha);
            int32_t leaving = change_basis(e, pivot_row, q, &alpha, cost);
            if (e->lu && leaving >= 0 && e->lu->n_updates > (e->cfg.refactor_after > 0 ? e->cfg.refactor_after : 10)) lu_refactor(e);
            record(n_done, cap, tp, te, tr, tl, 1, q, pivot_row, leaving);
            e->nr_zero_level++;
        } else {
            /* NB: the reference pushes the artificial's INDEX (phase_one.rs:252).  artificial_removal 1 (an extension)
             * removes the artificial's OWN constraint together with the basis position it sits in: the two positions
             * are exchanged first (rows of B^-1, b, basis), so that one index names both -- the pair (own row, position)
             * always leaves a basis of the filtered problem, (position, position) only if B^-1[r][r] != 0 */
            if (e->cfg.artificial_removal) {
                int32_t o = e->column_to_row[a];
                if (o != pivot_row) {
                    e->nr_exchanges++;
                    if (e->rows) { svec tr_ = e->rows[o]; e->rows[o] = e->rows[pivot_row]; e->rows[pivot_row] = tr_; }     /* (LU: re-inverted below) */
                    double tb = e->b[o]; e->b[o] = e->b[pivot_row]; e->b[pivot_row] = tb;
                    int32_t ti = e->basis_indices[o]; e->basis_indices[o] = e->basis_indices[pivot_row]; e->basis_indices[pivot_row] = ti;
                }
                rows_to_remove[nrem++] = o;
            } else {
                rows_to_remove[nrem++] = a;
            }
        }
    }
    sv_free(&alpha);
    /* ascending and distinct: switch_to_phase_two walks the list beside the rows (literal mode pushes ascending indices) */
    for (int32_t i = 1; i < nrem; i++) {
        int32_t v = rows_to_remove[i], j = i - 1;
        while (j >= 0 && rows_to_remove[j] > v) { rows_to_remove[j + 1] = rows_to_remove[j]; j--; }
        rows_to_remove[j + 1] = v;
    }
    int32_t uniq = 0;
    for (int32_t i = 0; i < nrem; i++) if (uniq == 0 || rows_to_remove[uniq - 1] != rows_to_remove[i]) rows_to_remove[uniq++] = rows_to_remove[i];
    return uniq;
}

/* BI::invert(basis columns) for the LU back-end (carry/mod.rs:602-614, 512-547): returns 0 when a basis column cannot be
 * generated (a wrapped artificial index: the reference's release build would index out of range there) or B is singular */
static int lu_refactor(oracle_engine_t *e) {
    int32_t m = e->m, n = kind_nr_columns(e);
    svec *cols = (svec *)calloc((size_t)m, sizeof(svec));
    int ok = 1;
    for (int32_t i = 0; i < m && ok; i++) {
        if (e->basis_indices[i] >= n) {
            /* an artificial variable that survived phase 1 with a wrapped index (switch_to_phase_two): the reference's release
             * build would index its provider out of range here; the column is still the unit column of the artificial's row,
             * which is what the GPU engines factorise (relp_engine_lu.cpp: lu_basis_columns) */
            int64_t a = (int64_t)e->wrapped_na - 1 - ((int64_t)INT32_MAX - e->basis_indices[i]);
            if (e->phase != 2 || a < 0 || a >= e->wrapped_na) { ok = 0; break; }
            sv_clear(&e->scratch_col); sv_push(&e->scratch_col, e->column_to_row[a], 1.0);
            filter_column(e, &e->scratch_col);
            if (e->scratch_col.n != 1) { ok = 0; break; }
            lsv_copy(&cols[i], &e->scratch_col);
            continue;
        }
        kind_original_column(e, e->basis_indices[i], &e->scratch_col);
        lsv_copy(&cols[i], &e->scratch_col);
    }
    lu_t *f = ok ? lu_invert_thr(m, cols, e->cfg.lu_threshold) : NULL;
    for (int32_t i = 0; i < m; i++) sv_free(&cols[i]);
    free(cols);
    if (!f) { e->lu_failed = 1; return 0; }
    lu_free(e->lu);
    e->lu = f;
    e->lu_refactorisations++;
    return 1;
}
/* the rows of B^-1 through the LU back-end the way create_minus_pi_from_artificial builds them (carry/mod.rs:226-236): one unit
 * FTRAN per column j, entries (i, v) appended to row i, i.e. every row sorted by j */
static svec *lu_inverse_rows(oracle_engine_t *e) {
    int32_t m = e->m;
    svec *rows = (svec *)calloc((size_t)m, sizeof(svec));
    svec unit, column, spike; sv_init(&unit); sv_init(&column); sv_init(&spike);
    for (int32_t j = 0; j < m; j++) {
        sv_clear(&unit); sv_push(&unit, j, 1.0);
        lu_generate_column(e->lu, &unit, &column, &spike);
        for (int64_t k = 0; k < column.n; k++) sv_push(&rows[column.d[k].idx], j, column.d[k].val);
    }
    sv_free(&unit); sv_free(&column); sv_free(&spike);
    return rows;
}

/* non_artificial.rs:151-220 + carry/mod.rs:484-510, 650-689 */
static void switch_to_phase_two(oracle_engine_t *e, const int32_t *rows_to_remove, int32_t nrem) {
    int32_t na = e->nr_artificial;
    int32_t m_old = e->m;
    if (nrem > 0) {
        /* from_artificial_removing_rows: drop rows/cols from B^-1, b, basis (RemoveBasisPart) */
        e->nr_filtered = nrem;
        e->filtered = (int32_t *)dup_mem(rows_to_remove, sizeof(int32_t) * (size_t)nrem);
        int32_t out = 0, f = 0;
        for (int32_t r = 0; r < m_old; r++) {
            if (f < nrem && rows_to_remove[f] == r) { f++; if (e->rows) sv_free(&e->rows[r]); continue; }
            if (e->rows) e->rows[out] = e->rows[r];
            e->b[out] = e->b[r]; e->basis_indices[out] = e->basis_indices[r]; out++;
        }
        e->m = out;
        for (int32_t r = 0; e->rows && r < e->m; r++) {     /* remove_sparse_indices on every row */
            svec *row = &e->rows[r]; int64_t o = 0; int32_t skipped = 0;
            for (int64_t k = 0; k < row->n; k++) {
                int64_t i = row->d[k].idx;
                while (skipped < nrem && rows_to_remove[skipped] < i) skipped++;
                if (skipped < nrem && rows_to_remove[skipped] == i) continue;
                row->d[o].idx = i - skipped; row->d[o].val = row->d[k].val; o++;
            }
            row->n = o;
        }
    }
    /* An artificial variable can survive remove_artificial_basis_variables: when it re-entered the basis in
     * another row than its own, the zero-level pivot of phase_one.rs:236 is made in its ORIGINAL row and
     * removes somebody else.  The reference's integration tests are release builds, where
     * `basis_column -= nr_artificial` (carry/mod.rs:524, 663) then wraps to a huge usize: a column without
     * cost that sorts last in Bland's tie-break and stays basic at value zero.  Same here, with the wrapped
     * index squeezed into int32: artificial a becomes INT32_MAX - (na - 1 - a) (order preserved). */
    for (int32_t r = 0; r < e->m; r++) {
        if (e->basis_indices[r] < na) e->basis_indices[r] = INT32_MAX - (na - 1 - e->basis_indices[r]);
        else e->basis_indices[r] -= na;
    }
    e->phase = 2;
    e->wrapped_na = na;
    e->nr_artificial = 0;
    int32_t n2 = md_nr_columns(e);
    memset(e->in_basis, 0, (size_t)e->n_flags);
    for (int32_t r = 0; r < e->m; r++) if (e->basis_indices[r] < e->n_flags) e->in_basis[e->basis_indices[r]] = 1;
    (void)n2;
    /* create_minus_pi_from_artificial, carry/mod.rs:214-248.  The reference builds every column of
     * B^-1 by a unit FTRAN and re-assembles rows; entry (i,j) of that is exactly rows[i][j], and the
     * accumulation order (i ascending, then j ascending) is kept. */
    svec *inv_rows = e->rows;
    if (e->lu) {
        /* the generic from_artificial_remove_rows (carry/mod.rs:512-547) re-inverts from the filtered columns; without removed
         * rows the factors are kept as they are (from_artificial, :484-510) */
        if (nrem > 0 && !lu_refactor(e)) return;
        inv_rows = lu_inverse_rows(e);
    }
    for (int32_t j = 0; j < e->m; j++) e->minus_pi[j] = 0.0;
    for (int32_t i = 0; i < e->m; i++) {
        if (e->basis_indices[i] >= e->nr_normal) continue;      /* cost None (slacks, wrapped artificials) */
        double c = md_cost_value(e, e->basis_indices[i]);
        for (int64_t k = 0; k < inv_rows[i].n; k++) e->minus_pi[inv_rows[i].d[k].idx] += inv_rows[i].d[k].val * c;
    }
    if (e->lu) { for (int32_t i = 0; i < e->m; i++) sv_free(&inv_rows[i]); free(inv_rows); }
    for (int32_t j = 0; j < e->m; j++) e->minus_pi[j] = -e->minus_pi[j];
    /* create_minus_obj_from_artificial, carry/mod.rs:258-271 */
    double objective = 0.0;
    for (int32_t r = 0; r < e->m; r++) if (e->basis_indices[r] < e->nr_normal) objective += e->b[r] * md_cost_value(e, e->basis_indices[r]);
    e->minus_objective = -objective;
    e->last_selected = -1;
}

/* ------------------------------------------------------------------------------------------ */
/* Driver loops (phase_one.rs:125-170, phase_two.rs:22-51)                                     */
/* ------------------------------------------------------------------------------------------ */
int oracle_run(oracle_engine_t *e, int64_t max_iters, int through_phases,
               int32_t *tp, int32_t *te, int32_t *tr, int32_t *tl, int64_t cap, int64_t *n_done_out) {
    int64_t n_done = 0;
    int status = ORACLE_ITERATION_LIMIT;
    svec *alpha = &e->scratch_alpha;
    while (n_done < max_iters) {
        int rule = e->phase == 1 ? e->cfg.phase_one_rule : e->cfg.phase_two_rule;
        int32_t q; double dq;
        if (!select_primal_pivot_column(e, rule, &q, &dq)) {
            if (e->phase == 2) { status = ORACLE_OPTIMAL; break; }
            double obj = -e->minus_objective;
            if (fabs(obj) > e->cfg.tol_feas * fmax(1.0, e->initial_phase1_objective)) { status = ORACLE_INFEASIBLE; break; }
            int32_t *rows_to_remove = (int32_t *)malloc(sizeof(int32_t) * (size_t)(e->nr_artificial + 1));
            int32_t nrem = remove_artificial_basis_variables(e, rows_to_remove, &n_done, cap, tp, te, tr, tl);
            switch_to_phase_two(e, rows_to_remove, nrem);
            free(rows_to_remove);
            if (!through_phases) { status = ORACLE_PHASE_ONE_DONE; break; }
            continue;
        }
        generate_column(e, q, alpha);
        int32_t r = select_primal_pivot_row(e, alpha);
        if (r < 0) {
            if (getenv("ORACLE_DEBUG")) { double mx = -1e300; for (int64_t k = 0; k < alpha->n; k++) mx = fmax(mx, alpha->d[k].val); fprintf(stderr, "[oracle] no row: q %d d %.17g alpha nnz %lld max %.17g\n", q, dq, (long long)alpha->n, mx); }
            status = e->phase == 2 ? ORACLE_UNBOUNDED : ORACLE_ERROR; break; }
        int32_t leaving = change_basis(e, r, q, alpha, dq);
        /* after_basis_change, carry/mod.rs:602-614: should_refactor() = updates.len() > 10 (lower_upper/mod.rs:199-202) */
        if (e->lu && leaving >= 0 && e->lu->n_updates > (e->cfg.refactor_after > 0 ? e->cfg.refactor_after : 10)) lu_refactor(e);
        if (e->lu_failed) { status = ORACLE_ERROR; break; }
        if (leaving < 0) { if (getenv("ORACLE_DEBUG")) fprintf(stderr, "[oracle] zero pivot: q %d r %d\n", q, r); status = ORACLE_ERROR; break; }
        record(&n_done, cap, tp, te, tr, tl, e->phase, q, r, leaving);
    }
    if (n_done_out) *n_done_out = n_done;
    return status;
}

int32_t oracle_m(const oracle_engine_t *e) { return e->m; }
int32_t oracle_n(const oracle_engine_t *e) { return kind_nr_columns(e); }
int32_t oracle_phase(const oracle_engine_t *e) { return e->phase; }
int32_t oracle_nr_artificial(const oracle_engine_t *e) { return e->nr_artificial; }
int32_t oracle_nr_filtered_rows(const oracle_engine_t *e) { return e->nr_filtered; }
int32_t oracle_nr_zero_level_pivots(const oracle_engine_t *e) { return e->nr_zero_level; }
int32_t oracle_nr_position_exchanges(const oracle_engine_t *e) { return e->nr_exchanges; }
void oracle_get_filtered_rows(const oracle_engine_t *e, int32_t *out) { for (int32_t k = 0; k < e->nr_filtered; k++) out[k] = e->filtered[k]; }
double  oracle_objective(const oracle_engine_t *e) { return -e->minus_objective; }
void oracle_get_b(const oracle_engine_t *e, double *out) { memcpy(out, e->b, sizeof(double) * (size_t)e->m); }
void oracle_get_minus_pi(const oracle_engine_t *e, double *out) { memcpy(out, e->minus_pi, sizeof(double) * (size_t)e->m); }
void oracle_get_basis(const oracle_engine_t *e, int32_t *out) { memcpy(out, e->basis_indices, sizeof(int32_t) * (size_t)e->m); }
void oracle_get_basis_inverse(const oracle_engine_t *e, double *out) {
    memset(out, 0, sizeof(double) * (size_t)e->m * (size_t)e->m);
    if (e->lu) {
        svec *rows = lu_inverse_rows((oracle_engine_t *)e);
        for (int32_t i = 0; i < e->m; i++) { for (int64_t k = 0; k < rows[i].n; k++) out[(size_t)i * e->m + rows[i].d[k].idx] = rows[i].d[k].val; sv_free(&rows[i]); }
        free(rows);
        return;
    }
    for (int32_t i = 0; i < e->m; i++)
        for (int64_t k = 0; k < e->rows[i].n; k++) out[(size_t)i * e->m + e->rows[i].d[k].idx] = e->rows[i].d[k].val;
}
int64_t oracle_basis_inverse_nnz(const oracle_engine_t *e) {
    int64_t t = 0;
    if (e->lu) { for (int32_t j = 0; j < e->m; j++) t += e->lu->lower[j].n + e->lu->upper[j].n; return t; }
    for (int32_t i = 0; i < e->m; i++) t += e->rows[i].n;
    return t;
}
void oracle_lu_stats(const oracle_engine_t *e, int64_t *out4) {
    out4[0] = e->lu_refactorisations; out4[1] = e->lu ? e->lu->n_updates : 0; out4[2] = 0; out4[3] = 0;
    if (e->lu) for (int32_t j = 0; j < e->m; j++) { out4[2] += e->lu->lower[j].n; out4[3] += e->lu->upper[j].n; }
}

/* ---- the LU back-end by itself (tests: the reference's known answers of lower_upper/mod.rs:488-867, decomposition/mod.rs:301-491) ---- */
struct oracle_lu { lu_t *f; svec column, spike; };
static void olu_columns(int32_t m, const int64_t *ptr, const int32_t *idx, const double *val, svec *cols) {
    for (int32_t j = 0; j < m; j++) { sv_init(&cols[j]); for (int64_t k = ptr[j]; k < ptr[j + 1]; k++) sv_push(&cols[j], idx[k], val[k]); }
}
/* `LUDecomposition { lower_triangular, upper_triangular, .. }` literally, P = Q = I: L column-major without the unit diagonal,
 * U column-major with the diagonal among the entries of each column (stored last: the largest row index) */
oracle_lu_t *oracle_lu_from_triangles(int32_t m, const int64_t *lptr, const int32_t *lidx, const double *lval,
                                      const int64_t *uptr, const int32_t *uidx, const double *uval) {
    oracle_lu_t *h = (oracle_lu_t *)calloc(1, sizeof(*h));
    h->f = lu_alloc(m);
    for (int32_t i = 0; i < m; i++) { h->f->rp_fwd[i] = i; h->f->cp_fwd[i] = i; h->f->cp_bwd[i] = i; }
    olu_columns(m, lptr, lidx, lval, h->f->lower);
    olu_columns(m, uptr, uidx, uval, h->f->upper);
    for (int32_t j = 0; j < m; j++) { qsort(h->f->lower[j].d, (size_t)h->f->lower[j].n, sizeof(tup), lsv_cmp); qsort(h->f->upper[j].d, (size_t)h->f->upper[j].n, sizeof(tup), lsv_cmp); }
    sv_init(&h->column); sv_init(&h->spike);
    return h;
}
/* LUDecomposition::invert(columns) (lower_upper/mod.rs:76-90); NULL when singular */
oracle_lu_t *oracle_lu_invert(int32_t m, const int64_t *cptr, const int32_t *cidx, const double *cval) {
    svec *cols = (svec *)calloc((size_t)m, sizeof(svec));
    olu_columns(m, cptr, cidx, cval, cols);
    lu_t *f = lu_invert(m, cols);
    for (int32_t j = 0; j < m; j++) sv_free(&cols[j]);
    free(cols);
    if (!f) return NULL;
    oracle_lu_t *h = (oracle_lu_t *)calloc(1, sizeof(*h));
    h->f = f; sv_init(&h->column); sv_init(&h->spike);
    return h;
}
void oracle_lu_destroy(oracle_lu_t *h) { if (!h) return; lu_free(h->f); sv_free(&h->column); sv_free(&h->spike); free(h); }
/* generate_column (FTRAN): dense result by basis position; the spike is kept for the next oracle_lu_change_basis */
void oracle_lu_generate_column(oracle_lu_t *h, int32_t n, const int32_t *idx, const double *val, double *out_m) {
    svec col; sv_init(&col);
    for (int32_t k = 0; k < n; k++) sv_push(&col, idx[k], val[k]);
    lu_generate_column(h->f, &col, &h->column, &h->spike);
    memset(out_m, 0, sizeof(double) * (size_t)h->f->m);
    for (int64_t k = 0; k < h->column.n; k++) out_m[h->column.d[k].idx] = h->column.d[k].val;
    sv_free(&col);
}
int oracle_lu_change_basis(oracle_lu_t *h, int32_t pivot_row_index) { return lu_change_basis(h->f, pivot_row_index, &h->spike); }
void oracle_lu_basis_inverse_row(oracle_lu_t *h, int32_t row, double *out_m) {
    svec r; sv_init(&r);
    lu_basis_inverse_row(h->f, row, &r);
    memset(out_m, 0, sizeof(double) * (size_t)h->f->m);
    for (int64_t k = 0; k < r.n; k++) out_m[r.d[k].idx] = r.d[k].val;
    sv_free(&r);
}
int32_t oracle_lu_nr_updates(const oracle_lu_t *h) { return h->f->n_updates; }
/* update k: the eta's pivot (= the RotateToBack index) and its values as a dense vector of m */
void oracle_lu_get_update(const oracle_lu_t *h, int32_t k, int32_t *pivot, double *values_m) {
    *pivot = h->f->etas[k].pivot;
    memset(values_m, 0, sizeof(double) * (size_t)h->f->m);
    for (int64_t t = 0; t < h->f->etas[k].values.n; t++) values_m[h->f->etas[k].values.d[t].idx] = h->f->etas[k].values.d[t].val;
}
/* which = 0: lower, 1: upper; dense m x m, row-major (entry [i][j] = column j, row i); permutations by forward index */
void oracle_lu_get_factor(const oracle_lu_t *h, int which, double *out_mm, int32_t *row_forward, int32_t *column_forward) {
    const int32_t m = h->f->m;
    memset(out_mm, 0, sizeof(double) * (size_t)m * (size_t)m);
    const svec *cols = which ? h->f->upper : h->f->lower;
    for (int32_t j = 0; j < m; j++) for (int64_t k = 0; k < cols[j].n; k++) out_mm[(size_t)cols[j].d[k].idx * m + j] = cols[j].d[k].val;
    if (row_forward) memcpy(row_forward, h->f->rp_fwd, sizeof(int32_t) * (size_t)m);
    if (column_forward) memcpy(column_forward, h->f->cp_fwd, sizeof(int32_t) * (size_t)m);
}
