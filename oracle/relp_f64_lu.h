/*
 * relp_f64_lu.h -- the reference's SECOND basis-inverse back-end in f64, for the CPU oracle (included by relp_f64.c; TEST
 * INFRASTRUCTURE ONLY, see relp_oracle.h): `LUDecomposition` with the Forrest-Tomlin-style update file,
 *   carry/lower_upper/mod.rs:35-374      (struct, invert, change_basis, generate_column, basis_inverse_row, the four solves)
 *   carry/lower_upper/eta_file.rs:14-156 (EtaFile: apply_left, apply_right, update_spike_pivot_value)
 *   carry/lower_upper/permutation/rotate_to_back.rs:15-110
 *   carry/lower_upper/decomposition/mod.rs:27-268, decomposition/pivoting.rs:45-81 (right-looking LU, Markowitz)
 * This is what `src/bin/main.rs:52` runs (`Carry<_, LUDecomposition<_>>`), refactorised from the basis columns whenever more
 * than 10 updates are pending (mod.rs:199-202).  Transcribed from oracle/relp_exact.py (the exact restatement pinned by the
 * reference's own LU known answers) with `double` for the field: same data structures -- column-major L without its unit
 * diagonal, column-major U with the diagonal stored last, sorted sparse vectors, an ordered work map for the solves -- the same
 * loops, including the scans over ALL later / earlier columns with a binary search each in the two left-hand solves
 * (mod.rs:314-320, 341-347: O(m log) per entry of the work vector), because this is the CPU baseline of the sparse path.
 * The factorisation keeps rows / columns in place and tracks positions instead of swapping (decomposition/mod.rs:219-268
 * moves data); the pivot chosen is the same: minimum (r - 1)(c - 1), ties to the lower permuted column, then the lower permuted
 * row (pivoting.rs:75-80: stable sort by column, first minimum).
 * Two f64 readings, both off in the exact sense of the word "literal" only: a cancellation residue below 1e-11 of its operands is an
 * exact zero (see lu_invert), and `lu_threshold` > 0 (oracle_config_t, default 0 = the reference) restricts the Markowitz search
 * to entries of at least that fraction of their column's largest active entry -- the threshold pivoting every f64 sparse LU has
 * (the GPU engine: 0.1); without it the literal search ends Netlib 25FV47 `infeasible` after 7,465 pivots on a corrupted inverse.
 */
#ifndef RELP_F64_LU_H
#define RELP_F64_LU_H

typedef struct { svec values; int32_t pivot; } lu_eta;            /* eta_file.rs:14-18 (len = m) */
typedef struct {
    int32_t m;
    int32_t *rp_fwd, *cp_fwd, *cp_bwd;      /* row / column permutation (full.rs): forward[original] = permuted */
    svec *lower, *upper;                    /* m columns each */
    lu_eta *etas; int32_t *rot;             /* updates: (EtaFile, RotateToBack(rot, m)) */
    int32_t n_updates, cap_updates;
    /* the ordered work map of the solves (BTreeMap<usize, F>, mod.rs:236-374) */
    double *wval; uint8_t *whas; int32_t *heap; int32_t heap_n, heap_cap; int wsign;
} lu_t;

/* ---- sorted sparse vector helpers (vector/sparse.rs) ---- */
static int64_t lsv_find(const svec *v, int64_t idx, int *found) {        /* binary search: position, or insertion point */
    int64_t lo = 0, hi = v->n;
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (v->d[mid].idx < idx) lo = mid + 1; else hi = mid; }
    *found = lo < v->n && v->d[lo].idx == idx;
    return lo;
}
static void lsv_insert(svec *v, int64_t pos, int64_t idx, double val) {
    sv_reserve(v, v->n + 1);
    memmove(v->d + pos + 1, v->d + pos, (size_t)(v->n - pos) * sizeof(tup));
    v->d[pos].idx = idx; v->d[pos].val = val; v->n++;
}
static void lsv_remove(svec *v, int64_t pos) {
    memmove(v->d + pos, v->d + pos + 1, (size_t)(v->n - pos - 1) * sizeof(tup));
    v->n--;
}
static void lsv_copy(svec *dst, const svec *src) {
    sv_clear(dst); sv_reserve(dst, src->n);
    if (src->n) memcpy(dst->d, src->d, (size_t)src->n * sizeof(tup));
    dst->n = src->n;
}
static int lsv_cmp(const void *a, const void *b) { int64_t x = ((const tup *)a)->idx, y = ((const tup *)b)->idx; return x < y ? -1 : x > y; }

/* eta_file.rs:136-156 */
static void lu_update_value(double difference, int found, int64_t pos, int64_t new_index, svec *vector) {
    if (difference != 0.0) {
        if (found) {
            double nv = vector->d[pos].val - difference;
            if (nv == 0.0) lsv_remove(vector, pos); else vector->d[pos].val = nv;
        } else {
            lsv_insert(vector, pos, new_index, -difference);
        }
    }
}
/* eta_file.rs:49-65: x := x R */
static void lu_eta_apply_left(const lu_eta *eta, svec *vector) {
    int found; int64_t pivot_pos = lsv_find(vector, eta->pivot, &found);
    if (!found) return;
    for (int64_t k = 0; k < eta->values.n; k++) {
        int has; int64_t pos = lsv_find(vector, eta->values.d[k].idx, &has);
        double difference = eta->values.d[k].val * vector->d[pivot_pos].val;     /* (the pivot lies left of every j) */
        lu_update_value(difference, has, pos, eta->values.d[k].idx, vector);
    }
}
/* eta_file.rs:72-104: x := R x */
static void lu_eta_apply_right(const lu_eta *eta, svec *vector) {
    int found; int64_t pivot_pos = lsv_find(vector, eta->pivot, &found);
    double total = 0.0;
    int64_t e = 0, v = pivot_pos;
    while (e < eta->values.n && v < vector->n) {
        int64_t ei = eta->values.d[e].idx, vi = vector->d[v].idx;
        if (ei < vi) e++;
        else if (ei == vi) { total += eta->values.d[e].val * vector->d[v].val; e++; v++; }
        else v++;
    }
    lu_update_value(total, found, pivot_pos, eta->pivot, vector);
}
/* eta_file.rs:111-133 */
static void lu_eta_update_spike_pivot_value(const lu_eta *eta, svec *spike) {
    int found; int64_t pos = lsv_find(spike, eta->pivot, &found);
    int64_t search = found ? pos + 1 : pos;
    double difference = 0.0;
    for (int64_t k = 0; k < eta->values.n; k++) {
        int64_t lo = search, hi = spike->n, j = eta->values.d[k].idx;
        while (lo < hi) { int64_t mid = (lo + hi) / 2; if (spike->d[mid].idx < j) lo = mid + 1; else hi = mid; }
        if (lo < spike->n && spike->d[lo].idx == j) difference += eta->values.d[k].val * spike->d[lo].val;
    }
    lu_update_value(difference, found, pos, eta->pivot, spike);
}

/* ---- RotateToBack(index, len): rotate_to_back.rs:15-110 ---- */
static int64_t lu_rot_forward(int32_t index, int32_t len, int64_t i) { return i < index ? i : (i == index ? len - 1 : i - 1); }
static int64_t lu_rot_backward(int32_t index, int32_t len, int64_t i) { return i < index ? i : (i < len - 1 ? i + 1 : index); }
static void lu_rot_forward_sorted(int32_t index, int32_t len, svec *items) {
    int found; int64_t pos = lsv_find(items, index, &found);
    if (found) {
        double moved = items->d[pos].val;
        for (int64_t k = pos + 1; k < items->n; k++) { items->d[k - 1].idx = items->d[k].idx - 1; items->d[k - 1].val = items->d[k].val; }
        items->d[items->n - 1].idx = len - 1; items->d[items->n - 1].val = moved;
    } else {
        for (int64_t k = pos; k < items->n; k++) items->d[k].idx -= 1;
    }
}
static void lu_rot_backward_sorted(int32_t index, int32_t len, svec *items) {
    if (items->n == 0) return;
    int found; int64_t pos = lsv_find(items, index, &found);
    if (pos == items->n) return;
    if (items->d[items->n - 1].idx == len - 1) {
        double moved = items->d[items->n - 1].val;
        for (int64_t k = items->n - 1; k > pos; k--) { items->d[k].idx = items->d[k - 1].idx + 1; items->d[k].val = items->d[k - 1].val; }
        items->d[pos].idx = index; items->d[pos].val = moved;
    } else {
        for (int64_t k = pos; k < items->n; k++) items->d[k].idx += 1;
    }
}

/* ---- the ordered work map (BTreeMap stand-in: dense values + presence flags + a lazily-cleaned binary heap) ---- */
static void lu_heap_push(lu_t *f, int32_t key) {
    if (f->heap_n == f->heap_cap) { f->heap_cap *= 2; f->heap = (int32_t *)realloc(f->heap, sizeof(int32_t) * (size_t)f->heap_cap); }
    int32_t i = f->heap_n++;
    f->heap[i] = key;
    while (i > 0) { int32_t p = (i - 1) / 2; if (f->heap[p] <= f->heap[i]) break; int32_t t = f->heap[p]; f->heap[p] = f->heap[i]; f->heap[i] = t; i = p; }
}
static int32_t lu_heap_pop(lu_t *f) {
    int32_t top = f->heap[0];
    f->heap[0] = f->heap[--f->heap_n];
    int32_t i = 0;
    for (;;) {
        int32_t l = 2 * i + 1, r = l + 1, s = i;
        if (l < f->heap_n && f->heap[l] < f->heap[s]) s = l;
        if (r < f->heap_n && f->heap[r] < f->heap[s]) s = r;
        if (s == i) break;
        int32_t t = f->heap[s]; f->heap[s] = f->heap[i]; f->heap[i] = t; i = s;
    }
    return top;
}
static void lu_map_begin(lu_t *f, const svec *items, int direction) {
    f->heap_n = 0; f->wsign = direction;
    for (int64_t k = 0; k < items->n; k++) {
        int32_t idx = (int32_t)items->d[k].idx;
        f->wval[idx] = items->d[k].val; f->whas[idx] = 1;
        lu_heap_push(f, direction * idx);
    }
}
static int lu_map_pop(lu_t *f, int32_t *index, double *value) {
    while (f->heap_n > 0) {
        int32_t k = f->wsign * lu_heap_pop(f);
        if (f->whas[k]) { f->whas[k] = 0; *index = k; *value = f->wval[k]; return 1; }
    }
    return 0;
}
/* mod.rs:359-374 */
static void lu_map_insert_or_shift_maybe_remove(lu_t *f, int32_t index, double change) {
    if (!f->whas[index]) {
        f->wval[index] = -change; f->whas[index] = 1;
        /* (the heap may still hold a stale key of this index: then this push is a duplicate, which pop skips) */
        lu_heap_push(f, f->wsign * index);
    } else {
        double nv = f->wval[index] - change;
        if (nv == 0.0) f->whas[index] = 0; else f->wval[index] = nv;
    }
}

/* ---- the four triangular solves ---- */
/* L y = rhs, mod.rs:236-255 */
static void lu_invert_lower_right(lu_t *f, const svec *rhs, svec *result) {
    lu_map_begin(f, rhs, +1);
    sv_clear(result);
    int32_t row; double value;
    while (lu_map_pop(f, &row, &value)) {
        if (row != f->m - 1) for (int64_t k = 0; k < f->lower[row].n; k++) lu_map_insert_or_shift_maybe_remove(f, (int32_t)f->lower[row].d[k].idx, value * f->lower[row].d[k].val);
        sv_push(result, row, value);
    }
}
/* U x = rhs, mod.rs:257-271, 292-304 */
static void lu_invert_upper_right(lu_t *f, const svec *rhs, svec *result) {
    lu_map_begin(f, rhs, -1);
    sv_clear(result);
    int32_t row; double value;
    while (lu_map_pop(f, &row, &value)) {
        const svec *column = &f->upper[row];
        double x = value / column->d[column->n - 1].val;          /* diagonal element stored last */
        for (int64_t k = 0; k + 1 < column->n; k++) lu_map_insert_or_shift_maybe_remove(f, (int32_t)column->d[k].idx, x * column->d[k].val);
        sv_push(result, row, x);
    }
    for (int64_t a = 0, b = result->n - 1; a < b; a++, b--) { tup t = result->d[a]; result->d[a] = result->d[b]; result->d[b] = t; }
}
/* y L = rhs (row vector), mod.rs:306-330 */
static void lu_invert_lower_left(lu_t *f, const svec *rhs, svec *result) {
    lu_map_begin(f, rhs, -1);
    sv_clear(result);
    int32_t column; double value;
    while (lu_map_pop(f, &column, &value)) {
        for (int32_t j = 0; j < column; j++) {
            const double *l = sv_get(&f->lower[j], column);
            if (l) lu_map_insert_or_shift_maybe_remove(f, j, value * *l);
        }
        sv_push(result, column, value);
    }
    for (int64_t a = 0, b = result->n - 1; a < b; a++, b--) { tup t = result->d[a]; result->d[a] = result->d[b]; result->d[b] = t; }
}
/* x U = rhs (row vector), mod.rs:332-356 */
static void lu_invert_upper_left(lu_t *f, const svec *rhs, svec *result) {
    lu_map_begin(f, rhs, +1);
    sv_clear(result);
    int32_t column; double value;
    while (lu_map_pop(f, &column, &value)) {
        const svec *c = &f->upper[column];
        double x = value / c->d[c->n - 1].val;
        for (int32_t j = column + 1; j < f->m; j++) {
            const double *u = sv_get(&f->upper[j], column);
            if (u) lu_map_insert_or_shift_maybe_remove(f, j, x * *u);
        }
        sv_push(result, column, x);
    }
}

/* ---- construction ---- */
static void lu_clear_updates(lu_t *f) {
    for (int32_t k = 0; k < f->n_updates; k++) sv_free(&f->etas[k].values);
    f->n_updates = 0;
}
static lu_t *lu_alloc(int32_t m) {
    lu_t *f = (lu_t *)calloc(1, sizeof(*f));
    f->m = m;
    f->rp_fwd = (int32_t *)malloc(sizeof(int32_t) * (size_t)m); f->cp_fwd = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    f->cp_bwd = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    f->lower = (svec *)calloc((size_t)m, sizeof(svec)); f->upper = (svec *)calloc((size_t)m, sizeof(svec));
    f->wval = (double *)calloc((size_t)m, sizeof(double)); f->whas = (uint8_t *)calloc((size_t)m, 1);
    f->heap_cap = 4 * m + 1024;
    f->heap = (int32_t *)malloc(sizeof(int32_t) * (size_t)f->heap_cap);
    return f;
}
static void lu_free(lu_t *f) {
    if (!f) return;
    lu_clear_updates(f);
    for (int32_t j = 0; j < f->m; j++) { sv_free(&f->lower[j]); sv_free(&f->upper[j]); }
    free(f->lower); free(f->upper); free(f->rp_fwd); free(f->cp_fwd); free(f->cp_bwd); free(f->etas); free(f->rot);
    free(f->wval); free(f->whas); free(f->heap);
    free(f);
}
/* mod.rs:66-74 */
static lu_t *lu_identity(int32_t m) {
    lu_t *f = lu_alloc(m);
    for (int32_t i = 0; i < m; i++) { f->rp_fwd[i] = i; f->cp_fwd[i] = i; f->cp_bwd[i] = i; sv_push(&f->upper[i], i, 1.0); }
    return f;
}

/* mod.rs:76-90 + decomposition/mod.rs:27-138: P B Q = L U of the given columns; returns NULL when a pivot is missing (singular) */
static lu_t *lu_invert_thr(int32_t m, const svec *columns, double threshold) {
    /* rows of the working copy as unsorted (original column, value) lists; positions instead of swaps */
    svec *rows = (svec *)calloc((size_t)m, sizeof(svec)), *lrm = (svec *)calloc((size_t)m, sizeof(svec));
    int32_t *nnz_row = (int32_t *)calloc((size_t)m, sizeof(int32_t)), *nnz_col = (int32_t *)calloc((size_t)m, sizeof(int32_t));
    int32_t *row_at = (int32_t *)malloc(sizeof(int32_t) * (size_t)m), *col_at = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    int32_t *rpos = (int32_t *)malloc(sizeof(int32_t) * (size_t)m), *cpos = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    int32_t *mark = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    double *colmax = (double *)calloc((size_t)m, sizeof(double));
    for (int32_t j = 0; j < m; j++) {
        for (int64_t k = 0; k < columns[j].n; k++) { sv_push(&rows[columns[j].d[k].idx], j, columns[j].d[k].val); nnz_col[j]++; }
        row_at[j] = j; col_at[j] = j; rpos[j] = j; cpos[j] = j; mark[j] = -1;
    }
    for (int32_t i = 0; i < m; i++) nnz_row[i] = (int32_t)rows[i].n;
    lu_t *f = lu_alloc(m);
    int ok = 1;
    for (int32_t k = 0; k < m && ok; k++) {
        /* pivoting.rs:45-81: minimum of (r - 1)(c - 1) over every remaining entry; ties: lower column, then lower row */
        int64_t best_key = -1; int32_t best_r = -1, best_c = -1;
        if (threshold > 0.0) {          /* f64 extension (0 = the reference): largest active entry per column, for the test below */
            for (int32_t p = k; p < m; p++) colmax[col_at[p]] = 0.0;
            for (int32_t p = k; p < m; p++) { const int32_t i = row_at[p]; for (int64_t t = 0; t < rows[i].n; t++) colmax[rows[i].d[t].idx] = fmax(colmax[rows[i].d[t].idx], fabs(rows[i].d[t].val)); }
        }
        for (int32_t p = k; p < m; p++) {
            const int32_t i = row_at[p];
            for (int64_t t = 0; t < rows[i].n; t++) {
                const int32_t c = (int32_t)rows[i].d[t].idx;
                if (threshold > 0.0 && fabs(rows[i].d[t].val) < threshold * colmax[c]) continue;
                const int64_t key = (int64_t)(nnz_row[i] - 1) * (int64_t)(nnz_col[c] - 1);
                if (best_r < 0 || key < best_key || (key == best_key && (cpos[c] < cpos[best_c] || (cpos[c] == cpos[best_c] && p < rpos[best_r])))) {
                    best_key = key; best_r = i; best_c = c;
                }
            }
        }
        if (best_r < 0) { ok = 0; break; }
        /* swap (pr, pc) to (k, k): decomposition/mod.rs:219-268 */
        { int32_t pr = rpos[best_r], other = row_at[k]; row_at[pr] = other; rpos[other] = pr; row_at[k] = best_r; rpos[best_r] = k; }
        { int32_t pc = cpos[best_c], other = col_at[k]; col_at[pc] = other; cpos[other] = pc; col_at[k] = best_c; cpos[best_c] = k; }
        svec *current = &rows[best_r];
        double pivot_value = 0.0;
        for (int64_t t = 0; t < current->n; t++) {
            nnz_row[best_r]--; nnz_col[current->d[t].idx]--;
            if (current->d[t].idx == best_c) pivot_value = current->d[t].val;
        }
        for (int32_t p = k + 1; p < m; p++) {
            const int32_t i = row_at[p];
            svec *row = &rows[i];
            int64_t at = -1;
            for (int64_t t = 0; t < row->n; t++) if (row->d[t].idx == best_c) { at = t; break; }
            if (at < 0) continue;
            const double ratio = row->d[at].val / pivot_value;
            row->d[at] = row->d[row->n - 1]; row->n--;
            nnz_row[i]--; nnz_col[best_c]--;
            /* subtract_multiple_of_row_from_other_row, decomposition/mod.rs:141-205 */
            for (int64_t t = 0; t < row->n; t++) mark[row->d[t].idx] = (int32_t)t;
            for (int64_t t = 0; t < current->n; t++) {
                const int32_t c = (int32_t)current->d[t].idx;
                if (c == best_c) continue;
                const double product = ratio * current->d[t].val;
                if (mark[c] >= 0) {
                    const int64_t w = mark[c];
                    /* decomposition/mod.rs:178 drops the entry when product == old value.  In f64 a cancellation that is exact
                     * over the rationals leaves a residue of a few ulps, an "entry" the next Markowitz search may pick as a
                     * pivot (Netlib SHARE1B: an inverse wrong by 6 % after one such refactorisation): residues below
                     * 1e-11 of the operands are dropped as well (the f64 reading of `==`, like tol_zero in the ratio test) */
                    const double old = row->d[w].val, nv = old - product;
                    if (fabs(nv) <= 1e-11 * fmax(fabs(old), fabs(product))) row->d[w].val = 0.0;      /* removed below */
                    else row->d[w].val = nv;
                } else {
                    sv_push(row, c, -product);
                    mark[c] = (int32_t)(row->n - 1);
                    nnz_row[i]++; nnz_col[c]++;
                }
            }
            int64_t o = 0;
            for (int64_t t = 0; t < row->n; t++) {
                mark[row->d[t].idx] = -1;
                if (row->d[t].val == 0.0) { nnz_row[i]--; nnz_col[row->d[t].idx]--; continue; }
                row->d[o++] = row->d[t];
            }
            row->n = o;
            sv_push(&lrm[i], k, ratio);
        }
    }
    if (ok) {
        /* upper: column-major by permuted position, sorted by row, diagonal last; lower likewise */
        for (int32_t p = 0; p < m; p++) {
            const int32_t i = row_at[p];
            for (int64_t t = 0; t < rows[i].n; t++) sv_push(&f->upper[cpos[rows[i].d[t].idx]], p, rows[i].d[t].val);
            for (int64_t t = 0; t < lrm[i].n; t++) sv_push(&f->lower[lrm[i].d[t].idx], p, lrm[i].d[t].val);
        }
        for (int32_t j = 0; j < m && ok; j++) if (f->upper[j].n == 0 || f->upper[j].d[f->upper[j].n - 1].idx != j) ok = 0;
        for (int32_t i = 0; i < m; i++) { f->rp_fwd[i] = rpos[i]; f->cp_fwd[i] = cpos[i]; f->cp_bwd[cpos[i]] = i; }
    }
    for (int32_t i = 0; i < m; i++) { sv_free(&rows[i]); sv_free(&lrm[i]); }
    free(rows); free(lrm); free(nnz_row); free(nnz_col); free(row_at); free(col_at); free(rpos); free(cpos); free(mark); free(colmax);
    if (!ok) { lu_free(f); return NULL; }
    return f;
}

static lu_t *lu_invert(int32_t m, const svec *columns) { return lu_invert_thr(m, columns, 0.0); }

/* FTRAN, mod.rs:157-190: column (indexed by basis position, sorted) and the spike saved for change_basis */
static void lu_generate_column(lu_t *f, const svec *original_column, svec *column, svec *spike) {
    svec rhs, w; sv_init(&rhs); sv_init(&w);
    for (int64_t k = 0; k < original_column->n; k++) sv_push(&rhs, f->rp_fwd[original_column->d[k].idx], original_column->d[k].val);
    lu_invert_lower_right(f, &rhs, &w);
    for (int32_t k = 0; k < f->n_updates; k++) { lu_eta_apply_right(&f->etas[k], &w); lu_rot_forward_sorted(f->rot[k], f->m, &w); }
    lsv_copy(spike, &w);
    lu_invert_upper_right(f, &w, column);
    for (int32_t k = f->n_updates - 1; k >= 0; k--) for (int64_t t = 0; t < column->n; t++) column->d[t].idx = lu_rot_backward(f->rot[k], f->m, column->d[t].idx);
    for (int64_t t = 0; t < column->n; t++) column->d[t].idx = f->cp_bwd[column->d[t].idx];
    qsort(column->d, (size_t)column->n, sizeof(tup), lsv_cmp);
    sv_free(&rhs); sv_free(&w);
}
/* BTRAN of a unit vector, mod.rs:204-222: row `row` of the basis inverse, indexed by original row, sorted */
static void lu_basis_inverse_row(lu_t *f, int32_t row, svec *out) {
    int64_t r = f->cp_fwd[row];
    for (int32_t k = 0; k < f->n_updates; k++) r = lu_rot_forward(f->rot[k], f->m, r);
    svec unit, w; sv_init(&unit); sv_init(&w);
    sv_push(&unit, r, 1.0);
    lu_invert_upper_left(f, &unit, &w);
    for (int32_t k = f->n_updates - 1; k >= 0; k--) { lu_rot_backward_sorted(f->rot[k], f->m, &w); lu_eta_apply_left(&f->etas[k], &w); }
    lu_invert_lower_left(f, &w, out);
    /* row_permutation.backward_sorted: permuted -> original row, sorted again */
    int32_t *bwd = (int32_t *)malloc(sizeof(int32_t) * (size_t)f->m);
    for (int32_t i = 0; i < f->m; i++) bwd[f->rp_fwd[i]] = i;
    for (int64_t t = 0; t < out->n; t++) out->d[t].idx = bwd[out->d[t].idx];
    qsort(out->d, (size_t)out->n, sizeof(tup), lsv_cmp);
    free(bwd); sv_free(&unit); sv_free(&w);
}
/* Forrest-Tomlin-style update, mod.rs:92-155; returns 0 when the spike has no pivot value (singular) */
static int lu_change_basis(lu_t *f, int32_t pivot_row_index, const svec *spike_in) {
    const int32_t m = f->m;
    int64_t p = f->cp_fwd[pivot_row_index];
    for (int32_t k = 0; k < f->n_updates; k++) p = lu_rot_forward(f->rot[k], m, p);
    svec u_bar, r; sv_init(&u_bar); sv_init(&r);
    for (int32_t j = (int32_t)p + 1; j < m; j++) { const double *u = sv_get(&f->upper[j], p); if (u) sv_push(&u_bar, j, *u); }
    lu_invert_upper_left(f, &u_bar, &r);
    for (int64_t k = 0; k < u_bar.n; k++) { int found; int64_t pos = lsv_find(&f->upper[u_bar.d[k].idx], p, &found); if (found) lsv_remove(&f->upper[u_bar.d[k].idx], pos); }
    if (f->n_updates == f->cap_updates) {
        f->cap_updates = f->cap_updates ? 2 * f->cap_updates : 16;
        f->etas = (lu_eta *)realloc(f->etas, sizeof(lu_eta) * (size_t)f->cap_updates);
        f->rot = (int32_t *)realloc(f->rot, sizeof(int32_t) * (size_t)f->cap_updates);
    }
    lu_eta *eta = &f->etas[f->n_updates];
    eta->values = r; eta->pivot = (int32_t)p;                         /* (r is moved into the eta) */
    svec spike; sv_init(&spike); lsv_copy(&spike, spike_in);
    lu_eta_update_spike_pivot_value(eta, &spike);
    int found; (void)lsv_find(&spike, p, &found);
    /* column p := spike; columns p + 1 .. move one to the left, column p to the back (mod.rs:139-148) */
    svec moved = spike;
    sv_free(&f->upper[p]);
    for (int32_t j = (int32_t)p; j + 1 < m; j++) f->upper[j] = f->upper[j + 1];
    f->upper[m - 1] = moved;
    for (int32_t j = (int32_t)p; j < m; j++) lu_rot_forward_sorted((int32_t)p, m, &f->upper[j]);
    f->rot[f->n_updates] = (int32_t)p;
    f->n_updates++;
    sv_free(&u_bar);
    return found;
}

#endif
