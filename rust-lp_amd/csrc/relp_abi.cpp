// relp_abi.cpp -- the extern "C" boundary declared in include/relp_engine.h.  Thin forwarding only.
#include <cstring>
#include <new>

#include "relp_engine.hpp"

using relp::Engine;

struct relp_engine { Engine impl; };

#define H(h) ((h)->impl)

extern "C" {

void relp_default_config(relp_config_t* cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->device = -1;
    cfg->phase_one_rule = RELP_RULE_FIRST_PROFITABLE_WITH_MEMORY;   // phase_one.rs:55,97
    cfg->phase_two_rule = RELP_RULE_STEEPEST_DESCENT;               // two_phase/mod.rs:44,101
    cfg->tol_cost = 1e-7; cfg->tol_pivot = 1e-5; cfg->tol_zero = 1e-11; cfg->tol_tie = 1e-9; cfg->tol_feas = 1e-7;
    cfg->poll_interval = 64;
    cfg->trace_capacity = 0;
    cfg->shard_rank = 0; cfg->shard_count = 1;
    cfg->update_block = -1;
    cfg->engine = RELP_ENGINE_REVISED;
}

void relp_robust_config(relp_config_t* cfg) {
    relp_default_config(cfg);
    cfg->ratio_rule = RELP_RATIO_LARGEST_PIVOT;
    cfg->artificial_removal = RELP_ARTIFICIAL_TEXTBOOK;
    cfg->pivot_rescue = 1;
    cfg->auto_reinversion = 1;
    cfg->engine = RELP_ENGINE_AUTO;
}
int32_t relp_engine_kind(const relp_engine_t* h) { return h ? H(h).engine_kind() : -1; }
relp_status_t relp_robust_stats(const relp_engine_t* h, int64_t* out4) { return (h && out4) ? H(h).robust_stats(out4) : RELP_E_ARG; }

const char* relp_last_error(const relp_engine_t* h) { return h ? H(h).last_error() : "null handle"; }
const char* relp_version(void) { return "relp-mi355x 0.2 (gfx950; engines: explicit inverse, dense tableau, sparse LU + Forrest-Tomlin)"; }

relp_status_t relp_engine_create(const relp_matrix_data_t* md, const relp_config_t* cfg, relp_engine_t** out) {
    if (!md || !cfg || !out) return RELP_E_ARG;
    relp_engine_t* h = new (std::nothrow) relp_engine_t();
    if (!h) return RELP_E_ALLOC;
    relp_status_t st = H(h).create(*md, *cfg);
    *out = h;   // returned even on failure so that relp_last_error can be read; caller destroys it
    return st;
}

void relp_engine_destroy(relp_engine_t* h) { delete h; }
relp_status_t relp_set_stream(relp_engine_t* h, void* s) { return h ? H(h).set_stream((hipStream_t)s) : RELP_E_ARG; }

relp_status_t relp_select_primal_pivot_column(relp_engine_t* h, int32_t rule, int32_t* found, int32_t* column, double* cost) {
    return h ? H(h).select_primal_pivot_column(rule, found, column, cost) : RELP_E_ARG;
}
relp_status_t relp_relative_costs(relp_engine_t* h, double* out) { return (h && out) ? H(h).relative_costs(out) : RELP_E_ARG; }
relp_status_t relp_generate_column(relp_engine_t* h, int32_t column, double* out) { return h ? H(h).generate_column(column, out) : RELP_E_ARG; }
relp_status_t relp_generate_element(relp_engine_t* h, int32_t row, int32_t column, double* out) {
    return h ? H(h).generate_element(row, column, out) : RELP_E_ARG;
}
relp_status_t relp_select_primal_pivot_row(relp_engine_t* h, int32_t* found, int32_t* row) {
    return h ? H(h).select_primal_pivot_row(found, row) : RELP_E_ARG;
}
relp_status_t relp_select_primal_pivot_row_of(relp_engine_t* h, const double* column, int32_t* found, int32_t* row) {
    return (h && column) ? H(h).select_primal_pivot_row_of(column, found, row) : RELP_E_ARG;
}
relp_status_t relp_bring_into_basis(relp_engine_t* h, int32_t column, int32_t row, double cost, int32_t* leaving) {
    return h ? H(h).bring_into_basis(column, row, cost, leaving) : RELP_E_ARG;
}

relp_status_t relp_run(relp_engine_t* h, int64_t max_iters, int64_t* done, int32_t* outcome) {
    return h ? H(h).run(max_iters, done, outcome) : RELP_E_ARG;
}
relp_status_t relp_solve_relaxation(relp_engine_t* h, int64_t max_iters, int32_t* outcome) {
    return h ? H(h).solve_relaxation(max_iters, outcome) : RELP_E_ARG;
}
relp_status_t relp_from_basis(relp_engine_t* h, const int32_t* basis) { return (h && basis) ? H(h).from_basis(basis) : RELP_E_ARG; }

relp_status_t relp_flush(relp_engine_t* h) { return h ? H(h).flush() : RELP_E_ARG; }
relp_status_t relp_set_reinversion_interval(relp_engine_t* h, int64_t pivots) { return h ? H(h).set_reinversion_interval(pivots) : RELP_E_ARG; }
int64_t relp_reinversions(const relp_engine_t* h) { return h ? H(h).reinversions() : -1; }
int32_t relp_update_block(const relp_engine_t* h) { return h ? H(h).update_block() : -1; }
relp_status_t relp_lu_stats(const relp_engine_t* h, int64_t* out8) { return (h && out8) ? H(h).lu_stats(out8) : RELP_E_ARG; }
relp_status_t relp_lu_lookahead_stats(const relp_engine_t* h, int64_t* out4) { return (h && out4) ? H(h).lu_lookahead_stats(out4) : RELP_E_ARG; }
relp_status_t relp_lu_kernel_layout(const relp_engine_t* h, int32_t* out4) { return (h && out4) ? H(h).lu_kernel_layout(out4) : RELP_E_ARG; }
relp_status_t relp_lu_set_device_factorisation(relp_engine_t* h, int32_t on) { return h ? H(h).lu_set_device_factorisation(on != 0) : RELP_E_ARG; }
relp_status_t relp_lu_factor_residual(relp_engine_t* h, double* out) { return (h && out) ? H(h).lu_factor_residual(out) : RELP_E_ARG; }
relp_status_t relp_lu_device_factorisation_stats(const relp_engine_t* h, int64_t* out6) { return (h && out6) ? H(h).luf_stats(out6) : RELP_E_ARG; }
relp_status_t relp_shard_inject_failure(relp_engine_t* h, int64_t after_pivots) { if (!h) return RELP_E_ARG; H(h).shard_inject_failure(after_pivots); return RELP_OK; }
relp_status_t relp_lu_phase_cycles(relp_engine_t* h, int64_t* out16) { return (h && out16) ? H(h).lu_phase_cycles(out16) : RELP_E_ARG; }
relp_status_t relp_basis_inverse_row(relp_engine_t* h, int32_t row, double* out_m) { return (h && out_m) ? H(h).basis_inverse_row(row, out_m) : RELP_E_ARG; }
relp_status_t relp_should_refactor(relp_engine_t* h, int32_t* out) { return (h && out) ? H(h).should_refactor(out) : RELP_E_ARG; }
relp_status_t relp_generate_column_of(relp_engine_t* h, const int32_t* idx, const double* val, int32_t nnz, double* out_m) {
    return h ? H(h).generate_column_of(idx, val, nnz, out_m) : RELP_E_ARG;
}
relp_status_t relp_cost_difference_of(relp_engine_t* h, const int32_t* idx, const double* val, int32_t nnz, double* out) {
    return h ? H(h).cost_difference_of(idx, val, nnz, out) : RELP_E_ARG;
}
relp_status_t relp_lu_change_basis(relp_engine_t* h, int32_t row) { return h ? H(h).lu_change_basis(row) : RELP_E_ARG; }
relp_status_t relp_lu_set_factors(relp_engine_t* h, const int64_t* l_ptr, const int32_t* l_idx, const double* l_val,
                                  const int64_t* u_ptr, const int32_t* u_idx, const double* u_val) {
    return h ? H(h).lu_set_factors(l_ptr, l_idx, l_val, u_ptr, u_idx, u_val) : RELP_E_ARG;
}
relp_status_t relp_lu_updates(relp_engine_t* h, int32_t* count) { return (h && count) ? H(h).lu_updates(count) : RELP_E_ARG; }
relp_status_t relp_lu_get_update(relp_engine_t* h, int32_t k, int32_t* pivot, int32_t* idx, double* val, int32_t cap, int32_t* nnz) {
    return h ? H(h).lu_get_update(k, pivot, idx, val, cap, nnz) : RELP_E_ARG;
}
relp_status_t relp_lu_get_upper(relp_engine_t* h, int64_t* col_ptr, int32_t* row_idx, double* values, int64_t cap, int64_t* nnz) {
    return h ? H(h).lu_get_upper(col_ptr, row_idx, values, cap, nnz) : RELP_E_ARG;
}
relp_status_t relp_shard_flush_begin(relp_engine_t* h, double** snap, int64_t* len) {
    return h ? H(h).shard_flush_begin(snap, len) : RELP_E_ARG;
}
relp_status_t relp_shard_flush_end(relp_engine_t* h) { return h ? H(h).shard_flush_end() : RELP_E_ARG; }
relp_status_t relp_shard_set_collectives(relp_engine_t* h, relp_allgather_fn ag, relp_allreduce_sum_fn ar, void* ctx) {
    return h ? H(h).shard_set_collectives(ag, ar, ctx) : RELP_E_ARG;
}
relp_status_t relp_shard_run(relp_engine_t* h, int64_t max_iters, int64_t* done, int32_t* outcome) {
    return h ? H(h).shard_run(max_iters, done, outcome) : RELP_E_ARG;
}
relp_status_t relp_rccl_attach(relp_engine_t* h, const uint8_t* id) { return (h && id) ? H(h).rccl_attach(id) : RELP_E_ARG; }

int32_t relp_nr_rows(const relp_engine_t* h) { return h ? H(h).nr_rows() : -1; }
int32_t relp_nr_columns(const relp_engine_t* h) { return h ? H(h).nr_columns() : -1; }
int32_t relp_phase(const relp_engine_t* h) { return h ? H(h).phase() : -1; }
int32_t relp_nr_artificial(const relp_engine_t* h) { return h ? H(h).nr_artificial() : -1; }
relp_status_t relp_get_objective(relp_engine_t* h, double* out) { return (h && out) ? H(h).get_objective(out) : RELP_E_ARG; }
relp_status_t relp_get_b(relp_engine_t* h, double* out) { return (h && out) ? H(h).get_vector(0, out) : RELP_E_ARG; }
relp_status_t relp_get_minus_pi(relp_engine_t* h, double* out) { return (h && out) ? H(h).get_vector(1, out) : RELP_E_ARG; }
relp_status_t relp_get_basis_indices(relp_engine_t* h, int32_t* out) { return (h && out) ? H(h).get_basis_indices(out) : RELP_E_ARG; }
relp_status_t relp_get_basis_inverse(relp_engine_t* h, double* out) { return (h && out) ? H(h).get_basis_inverse(out) : RELP_E_ARG; }
relp_status_t relp_current_bfs(relp_engine_t* h, int32_t* cols, double* vals, int32_t cap, int32_t* count) {
    return h ? H(h).current_bfs(cols, vals, cap, count) : RELP_E_ARG;
}
relp_status_t relp_get_iterations(relp_engine_t* h, int64_t* out) { return (h && out) ? H(h).get_iterations(out) : RELP_E_ARG; }
relp_status_t relp_get_degenerate_pivots(relp_engine_t* h, int64_t* out) { return (h && out) ? H(h).get_degenerate_pivots(out) : RELP_E_ARG; }
relp_status_t relp_get_trace(relp_engine_t* h, int32_t* phase, int32_t* entering, int32_t* row, int32_t* leaving,
                             int64_t cap, int64_t* count) {
    return h ? H(h).get_trace(phase, entering, row, leaving, cap, count) : RELP_E_ARG;
}
relp_status_t relp_check_basis(relp_engine_t* h, double* e1, double* e2, double* mb) { return h ? H(h).check_basis(e1, e2, mb) : RELP_E_ARG; }

relp_status_t relp_profile_enable(relp_engine_t* h, int32_t enable, int64_t max_launches, int32_t sample_every) {
    return h ? H(h).profile_enable(enable != 0, max_launches, sample_every) : RELP_E_ARG;
}
relp_status_t relp_profile_read(relp_engine_t* h, int32_t kid, int64_t* launches, double* total_ms) {
    return h ? H(h).profile_read(kid, launches, total_ms) : RELP_E_ARG;
}

relp_status_t relp_synth_fill_dense(double* device_a, int64_t ld, int32_t m, int32_t n, uint64_t seed,
                                    int64_t first_column, void* hip_stream) {
    if (!device_a || ld < m || m < 0 || n < 0) return RELP_E_ARG;
    relp::launch_fill_dense(device_a, ld, m, n, seed, first_column, (hipStream_t)hip_stream);
    if (hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) return RELP_E_HIP;
    return hipGetLastError() == hipSuccess ? RELP_OK : RELP_E_HIP;
}
relp_status_t relp_device_alloc(void** out, int64_t bytes) {
    if (!out || bytes < 0) return RELP_E_ARG;
    return hipMalloc(out, (size_t)(bytes ? bytes : 1)) == hipSuccess ? RELP_OK : RELP_E_ALLOC;
}
relp_status_t relp_device_free(void* p) { return hipFree(p) == hipSuccess ? RELP_OK : RELP_E_HIP; }

relp_status_t relp_shard_ranges(const relp_engine_t* h, int32_t* col_lo, int32_t* col_hi, int32_t* row_lo,
                                int32_t* row_hi, int32_t* stride) {
    if (!h) return RELP_E_ARG;
    H(h).shard_ranges(col_lo, col_hi, row_lo, row_hi, stride);
    return RELP_OK;
}
void relp_shard_column_range(int32_t nr_normal, int32_t rank, int32_t count, int32_t* lo, int32_t* hi) {
    if (count < 1) count = 1;
    const int32_t per = (nr_normal + count - 1) / count;
    int32_t a = rank * per; if (a > nr_normal) a = nr_normal;
    int32_t b = a + per; if (b > nr_normal) b = nr_normal;
    if (lo) *lo = a;
    if (hi) *hi = b;
}
int64_t relp_shard_candidate_len(const relp_engine_t* h) { return h ? H(h).candidate_len() : -1; }
int64_t relp_shard_rho_len(const relp_engine_t* h) { return h ? H(h).rho_len() : -1; }
relp_status_t relp_shard_price(relp_engine_t* h, double* c) { return (h && c) ? H(h).shard_price(c) : RELP_E_ARG; }
relp_status_t relp_shard_select_column(relp_engine_t* h, const double* c, int32_t count) {
    return (h && c) ? H(h).shard_select_column(c, count) : RELP_E_ARG;
}
relp_status_t relp_shard_ftran(relp_engine_t* h, double* a) { return (h && a) ? H(h).shard_ftran(a) : RELP_E_ARG; }
relp_status_t relp_shard_ratio(relp_engine_t* h, const double* a, int32_t count, double* rho) {
    return (h && a && rho) ? H(h).shard_ratio(a, count, rho) : RELP_E_ARG;
}
relp_status_t relp_shard_update(relp_engine_t* h, const double* rho) { return (h && rho) ? H(h).shard_update(rho) : RELP_E_ARG; }
relp_status_t relp_shard_pivot(relp_engine_t* h) { return h ? H(h).shard_pivot() : RELP_E_ARG; }
relp_status_t relp_shard_plan(const relp_matrix_data_t* md, const relp_config_t* cfg, int32_t* col_lo, int32_t* col_hi) {
    if (!md || !cfg) return RELP_E_ARG;
    const int32_t G = cfg->shard_count < 1 ? 1 : cfg->shard_count, g = cfg->shard_rank;
    if (cfg->engine != RELP_ENGINE_TABLEAU) { relp_shard_column_range(md->nr_normal, g, G, col_lo, col_hi); return RELP_OK; }
    // stored columns = artificial | structural | virtual (see relp_engine.cpp)
    int32_t nr_bounds = 0;
    for (int32_t j = 0; j < md->nr_normal; ++j) if (md->upper_bound && md->upper_bound[j] < 1e300 && md->upper_bound[j] > -1e300) ++nr_bounds;
    const int32_t mc = md->nr_eq + md->nr_range + md->nr_le + md->nr_ge;
    const int32_t m = mc + nr_bounds + md->nr_range;
    const int32_t nr_real = md->nr_le + nr_bounds + md->nr_range;
    const int32_t na = m - nr_real;
    const int32_t n_store = na + md->nr_normal + md->nr_range + md->nr_le + md->nr_ge + nr_bounds + md->nr_range;
    int32_t per = (n_store + G - 1) / G; per += per % 2;
    int32_t lo = g * per; if (lo > n_store) lo = n_store;
    int32_t hi = lo + per; if (hi > n_store) hi = n_store;
    int32_t a = lo - na; if (a < 0) a = 0; if (a > md->nr_normal) a = md->nr_normal;
    int32_t b = hi - na; if (b < 0) b = 0; if (b > md->nr_normal) b = md->nr_normal;
    if (b < a) b = a;
    if (col_lo) *col_lo = a;
    if (col_hi) *col_hi = b;
    return RELP_OK;
}
relp_status_t relp_poll(relp_engine_t* h, int32_t* outcome, int64_t* iterations) { return h ? H(h).poll(outcome, iterations) : RELP_E_ARG; }

}  // extern "C"
