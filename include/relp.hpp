// relp.hpp -- C++17 host-side mirror of the reference's interface for the pivot path, header-only, over the
// C ABI of relp_engine.h (link with -lrelp_engine; no HIP headers needed on this side).
//
// The reference (/root/reference, Rust) drives the path through three types; the same names, argument
// meaning and failure behaviour are kept here so that code and tests written against the reference read the
// same:
//   MatrixData        src/data/linear_program/... matrix_provider/matrix_data.rs:54-90 (+ `new`, :92-147)
//   Tableau           src/algorithm/two_phase/tableau/mod.rs:24-289   (`Tableau<Carry<f64, _>, Partially | NonArtificial>`)
//   PivotRule         src/algorithm/two_phase/strategy/pivot_rule.rs:21-126
//   OptimizationResult src/algorithm/mod.rs:46-50;  SolveRelaxation::solve_relaxation  two_phase/mod.rs:30-76
// Where the reference panics (`expect`, `panic!`, `debug_assert!`), these functions throw relp_host::Error
// carrying the status code and the engine's message; nothing aborts the process.
//
// The field is f64 (the build's extension; the reference has no float field): comparisons that are exact in the
// reference carry the tolerances of relp_config_t.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "relp_engine.h"

namespace relp_host {

class Error : public std::runtime_error {
  public:
    Error(relp_status_t status, const std::string& what) : std::runtime_error(what), status_(status) {}
    relp_status_t status() const { return status_; }

  private:
    relp_status_t status_;
};

// data/linear_algebra/vector/sparse.rs: (index, value) pairs sorted by index, zeros dropped
using SparseVector = std::vector<std::pair<int32_t, double>>;

// strategy/pivot_rule.rs:38,62,97
enum class PivotRule : int32_t {
    FirstProfitable = RELP_RULE_FIRST_PROFITABLE,
    FirstProfitableWithMemory = RELP_RULE_FIRST_PROFITABLE_WITH_MEMORY,
    SteepestDescent = RELP_RULE_STEEPEST_DESCENT,
};

// which device representation maintains the basis inverse: the `IM` type parameter of the reference
// (`Carry<_, BasisInverseRows<_>>`, `Carry<_, LUDecomposition<_>>`) plus the dense tableau
enum class InverseMaintenance : int32_t {
    BasisInverseRows = RELP_ENGINE_REVISED,
    DenseTableau = RELP_ENGINE_TABLEAU,
    LUDecomposition = RELP_ENGINE_LU,
};

// algorithm/mod.rs:46-50
struct OptimizationResult {
    enum Kind { Infeasible, FiniteOptimum, Unbounded } kind;
    SparseVector solution;       // FiniteOptimum: the basic feasible solution over the provider's columns
    bool operator==(const OptimizationResult& o) const { return kind == o.kind && solution == o.solution; }
};

// matrix_provider/matrix_data.rs:54-90.  Rows ordered [== | range | <= | >=]; slack and bound columns / rows
// are virtual.  Owns its arrays (the reference borrows them).
class MatrixData {
  public:
    // `MatrixData::new(constraints, b, ranges, nr_eq, nr_range, nr_le, nr_ge, variables)` (:92-147) with the
    // constraints given like `ColumnMajor::from_test_data`: dense rows
    static MatrixData from_rows(const std::vector<std::vector<double>>& rows, int32_t nr_columns, std::vector<double> b,
                                std::vector<double> ranges, int32_t nr_eq, int32_t nr_range, int32_t nr_le, int32_t nr_ge,
                                std::vector<double> cost, std::vector<double> upper_bound = {}) {
        MatrixData md;
        md.set_counts(nr_columns, nr_eq, nr_range, nr_le, nr_ge);
        if ((int32_t)rows.size() != md.nr_constraints()) throw Error(RELP_E_ARG, "row count does not match the constraint counts");
        md.col_ptr_.assign(1, 0);
        for (int32_t j = 0; j < nr_columns; ++j) {
            for (int32_t i = 0; i < (int32_t)rows.size(); ++i) {
                if ((int32_t)rows[i].size() != nr_columns) throw Error(RELP_E_ARG, "ragged constraint rows");
                if (rows[i][j] != 0.0) { md.row_idx_.push_back(i); md.values_.push_back(rows[i][j]); }
            }
            md.col_ptr_.push_back((int64_t)md.row_idx_.size());
        }
        md.finish(std::move(b), std::move(ranges), std::move(cost), std::move(upper_bound));
        return md;
    }
    // sparse columns in CSC (row indices sorted inside a column)
    static MatrixData from_csc(int32_t nr_columns, std::vector<int64_t> col_ptr, std::vector<int32_t> row_idx,
                               std::vector<double> values, std::vector<double> b, std::vector<double> ranges, int32_t nr_eq,
                               int32_t nr_range, int32_t nr_le, int32_t nr_ge, std::vector<double> cost,
                               std::vector<double> upper_bound = {}) {
        MatrixData md;
        md.set_counts(nr_columns, nr_eq, nr_range, nr_le, nr_ge);
        if ((int32_t)col_ptr.size() != nr_columns + 1) throw Error(RELP_E_ARG, "col_ptr must have nr_columns + 1 entries");
        md.col_ptr_ = std::move(col_ptr); md.row_idx_ = std::move(row_idx); md.values_ = std::move(values);
        md.finish(std::move(b), std::move(ranges), std::move(cost), std::move(upper_bound));
        return md;
    }

    int32_t nr_normal_variables() const { return nr_normal_; }
    int32_t nr_constraints() const { return nr_eq_ + nr_range_ + nr_le_ + nr_ge_; }          // matrix_data.rs:395
    int32_t nr_variable_bounds() const {                                                      // :399-405
        int32_t k = 0;
        for (double u : upper_) k += std::isfinite(u) ? 1 : 0;
        return k + nr_range_;
    }
    int32_t nr_rows() const { return nr_constraints() + nr_variable_bounds(); }               // :407
    // normal + range/<=/>= slacks + bound slacks (:37-52)
    int32_t nr_columns() const { return nr_normal_ + nr_range_ + nr_le_ + nr_ge_ + nr_variable_bounds(); }

    // Beyond the reference (which solves the data as read): geometric scaling, rows then columns by 1 / sqrt(min |a| max |a|) of the
    // line, `sweeps` times, every factor rounded to a power of two -- A' = R A S, b' = R b, ranges' = R ranges, c' = S c, upper
    // bounds' = ub / s; x = S x' and the objective value is unchanged.  What the PILOT family needs (rust-lp_amd/matrix_data.py:
    // scaled, the same procedure; profiles/r04_corpus_verified.md).
    struct Scaled;
    Scaled scaled(int sweeps = 4) const;
    // a basic feasible solution of scaled().data in the units of *this (columns as in matrix_data.rs:403-409)
    SparseVector unscale(const SparseVector& bfs, const std::vector<double>& r, const std::vector<double>& s) const {
        std::vector<int32_t> bounded;
        for (int32_t j = 0; j < nr_normal_; ++j) if (std::isfinite(upper_[j])) bounded.push_back(j);
        const int32_t o_range = nr_normal_, o_le = o_range + nr_range_, o_ge = o_le + nr_le_, o_bound = o_ge + nr_ge_, o_rb = o_bound + (int32_t)bounded.size();
        SparseVector out;
        for (const auto& e : bfs) {
            const int32_t c = e.first;
            double v = e.second;
            if (c < o_range) v *= s[c];
            else if (c < o_le) v /= r[nr_eq_ + (c - o_range)];
            else if (c < o_ge) v /= r[nr_eq_ + nr_range_ + (c - o_le)];
            else if (c < o_bound) v /= r[nr_eq_ + nr_range_ + nr_le_ + (c - o_ge)];
            else if (c < o_rb) v *= s[bounded[c - o_bound]];
            else v /= r[nr_eq_ + (c - o_rb)];
            out.emplace_back(c, v);
        }
        return out;
    }

    relp_matrix_data_t view() const {
        relp_matrix_data_t v{};
        v.nr_normal = nr_normal_; v.nr_eq = nr_eq_; v.nr_range = nr_range_; v.nr_le = nr_le_; v.nr_ge = nr_ge_;
        v.format = RELP_FORMAT_CSC; v.matrix_memory = RELP_MEM_HOST;
        v.col_ptr = col_ptr_.data(); v.row_idx = row_idx_.data(); v.values = values_.data();
        v.dense = nullptr; v.dense_ld = 0;
        v.b = b_.data(); v.ranges = ranges_.data(); v.cost = cost_.data(); v.upper_bound = upper_.data();
        return v;
    }

  private:
    void set_counts(int32_t n, int32_t eq, int32_t rg, int32_t le, int32_t ge) {
        if (n < 0 || eq < 0 || rg < 0 || le < 0 || ge < 0) throw Error(RELP_E_ARG, "negative count");
        nr_normal_ = n; nr_eq_ = eq; nr_range_ = rg; nr_le_ = le; nr_ge_ = ge;
    }
    void finish(std::vector<double> b, std::vector<double> ranges, std::vector<double> cost, std::vector<double> upper) {
        if (upper.empty()) upper.assign(nr_normal_, std::numeric_limits<double>::infinity());
        if ((int32_t)b.size() != nr_constraints() || (int32_t)ranges.size() != nr_range_ || (int32_t)cost.size() != nr_normal_ ||
            (int32_t)upper.size() != nr_normal_)
            throw Error(RELP_E_ARG, "vector lengths do not match the counts");
        b_ = std::move(b); ranges_ = std::move(ranges); cost_ = std::move(cost); upper_ = std::move(upper);
        if (ranges_.empty()) ranges_.reserve(1);
    }
    int32_t nr_normal_ = 0, nr_eq_ = 0, nr_range_ = 0, nr_le_ = 0, nr_ge_ = 0;
    std::vector<int64_t> col_ptr_;
    std::vector<int32_t> row_idx_;
    std::vector<double> values_, b_, ranges_, cost_, upper_;
};

struct MatrixData::Scaled { MatrixData data; std::vector<double> row_scale, column_scale; };
inline MatrixData::Scaled MatrixData::scaled(int sweeps) const {
    const int32_t m = nr_constraints(), n = nr_normal_;
    std::vector<double> r(m, 1.0), s(n, 1.0);
    auto pass = [&](bool by_row) {
        const int32_t count = by_row ? m : n;
        std::vector<double> lo(count, std::numeric_limits<double>::infinity()), hi(count, 0.0);
        for (int32_t j = 0; j < n; ++j)
            for (int64_t e = col_ptr_[j]; e < col_ptr_[j + 1]; ++e) {
                const double v = std::fabs(values_[e]) * r[row_idx_[e]] * s[j];
                if (v == 0.0) continue;
                const int32_t l = by_row ? row_idx_[e] : j;
                lo[l] = std::min(lo[l], v); hi[l] = std::max(hi[l], v);
            }
        std::vector<double>& f = by_row ? r : s;
        for (int32_t l = 0; l < count; ++l) if (hi[l] > 0.0) f[l] /= std::sqrt(lo[l] * hi[l]);
    };
    for (int k = 0; k < sweeps; ++k) { pass(true); pass(false); }
    for (double& v : r) v = std::exp2(std::nearbyint(std::log2(v)));
    for (double& v : s) v = std::exp2(std::nearbyint(std::log2(v)));
    Scaled out{*this, r, s};
    MatrixData& d = out.data;
    for (int32_t j = 0; j < n; ++j) {
        for (int64_t e = col_ptr_[j]; e < col_ptr_[j + 1]; ++e) d.values_[e] = values_[e] * r[row_idx_[e]] * s[j];
        d.cost_[j] = cost_[j] * s[j];
        d.upper_[j] = upper_[j] / s[j];
    }
    for (int32_t i = 0; i < m; ++i) d.b_[i] = b_[i] * r[i];
    for (int32_t k = 0; k < nr_range_; ++k) d.ranges_[k] = ranges_[k] * r[nr_eq_ + k];
    return out;
}

// relp_config_t with the reference's defaults (FirstProfitableWithMemory in phase 1, phase_one.rs:55,97;
// SteepestDescent in phase 2, two_phase/mod.rs:44)
struct Options {
    relp_config_t cfg;
    Options() { relp_default_config(&cfg); }
    Options& inverse_maintenance(InverseMaintenance im) { cfg.engine = (int32_t)im; return *this; }
    Options& phase_one_rule(PivotRule r) { cfg.phase_one_rule = (int32_t)r; return *this; }
    Options& phase_two_rule(PivotRule r) { cfg.phase_two_rule = (int32_t)r; return *this; }
    Options& pivot_rule(PivotRule r) { return phase_one_rule(r).phase_two_rule(r); }
    Options& update_block(int32_t k) { cfg.update_block = k; return *this; }
    Options& trace_capacity(int32_t n) { cfg.trace_capacity = n; return *this; }
    Options& device(int32_t d) { cfg.device = d; return *this; }
    Options& shard(int32_t rank, int32_t count) { cfg.shard_rank = rank; cfg.shard_count = count; return *this; }
    // relp_robust_config: every f64 safeguard and RELP_ENGINE_AUTO instead of the reference's literal rules (no per-file knobs)
    static Options robust() { Options o; relp_robust_config(&o.cfg); return o; }
};

// tableau/mod.rs:24-38.  Starts like `Tableau::<_, Partially<_>>::new(&matrix_data)` (artificial variables on
// the rows the initial basis does not cover, kind/artificial/partially.rs:125-206) and becomes the
// `NonArtificial` tableau when phase 1 ends or `from_basis` is called.  Column indices follow the current kind:
// artificial columns come first while phase() == 1 (partially.rs:72-80).
class Tableau {
  public:
    explicit Tableau(const MatrixData& data, const Options& options = Options()) {
        const relp_matrix_data_t v = data.view();
        relp_engine_t* h = nullptr;
        const relp_status_t st = relp_engine_create(&v, &options.cfg, &h);
        if (st != RELP_OK) {
            const std::string msg = h ? relp_last_error(h) : "relp_engine_create failed";
            if (h) relp_engine_destroy(h);
            throw Error(st, msg);
        }
        h_ = h;
    }
    ~Tableau() { if (h_) relp_engine_destroy(h_); }
    Tableau(Tableau&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Tableau& operator=(Tableau&& o) noexcept { if (this != &o) { if (h_) relp_engine_destroy(h_); h_ = o.h_; o.h_ = nullptr; } return *this; }
    Tableau(const Tableau&) = delete;
    Tableau& operator=(const Tableau&) = delete;

    relp_engine_t* handle() { return h_; }

    // ---- tableau/mod.rs accessors ------------------------------------------------------------------------
    int32_t nr_rows() const { return relp_nr_rows(h_); }                                   // :199
    int32_t nr_columns() const { return relp_nr_columns(h_); }                             // :204
    int32_t nr_artificial_variables() const { return relp_phase(h_) == 1 ? relp_nr_artificial(h_) : 0; }
    int32_t phase() const { return relp_phase(h_); }
    double objective_function_value() { double v = 0; ck(relp_get_objective(h_, &v)); return v; }        // :93
    std::vector<double> relative_costs() { std::vector<double> d(nr_columns()); ck(relp_relative_costs(h_, d.data())); return d; }
    double relative_cost(int32_t j) {                                                      // :102-108
        if (j < 0 || j >= nr_columns()) throw Error(RELP_E_ARG, "column out of range");
        return relative_costs()[j];
    }
    // :122-126, FTRAN; dense m entries (the reference returns the sparse form, see `sparse`)
    std::vector<double> generate_column(int32_t j) { std::vector<double> c(nr_rows()); ck(relp_generate_column(h_, j, c.data())); return c; }
    double generate_element(int32_t i, int32_t j) { double v = 0; ck(relp_generate_element(h_, i, j, &v)); return v; }   // :129-134
    // :221-247 on a column the caller holds (the reference's signature)
    std::optional<int32_t> select_primal_pivot_row(const std::vector<double>& column) {
        if ((int32_t)column.size() != nr_rows()) throw Error(RELP_E_ARG, "column must have nr_rows() entries");
        int32_t found = 0, row = -1;
        ck(relp_select_primal_pivot_row_of(h_, column.data(), &found, &row));
        return found ? std::optional<int32_t>(row) : std::nullopt;
    }
    // ... and on the last generated column, which is still on the device
    std::optional<int32_t> select_primal_pivot_row() {
        int32_t found = 0, row = -1;
        ck(relp_select_primal_pivot_row(h_, &found, &row));
        return found ? std::optional<int32_t>(row) : std::nullopt;
    }
    // PivotRule::select_primal_pivot_column (pivot_rule.rs:25-33): Some((column, relative cost)) or None
    std::optional<std::pair<int32_t, double>> select_primal_pivot_column(PivotRule rule) {
        int32_t found = 0, j = -1; double d = 0;
        ck(relp_select_primal_pivot_column(h_, (int32_t)rule, &found, &j, &d));
        return found ? std::optional<std::pair<int32_t, double>>({j, d}) : std::nullopt;
    }
    // :47-60; consumes the last generated column; returns the leaving column
    int32_t bring_into_basis(int32_t column, int32_t row, double relative_cost) {
        int32_t leaving = -1;
        ck(relp_bring_into_basis(h_, column, row, relative_cost, &leaving));
        return leaving;
    }
    bool is_in_basis(int32_t column) {                                                     // :140
        for (int32_t j : basis_indices()) if (j == column) return true;
        return false;
    }
    std::vector<int32_t> basis_indices() { std::vector<int32_t> v(nr_rows()); ck(relp_get_basis_indices(h_, v.data())); return v; }
    std::vector<double> b() { std::vector<double> v(nr_rows()); ck(relp_get_b(h_, v.data())); return v; }
    std::vector<double> minus_pi() { std::vector<double> v(nr_rows()); ck(relp_get_minus_pi(h_, v.data())); return v; }
    // row-major m x m
    std::vector<double> basis_inverse() { const size_t m = nr_rows(); std::vector<double> v(m * m); ck(relp_get_basis_inverse(h_, v.data())); return v; }
    SparseVector current_bfs() {                                                           // carry/mod.rs:616-625
        const int32_t m = nr_rows();
        std::vector<int32_t> cols(m); std::vector<double> vals(m); int32_t count = 0;
        ck(relp_current_bfs(h_, cols.data(), vals.data(), m, &count));
        SparseVector out;
        for (int32_t k = 0; k < count; ++k) out.emplace_back(cols[k], vals[k]);
        return out;
    }
    int64_t iterations() { int64_t it = 0; ck(relp_get_iterations(h_, &it)); return it; }
    // pivots so far with ratio exactly 0 (the basis changed, the vertex did not)
    int64_t degenerate_pivots() { int64_t n = 0; ck(relp_get_degenerate_pivots(h_, &n)); return n; }
    // f64 only (the exact reference never refactors `BasisInverseRows`): rebuild B^-1, b, -pi from the basis columns
    // every `pivots` basis changes; revised and tableau engines, default 1,000 for sparse input below 4,097 rows, 0 = never
    void set_reinversion_interval(int64_t pivots) { ck(relp_set_reinversion_interval(h_, pivots)); }
    int64_t reinversions() const { return relp_reinversions(h_); }

    // ---- loops ---------------------------------------------------------------------------------------------
    // phase_one::primal (phase_one.rs:125-170) / phase_two::primal (phase_two.rs:22-51): up to max_iters basis
    // changes of the current phase
    relp_outcome_t run(int64_t max_iters = std::numeric_limits<int64_t>::max(), int64_t* done = nullptr) {
        int32_t oc = 0; int64_t n = 0;
        ck(relp_run(h_, max_iters, &n, &oc));
        if (done) *done = n;
        return (relp_outcome_t)oc;
    }
    // SolveRelaxation::solve_relaxation (two_phase/mod.rs:30-76)
    OptimizationResult solve_relaxation(int64_t max_iters = std::numeric_limits<int64_t>::max()) {
        int32_t oc = 0;
        ck(relp_solve_relaxation(h_, max_iters, &oc));
        switch (oc) {
            case RELP_OPTIMAL: return {OptimizationResult::FiniteOptimum, current_bfs()};
            case RELP_UNBOUNDED: return {OptimizationResult::Unbounded, {}};
            case RELP_INFEASIBLE: return {OptimizationResult::Infeasible, {}};
            case RELP_NO_ROW_PHASE_ONE: throw Error(RELP_E_STATE, "Artificial cost can not be unbounded.");   // phase_one.rs:143
            default: throw Error(RELP_E_STATE, "iteration limit reached");
        }
    }
    // ---- one LP on several GPUs (no counterpart in the reference, which is single-threaded) ---------------------
    // Options::shard(rank, count) at construction, the owned structural columns in `MatrixData` (relp_shard_plan);
    // rank 0 makes the RCCL id, the host program hands it to every rank, every rank attaches, then the loop runs
    // inside the library with one all-gather per pivot.
    static std::vector<uint8_t> rccl_unique_id() {
        std::vector<uint8_t> id(RELP_RCCL_ID_BYTES);
        const relp_status_t st = relp_rccl_unique_id(id.data());
        if (st != RELP_OK) throw Error(st, "no RCCL library could be loaded");
        return id;
    }
    void attach_rccl(const std::vector<uint8_t>& id) {
        if (id.size() != RELP_RCCL_ID_BYTES) throw Error(RELP_E_ARG, "the RCCL unique id has 128 bytes");
        ck(relp_rccl_attach(h_, id.data()));
    }
    relp_outcome_t shard_run(int64_t max_iters = std::numeric_limits<int64_t>::max(), int64_t* done = nullptr) {
        int32_t oc = 0; int64_t n = 0;
        ck(relp_shard_run(h_, max_iters, &n, &oc));
        if (done) *done = n;
        return (relp_outcome_t)oc;
    }

    // InverseMaintener::from_basis (carry/mod.rs:428-463): provider column per row; the tableau is NonArtificial after
    void from_basis(const std::vector<int32_t>& basis_columns) {
        if ((int32_t)basis_columns.size() != nr_rows()) throw Error(RELP_E_ARG, "one basis column per row");
        ck(relp_from_basis(h_, basis_columns.data()));
    }

    // ---- the `BasisInverse` surface (carry/mod.rs:68-157) ----------------------------------------------------
    // basis_inverse_row (carry/mod.rs:145-150; lower_upper/mod.rs:204-222; basis_inverse_rows.rs:181-183)
    std::vector<double> basis_inverse_row(int32_t row) {
        std::vector<double> out(nr_rows());
        ck(relp_basis_inverse_row(h_, row, out.data()));
        return out;
    }
    // should_refactor (lower_upper/mod.rs:199-202; basis_inverse_rows.rs:175-179)
    bool should_refactor() { int32_t v = 0; ck(relp_should_refactor(h_, &v)); return v != 0; }
    // generate_column(original_column) (carry/mod.rs:108-118) for sorted (row, value) pairs
    std::vector<double> generate_column(const SparseVector& original_column) {
        std::vector<int32_t> idx; std::vector<double> val;
        for (auto& e : original_column) { idx.push_back(e.first); val.push_back(e.second); }
        std::vector<double> out(nr_rows());
        ck(relp_generate_column_of(h_, idx.data(), val.data(), (int32_t)idx.size(), out.data()));
        return out;
    }
    // InverseMaintener::cost_difference (carry/mod.rs:572-577)
    double cost_difference(const SparseVector& original_column) {
        std::vector<int32_t> idx; std::vector<double> val;
        for (auto& e : original_column) { idx.push_back(e.first); val.push_back(e.second); }
        double out = 0.0;
        ck(relp_cost_difference_of(h_, idx.data(), val.data(), (int32_t)idx.size(), &out));
        return out;
    }
    // LUDecomposition: change_basis on the inverse alone (lower_upper/mod.rs:92-155) and `updates.len()`
    void lu_change_basis(int32_t pivot_row_index) { ck(relp_lu_change_basis(h_, pivot_row_index)); }
    int32_t lu_updates() { int32_t n = 0; ck(relp_lu_updates(h_, &n)); return n; }

  private:
    void ck(relp_status_t st) {
        if (st != RELP_OK) throw Error(st, relp_last_error(h_));
    }
    relp_engine_t* h_ = nullptr;
};

// zeros dropped, like the SparseVector the reference's generate_column returns
inline SparseVector sparse(const std::vector<double>& dense, double zero = 0.0) {
    SparseVector out;
    for (int32_t i = 0; i < (int32_t)dense.size(); ++i)
        if (std::fabs(dense[i]) > zero) out.emplace_back(i, dense[i]);
    return out;
}

// `data.solve_relaxation::<IM>()` (two_phase/mod.rs:30)
inline OptimizationResult solve_relaxation(const MatrixData& data, const Options& options = Options()) {
    Tableau t(data, options);
    return t.solve_relaxation();
}

// Beyond the reference: one answer, checked, out of several attempts (rust-lp_amd/engine.py: solve_verified is the same
// procedure).  Legs of (data as read | scaled, configuration, engine) in a fixed order, each with a pivot budget; `FiniteOptimum`
// stands only with relp_check_basis residuals inside (1e-5, 1e-3, b >= -1e-6) -- its solution is in the units of the data, whichever
// leg found it --, `Infeasible` / `Unbounded` only when every leg has run, none reached a verified optimum and two engines said so
// on the data as read.  An unverified result is the last leg's, and may be none at all.
struct VerifiedResult { std::optional<OptimizationResult> result; bool verified = false; bool scaled = false; int legs_tried = 0; };
inline VerifiedResult solve_verified(const MatrixData& data, int64_t pivots_per_leg = -1) {
    // (robust = 2: the safeguards with PivotRule::SteepestDescent in phase 1 as well -- what ends the cycling of TUFF, DEGEN3, CYCLE)
    // (times: a long leg with that many times the pivot budget -- DFL001 takes 1,990,000 pivots on the last one)
    struct Leg { bool scaled; int robust; int32_t engine; int times = 1; };
    static const Leg legs[13] = {{false, 1, RELP_ENGINE_LU}, {true, 1, RELP_ENGINE_LU}, {true, 2, RELP_ENGINE_LU}, {true, 2, RELP_ENGINE_TABLEAU},
                                 {false, 2, RELP_ENGINE_LU}, {true, 1, RELP_ENGINE_REVISED}, {false, 1, RELP_ENGINE_REVISED},
                                 {false, 1, RELP_ENGINE_TABLEAU}, {true, 1, RELP_ENGINE_TABLEAU},
                                 {false, 0, RELP_ENGINE_LU}, {false, 0, RELP_ENGINE_REVISED}, {false, 0, RELP_ENGINE_TABLEAU},
                                 {false, 2, RELP_ENGINE_TABLEAU, 12}};
    VerifiedResult out;
    std::optional<MatrixData::Scaled> sc;
    uint32_t infeasible_on = 0, unbounded_on = 0;      // bit per engine, data as read
    for (const Leg& leg : legs) {
        Options o = leg.robust ? Options::robust() : Options();
        if (leg.robust == 2) o.pivot_rule(PivotRule::SteepestDescent);
        o.inverse_maintenance((InverseMaintenance)leg.engine);
        ++out.legs_tried;
        if (leg.scaled && !sc) sc = data.scaled();
        try {
            Tableau t(leg.scaled ? sc->data : data, o);
            const int64_t budget = leg.times * (pivots_per_leg >= 0 ? pivots_per_leg : 30 * ((int64_t)t.nr_rows() + t.nr_columns()));
            int64_t total = 0, done = 0;
            relp_outcome_t oc = RELP_RUNNING;
            while (total < budget) {
                oc = t.run(std::min<int64_t>(20000, budget - total), &done);
                total += done;
                if (oc != RELP_RUNNING && oc != RELP_PHASE_ONE_DONE) break;
            }
            if (oc == RELP_OPTIMAL) {
                double ident = 0, basic = 0, min_b = 0;
                if (relp_check_basis(t.handle(), &ident, &basic, &min_b) == RELP_OK && ident <= 1e-5 && basic <= 1e-3 && min_b >= -1e-6) {
                    SparseVector x = t.current_bfs();
                    if (leg.scaled) x = data.unscale(x, sc->row_scale, sc->column_scale);
                    out.result = OptimizationResult{OptimizationResult::FiniteOptimum, std::move(x)};
                    out.verified = true; out.scaled = leg.scaled;
                    return out;
                }
            } else if (oc == RELP_INFEASIBLE || oc == RELP_UNBOUNDED) {
                out.result = OptimizationResult{oc == RELP_INFEASIBLE ? OptimizationResult::Infeasible : OptimizationResult::Unbounded, {}};
                // (a claim counts only from a state that is still a basis: an exploded tableau also ends `infeasible`)
                double ident = 0, basic = 0, min_b = 0;
                if (!leg.scaled && relp_check_basis(t.handle(), &ident, &basic, &min_b) == RELP_OK && ident <= 1e-5 && basic <= 1e-3 &&
                    std::isfinite(t.objective_function_value()))
                    (oc == RELP_INFEASIBLE ? infeasible_on : unbounded_on) |= 1u << leg.engine;
            }
        } catch (const Error&) {
            // (a leg that fails -- no row in phase 1, a singular basis -- is a leg without an answer)
        }
    }
    auto two = [](uint32_t bits) { return (bits & (bits - 1)) != 0; };
    if (two(infeasible_on)) { out.result = OptimizationResult{OptimizationResult::Infeasible, {}}; out.verified = true; }
    else if (two(unbounded_on)) { out.result = OptimizationResult{OptimizationResult::Unbounded, {}}; out.verified = true; }
    return out;
}

}  // namespace relp_host
