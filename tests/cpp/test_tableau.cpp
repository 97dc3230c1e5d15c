// C++ host-side parity tests over include/relp.hpp, written the way the reference writes its own unit tests for
// the pivot path (names of the mirrored #[test] functions in the comments; paths relative to /root/reference/).
// The expected values are the reference's known answers (data), in f64 with |error| <= 1e-12.
// Runs on the GPU box: `tests/cpp/test_tableau` (built by tests/cpp/Makefile, `__graft_entry__.build()`);
// exit code 0 = all checks passed.  tests/test_cpp_host.py runs it under pytest (-m gpu).
#include <cmath>
#include <cstdio>
#include <string>

#include "relp.hpp"

using namespace relp_host;

static int g_checks = 0, g_failed = 0;
#define CHECK(cond)                                                                                   \
    do {                                                                                              \
        ++g_checks;                                                                                   \
        if (!(cond)) { ++g_failed; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); }    \
    } while (0)
#define CHECK_THROWS(expr, code)                                                                      \
    do {                                                                                              \
        ++g_checks;                                                                                   \
        bool thrown_ = false;                                                                         \
        try { (void)(expr); } catch (const Error& e) { thrown_ = e.status() == (code); }              \
        if (!thrown_) { ++g_failed; std::printf("FAILED %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expr, #code); } \
    } while (0)

static bool near(double a, double b, double tol = 1e-12) { return std::fabs(a - b) <= tol * std::fmax(1.0, std::fabs(b)); }
static bool near(const std::vector<double>& a, const std::vector<double>& b, double tol = 1e-12) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) if (!near(a[i], b[i], tol)) return false;
    return true;
}
static bool near(const SparseVector& a, const SparseVector& b, double tol = 1e-12) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) if (a[i].first != b[i].first || !near(a[i].second, b[i].second, tol)) return false;
    return true;
}

// src/tests/problem_2.rs:73-114 `create_matrix_data_data` + `matrix_data_form` (3 equality rows, 5 columns)
static MatrixData problem_2() {
    return MatrixData::from_rows({{3, 2, 1, 0, 0}, {5, 1, 1, 1, 0}, {2, 5, 1, 0, 1}}, 5, {1, 3, 4}, {}, 3, 0, 0, 0, {1, 1, 1, 1, 1});
}
// src/tests/problem_1.rs:313-365 (TESTPROB after standardisation: one ==, one >= row, two bounded variables)
static MatrixData problem_1() {
    const double inf = std::numeric_limits<double>::infinity();
    return MatrixData::from_rows({{0, -1, 1}, {1, 0, 1}}, 3, {6, 10}, {}, 1, 0, 0, 1, {1, 4, 9}, {4, 2, inf});
}
// problem_2::artificial_tableau_form: Tableau::<_, Partially<_>>::new
static Tableau artificial_tableau(const MatrixData& md, InverseMaintenance im) {
    return Tableau(md, Options().inverse_maintenance(im).pivot_rule(PivotRule::FirstProfitable));
}
// tableau/mod.rs:378-403 helper `tableau`: NonArtificial tableau on the basis (2, 3, 4)
static Tableau tableau(const MatrixData& md, InverseMaintenance im) {
    Tableau t(md, Options().inverse_maintenance(im).pivot_rule(PivotRule::FirstProfitable));
    t.from_basis({2, 3, 4});
    return t;
}
// problem_2::tableau_form: the tableau after phase 1 (basis (1, 3, 4), objective 9/2)
static Tableau tableau_form(const MatrixData& md, InverseMaintenance im) {
    Tableau t = artificial_tableau(md, im);
    CHECK(t.run() == RELP_PHASE_ONE_DONE);
    return t;
}

// tableau/mod.rs: cost, relative_cost, generate_column, bring_into_basis
static void tableau_tests(InverseMaintenance im, bool has_from_basis) {
    const MatrixData md = problem_2();
    {   // cost
        Tableau art = artificial_tableau(md, im);
        CHECK(near(art.objective_function_value(), 8));
        CHECK(art.nr_rows() == 3 && art.nr_artificial_variables() == 3 && art.nr_columns() == 3 + 5);
        if (has_from_basis) CHECK(near(tableau(md, im).objective_function_value(), 6));
    }
    {   // relative_cost
        Tableau art = artificial_tableau(md, im);
        CHECK(near(art.relative_cost(0), 0));
        CHECK(near(art.relative_cost(art.nr_artificial_variables() + 0), -10));
        if (has_from_basis) {
            Tableau t = tableau(md, im);
            CHECK(near(t.relative_cost(0), -3));
            CHECK(near(t.relative_cost(1), -3));
            CHECK(near(t.relative_cost(2), 0));
        }
    }
    {   // generate_column
        Tableau art = artificial_tableau(md, im);
        const int32_t j = art.nr_artificial_variables() + 0;
        CHECK(near(art.generate_column(j), {3, 5, 2}));
        CHECK(near(art.relative_cost(j), -10));
        CHECK(near(art.generate_element(1, j), 5));
        if (has_from_basis) {
            Tableau t = tableau(md, im);
            CHECK(near(t.generate_column(0), {3, 2, -1}));
            CHECK(near(sparse(t.generate_column(3)), SparseVector{{1, 1.0}}));
            CHECK(near(t.relative_cost(0), -3));
        }
    }
    {   // bring_into_basis
        Tableau art = artificial_tableau(md, im);
        const int32_t column = art.nr_artificial_variables() + 0;
        const std::vector<double> column_data = art.generate_column(column);
        const auto row = art.select_primal_pivot_row(column_data);
        CHECK(row.has_value() && *row == 0);
        CHECK(art.select_primal_pivot_row() == row);            // the device-resident column gives the same row
        const double cost = art.relative_cost(column);
        CHECK(art.bring_into_basis(column, *row, cost) == 0);  // artificial 0 leaves
        CHECK(art.is_in_basis(column));
        CHECK(!art.is_in_basis(0));
        CHECK(near(art.objective_function_value(), 14.0 / 3.0));
        if (has_from_basis) {
            Tableau t = tableau(md, im);
            const std::vector<double> cd = t.generate_column(1);
            const auto r = t.select_primal_pivot_row(cd);
            CHECK(r.has_value());
            t.bring_into_basis(1, *r, t.relative_cost(1));
            CHECK(t.is_in_basis(1));
            CHECK(near(t.objective_function_value(), 9.0 / 2.0));
        }
    }
    {   // "Pivot value can't be zero." (carry/mod.rs:291): the reference panics, the engine reports
        Tableau art = artificial_tableau(md, im);
        const int32_t column = art.nr_artificial_variables() + 3;          // (0, 1, 0): zero in row 0
        art.generate_column(column);
        CHECK_THROWS(art.bring_into_basis(column, 0, art.relative_cost(column)), RELP_E_ZERO_PIVOT);
        CHECK_THROWS(art.generate_column(99), RELP_E_ARG);
    }
}

// strategy/pivot_rule.rs: find_profitable_column, find_pivot_row
static void pivot_rule_tests(InverseMaintenance im) {
    const MatrixData md = problem_2();
    {
        Tableau art = artificial_tableau(md, im);
        const auto pick = art.select_primal_pivot_column(PivotRule::FirstProfitable);
        CHECK(pick.has_value() && pick->first == 3);
        Tableau t = tableau_form(md, im);
        CHECK(!t.select_primal_pivot_column(PivotRule::FirstProfitable).has_value());
    }
    {
        Tableau art = artificial_tableau(md, im);
        CHECK(art.select_primal_pivot_row({3, 5, 2}) == std::optional<int32_t>(0));
        CHECK(art.select_primal_pivot_row({2, 1, 5}) == std::optional<int32_t>(0));
        CHECK(!art.select_primal_pivot_row({-1, 0, -2}).has_value());      // None = unbounded direction
        Tableau t = tableau_form(md, im);
        CHECK(t.select_primal_pivot_row({3, 2, -1}) == std::optional<int32_t>(0));
        CHECK(t.select_primal_pivot_row({2, -1, 3}) == std::optional<int32_t>(0));
    }
}

// two_phase/mod.rs: simplex, solve_matrix, solve_relaxation_1; src/tests/problem_1.rs, problem_2.rs pins
static void two_phase_tests(InverseMaintenance im) {
    {   // simplex: phase_two::primal::<_, _, FirstProfitable> on tableau_form
        Tableau t = tableau_form(problem_2(), im);
        // post-phase-1 carry, src/tests/problem_2.rs:141-174
        CHECK(near(t.objective_function_value(), 9.0 / 2.0));
        CHECK(near(t.minus_pi(), {2.5, -1, -1}));
        CHECK(near(t.b(), {0.5, 2.5, 1.5}));
        CHECK((t.basis_indices() == std::vector<int32_t>{1, 3, 4}));
        CHECK(t.run() == RELP_OPTIMAL);
        CHECK(near(t.objective_function_value(), 9.0 / 2.0));
    }
    {   // solve_matrix
        const OptimizationResult result = solve_relaxation(problem_2(), Options().inverse_maintenance(im));
        CHECK(result.kind == OptimizationResult::FiniteOptimum);
        CHECK(near(result.solution, SparseVector{{1, 0.5}, {3, 2.5}, {4, 1.5}}));
    }
    {   // solve_relaxation_1
        const MatrixData data = MatrixData::from_rows({{1, 0}, {1, 1}}, 2, {1.5, 2.5}, {}, 0, 0, 2, 0, {-2, -1});
        CHECK(data.nr_columns() == 4);
        const OptimizationResult result = solve_relaxation(data, Options().inverse_maintenance(im));
        CHECK(result.kind == OptimizationResult::FiniteOptimum);
        CHECK(near(result.solution, SparseVector{{0, 1.5}, {1, 1.0}}));
    }
    {   // src/tests/problem_1.rs:376-431 with FirstProfitable in both phases
        Tableau t(problem_1(), Options().inverse_maintenance(im).pivot_rule(PivotRule::FirstProfitable));
        CHECK(t.nr_rows() == 4 && t.nr_columns() == 2 + 6);
        CHECK(near(t.objective_function_value(), 16));           // artificial carry: -obj = -16
        CHECK(near(t.minus_pi(), {-1, -1, 0, 0}));
        CHECK(near(t.b(), {6, 10, 4, 2}));
        CHECK(t.run() == RELP_PHASE_ONE_DONE);
        CHECK(near(t.objective_function_value(), 58));
        CHECK(near(t.minus_pi(), {4, -13, 12, 0}));
        CHECK(near(t.b(), {6, 0, 4, 2}));
        CHECK((t.basis_indices() == std::vector<int32_t>{2, 1, 0, 5}));
        CHECK(near(t.basis_inverse(), {0, 1, -1, 0, -1, 1, -1, 0, 0, 0, 1, 0, 1, -1, 1, 1}));
        CHECK(t.run() == RELP_OPTIMAL);
        CHECK(near(t.current_bfs(), SparseVector{{0, 4.0}, {2, 6.0}, {5, 2.0}}));
        CHECK(near(t.objective_function_value(), 58));          // 1*4 + 9*6; the general form's 54 includes its fixed cost
    }
    {   // OptimizationResult::{Unbounded, Infeasible}
        const MatrixData unbounded = MatrixData::from_rows({{1, -1}, {1, 0}}, 2, {1, 5}, {}, 0, 0, 2, 0, {-1, -1});
        CHECK(solve_relaxation(unbounded, Options().inverse_maintenance(im)).kind == OptimizationResult::Unbounded);
        const MatrixData infeasible = MatrixData::from_rows({{1}, {1}}, 1, {1, 2}, {}, 2, 0, 0, 0, {1});
        CHECK(solve_relaxation(infeasible, Options().inverse_maintenance(im)).kind == OptimizationResult::Infeasible);
    }
}

// The `BasisInverse` surface (carry/mod.rs:68-157) on every engine: after from_basis((2, 3, 4)) on problem_2 the basis is
// B = [[1, 0, 0], [1, 1, 0], [1, 0, 1]] (columns 2, 3, 4), B^-1 = [[1, 0, 0], [-1, 1, 0], [-1, 0, 1]]
static void basis_inverse_tests(InverseMaintenance im) {
    const MatrixData md = problem_2();
    Tableau t = tableau(md, im);
    CHECK(near(t.basis_inverse_row(0), {1, 0, 0}));
    CHECK(near(t.basis_inverse_row(1), {-1, 1, 0}));
    CHECK(near(t.basis_inverse_row(2), {-1, 0, 1}));
    CHECK(!t.should_refactor());
    // generate_column(original_column) for a column the caller supplies = the provider's column 0, (3, 5, 2)
    CHECK(near(t.generate_column(SparseVector{{0, 3.0}, {1, 5.0}, {2, 2.0}}), t.generate_column(0)));
    CHECK(near(t.generate_column(SparseVector{{1, 1.0}}), {0, 1, 0}));
    // cost_difference = (-pi) . column; the basis costs are all 1, so -pi = -(1, 1, 1) B^-1 = (1, -1, -1)
    CHECK(near(t.cost_difference(SparseVector{{0, 3.0}, {1, 5.0}, {2, 2.0}}), 3 - 5 - 2));
    CHECK(near(t.relative_cost(0), 1 + t.cost_difference(SparseVector{{0, 3.0}, {1, 5.0}, {2, 2.0}})));
    if (im == InverseMaintenance::LUDecomposition) {
        // change_basis on the inverse alone: column (3, 5, 2) replaces basis position 0, row 0 of the new inverse is
        // (1/3, 0, 0), and one update is pending
        t.generate_column(SparseVector{{0, 3.0}, {1, 5.0}, {2, 2.0}});
        t.lu_change_basis(0);
        CHECK(t.lu_updates() == 1);
        CHECK(near(t.basis_inverse_row(0), {1.0 / 3.0, 0, 0}));
        CHECK(near(t.basis_inverse_row(1), {-5.0 / 3.0 + 0.0, 1, 0}));
        CHECK(near(t.generate_column(SparseVector{{0, 3.0}, {1, 5.0}, {2, 2.0}}), {1, 0, 0}));
    }
}

int main() {
    try {
        const struct { InverseMaintenance im; const char* name; bool from_basis; } kinds[] = {
            {InverseMaintenance::BasisInverseRows, "BasisInverseRows", true},
            {InverseMaintenance::LUDecomposition, "LUDecomposition", true},
            {InverseMaintenance::DenseTableau, "DenseTableau", true},
        };
        for (const auto& k : kinds) {
            const int before = g_failed;
            tableau_tests(k.im, k.from_basis);
            pivot_rule_tests(k.im);
            two_phase_tests(k.im);
            std::printf("%-18s %s\n", k.name, g_failed == before ? "ok" : "FAILED");
        }
        for (const auto& k : kinds) basis_inverse_tests(k.im);
        {   // change_basis on the inverse alone is the LU back end's (lower_upper/mod.rs:92-155): reported elsewhere, not aborted
            Tableau t(problem_2(), Options().inverse_maintenance(InverseMaintenance::BasisInverseRows));
            CHECK_THROWS(t.lu_change_basis(0), RELP_E_UNSUPPORTED);
            CHECK_THROWS(t.basis_inverse_row(7), RELP_E_ARG);
        }
        CHECK_THROWS(MatrixData::from_rows({{1, 2}}, 2, {1, 2}, {}, 0, 0, 1, 0, {1, 1}), RELP_E_ARG);
        {   // the loop inside the library on a communicator of one rank (RCCL): the single-engine result
            Tableau t(problem_2(), Options().inverse_maintenance(InverseMaintenance::DenseTableau).shard(0, 1));
            CHECK_THROWS(t.shard_run(), RELP_E_STATE);                       // no collectives attached yet
            t.attach_rccl(Tableau::rccl_unique_id());
            CHECK(t.shard_run() == RELP_PHASE_ONE_DONE);
            CHECK(t.shard_run() == RELP_OPTIMAL);
            CHECK(near(t.current_bfs(), SparseVector{{1, 0.5}, {3, 2.5}, {4, 1.5}}));
        }
        {   // solve_verified: the first leg (safeguards, LU) answers and the answer passes relp_check_basis
            const VerifiedResult v = solve_verified(problem_2());
            CHECK(v.verified && v.legs_tried == 1 && v.result.has_value());
            CHECK(v.result->kind == OptimizationResult::FiniteOptimum && near(v.result->solution, SparseVector{{1, 0.5}, {3, 2.5}, {4, 1.5}}));
            // the same LP scaled: factors are powers of two, the solution comes back in the units of the data
            const MatrixData::Scaled sc = problem_2().scaled();
            for (double f : sc.row_scale) CHECK(std::exp2(std::nearbyint(std::log2(f))) == f);
            for (double f : sc.column_scale) CHECK(std::exp2(std::nearbyint(std::log2(f))) == f);
            Tableau ts(sc.data, Options::robust());
            const OptimizationResult rs = ts.solve_relaxation();
            CHECK(rs.kind == OptimizationResult::FiniteOptimum);
            CHECK(near(problem_2().unscale(rs.solution, sc.row_scale, sc.column_scale), SparseVector{{1, 0.5}, {3, 2.5}, {4, 1.5}}));
        }
    } catch (const std::exception& e) {
        std::printf("unexpected exception: %s\n", e.what());
        return 2;
    }
    std::printf("%d checks, %d failed\n", g_checks, g_failed);
    return g_failed == 0 ? 0 : 1;
}
