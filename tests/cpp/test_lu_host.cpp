// test_lu_host.cpp -- the host side of the sparse LU engine without a GPU (rust-lp_amd/csrc/relp_lu.{hpp,cpp}): the
// factorisation P B Q = L U, the level schedules, the fusion of levels into groups (fuse_levels) with the per-pivot "via"
// lists a Forrest-Tomlin update uses to mask a pivot, and the "ELL by pass" image the persistent kernel solves from
// (ell_pack), executed here pass by pass exactly as relp_lu_device.h: ell_solve does (lane sums, first lane stores).
//
// Cases: the matrices of the reference's own factorisation tests (lower_upper/decomposition/mod.rs:301-491: identity,
// off-diagonal, the two Wikipedia examples with the known FTRAN answers of mod.rs:470-489), the 5 x 5 matrix of
// lower_upper/mod.rs:779-867, and seeded random LP-like bases.  The reference's exact factors are not the bar here (its
// Markowitz search is exhaustive, ours is restricted and thresholded, see relp_lu.hpp): the bar is P B Q = L U to 1e-12 and
// solves that agree with a dense Gaussian elimination.
#include "relp_lu.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

using namespace relp;
using Cols = std::vector<std::vector<std::pair<int32_t, double>>>;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond, ...) do { ++g_checks; if (!(cond)) { ++g_failed; std::printf("FAILED %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static std::vector<double> dense_of(int m, const Cols& cols) {
    std::vector<double> a((size_t)m * m, 0.0);
    for (int j = 0; j < m; ++j) for (auto& e : cols[j]) a[(size_t)e.first * m + j] = e.second;
    return a;
}
// x = A^-1 b by Gaussian elimination with partial pivoting (reference answer)
static std::vector<double> dense_solve(int m, std::vector<double> a, std::vector<double> b, bool transposed) {
    if (transposed) for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) std::swap(a[(size_t)i * m + j], a[(size_t)j * m + i]);
    for (int k = 0; k < m; ++k) {
        int piv = k;
        for (int i = k + 1; i < m; ++i) if (std::fabs(a[(size_t)i * m + k]) > std::fabs(a[(size_t)piv * m + k])) piv = i;
        for (int j = 0; j < m; ++j) std::swap(a[(size_t)k * m + j], a[(size_t)piv * m + j]);
        std::swap(b[k], b[piv]);
        for (int i = k + 1; i < m; ++i) {
            const double f = a[(size_t)i * m + k] / a[(size_t)k * m + k];
            if (f == 0.0) continue;
            for (int j = k; j < m; ++j) a[(size_t)i * m + j] -= f * a[(size_t)k * m + j];
            b[i] -= f * b[k];
        }
    }
    for (int k = m - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < m; ++j) s -= a[(size_t)k * m + j] * b[j];
        b[k] = s / a[(size_t)k * m + k];
    }
    return b;
}

// max |(P B Q)[k][l] - (L U)[k][l]| from the pull-form schedules: Lf rows of L (unit diagonal), Uf rows of U
static double residual(int m, const Cols& cols, const LUFactors& f) {
    const std::vector<double> a = dense_of(m, cols);
    std::vector<double> L((size_t)m * m, 0.0), U((size_t)m * m, 0.0);
    for (int k = 0; k < m; ++k) {
        L[(size_t)k * m + k] = 1.0;
        for (int e = f.Lf.ptr[k]; e < f.Lf.ptr[k + 1]; ++e) L[(size_t)k * m + f.Lf.idx[e]] = f.Lf.val[e];
        U[(size_t)k * m + k] = f.Uf.diag[k];
        for (int e = f.Uf.ptr[k]; e < f.Uf.ptr[k + 1]; ++e) U[(size_t)k * m + f.Uf.idx[e]] = f.Uf.val[e];
    }
    double worst = 0.0;
    for (int k = 0; k < m; ++k)
        for (int l = 0; l < m; ++l) {
            double s = 0.0;
            for (int t = 0; t < m; ++t) s += L[(size_t)k * m + t] * U[(size_t)t * m + l];
            worst = std::max(worst, std::fabs(s - a[(size_t)f.rowperm[k] * m + f.colperm[l]]));
        }
    return worst;
}

// the plain level-by-level solve (what relp_lu.cpp: solve_schedule_host does), optionally with pivots masked the way a
// Forrest-Tomlin update deletes row and column p of U: x[p] = 0
static void solve_plain(const TriangularSchedule& s, std::vector<double>& x, const std::vector<char>* masked = nullptr) {
    for (size_t l = 0; l + 1 < s.level_ptr.size(); ++l)
        for (int t = s.level_ptr[l]; t < s.level_ptr[l + 1]; ++t) {
            const int k = s.level_rows[t];
            double sum = x[k];
            for (int e = s.ptr[k]; e < s.ptr[k + 1]; ++e) sum -= s.val[e] * x[s.idx[e]];
            x[k] = (masked && (*masked)[k]) ? 0.0 : sum / s.diag[k];
        }
}

// the solve as the device runs it: right-hand side copied behind x, then pass by pass; every row = 2^lg lanes whose sum of
// -value * x[index] is multiplied by 1 / diagonal and stored by the first lane
// first_group: the hyper-sparse start -- passes of earlier groups are skipped (EllPacked::reach says when that is allowed)
static void solve_ell(const EllPacked& e, int m, int rhs_base, std::vector<double> x_in, std::vector<double>* out, int first_group = 0) {
    std::vector<double> x((size_t)2 * m + 2, 0.0);
    for (int i = 0; i < m; ++i) x[i] = x_in[i];
    if (e.rhs_src.empty()) for (int i = 0; i < m; ++i) x[rhs_base + i] = x_in[i];            // a copy per pivot (ell_stage)
    else for (size_t i = 0; i < e.rhs_src.size(); ++i) x[rhs_base + i] = x_in[e.rhs_src[i]];    // the compacted copies
    for (int k : e.triv) x[k] *= e.rdiag[k];                       // the rows without entries: one loop in front of the passes
    // both packings: 16-bit slots (index | lg << 13) and the 32-bit ones of large bases (index | lg << 24)
    const bool wide = e.sidx.empty() && !e.sidx32.empty();
    const int shift = wide ? kEllLgShiftWide : kEllLgShift;
    auto slot = [&](int at) { return wide ? (int)e.sidx32[at] : (int)e.sidx[at]; };
    for (const EllPassHost& ps : e.passes) {
        if (ps.level < first_group) continue;
        std::vector<std::pair<int, double>> stores;
        for (int lane = 0; lane < ps.lanes;) {
            const int iv = slot(ps.lane0 + lane), lg = iv >> shift, k = iv & ((1 << shift) - 1);
            double sum = 0.0;
            for (int j = 0; j < (1 << lg); ++j) {
                const int jv = slot(ps.lane0 + lane + j) & ((1 << shift) - 1);
                sum += -e.sval[ps.lane0 + lane + j] * x[jv];
            }
            if (!e.rovf.empty())
                for (int o = e.rovf[2 * k]; o < e.rovf[2 * k + 1]; ++o) sum += -e.oval[o] * x[wide ? (int)e.oidx32[o] : (int)e.oidx[o]];
            stores.emplace_back(k, sum * e.rdiag[k]);
            lane += 1 << lg;
        }
        for (auto& st : stores) x[st.first] = st.second;           // (the rows of a pass do not read each other's results)
    }
    out->assign(x.begin(), x.begin() + m);
}

static double max_diff(const std::vector<double>& a, const std::vector<double>& b) {
    double d = 0.0, n = 1.0;
    for (size_t i = 0; i < a.size(); ++i) { d = std::max(d, std::fabs(a[i] - b[i])); n = std::max(n, std::fabs(b[i])); }
    return d / n;
}

static void check_matrix(const char* name, int m, const Cols& cols_in, std::mt19937_64& rng) {
    Cols cols = cols_in;
    for (auto& c : cols) std::sort(c.begin(), c.end());
    LUFactors f; std::string err;
    const bool ok = lu_factor(m, cols, &f, &err);
    CHECK(ok, "%s: lu_factor failed: %s", name, err.c_str());
    if (!ok) return;
    const double res = residual(m, cols, f);
    CHECK(res <= 1e-12 * std::max(1.0, (double)m), "%s: |P B Q - L U| = %.3e", name, res);
    // FTRAN / BTRAN against the dense solve
    const std::vector<double> a = dense_of(m, cols);
    std::vector<double> rhs(m), x, z;
    for (auto& v : rhs) v = (double)((int)(rng() % 19) - 9);
    lu_ftran_host(f, rhs, &x);
    CHECK(max_diff(x, dense_solve(m, a, rhs, false)) <= 1e-9, "%s: FTRAN differs from the dense solve by %.3e", name,
          max_diff(x, dense_solve(m, a, rhs, false)));
    lu_btran_host(f, rhs, &z);
    CHECK(max_diff(z, dense_solve(m, a, rhs, true)) <= 1e-9, "%s: BTRAN differs from the dense solve by %.3e", name,
          max_diff(z, dense_solve(m, a, rhs, true)));
    // the four schedules: fused and packed, executed like the device does, against the plain solve
    const TriangularSchedule* sch[4] = {&f.Lf, &f.Uf, &f.Ub, &f.Lb};
    const char* nm[4] = {"L", "U", "U'", "L'"};
    for (int k = 0; k < 4; ++k) {
        const bool maskable = k == 1 || k == 2;
        for (int cap : {0, 64, 256}) {
            FusedSchedule fs;
            fuse_levels(*sch[k], maskable, maskable, cap, &fs);
            EllPacked e;
            ell_pack(fs, maskable, &e);
            std::vector<double> b(m), want, got;
            for (auto& v : b) v = (rng() % 3 == 0) ? (double)((int)(rng() % 13) - 6) : 0.0;
            want = b;
            solve_plain(*sch[k], want);
            solve_ell(e, m, fs.rhs_base, b, &got);
            CHECK(max_diff(got, want) <= 1e-9, "%s %s cap %d: packed solve differs by %.3e", name, nm[k], cap, max_diff(got, want));
            {   // hyper-sparse start: a right-hand side with a few entries, the sweep begins at the first group they reach
                std::vector<double> bs(m, 0.0), want_s, got_s;
                for (int t = 0; t < 3; ++t) bs[rng() % m] = (double)((int)(rng() % 9) + 1);
                int g0 = 0x7fffffff;
                for (int i = 0; i < m; ++i) if (bs[i] != 0.0) g0 = std::min(g0, e.reach[i]);
                want_s = bs;
                solve_plain(*sch[k], want_s);
                solve_ell(e, m, fs.rhs_base, bs, &got_s, g0);
                CHECK(max_diff(got_s, want_s) <= 1e-9, "%s %s cap %d: sweep started at group %d differs by %.3e", name, nm[k], cap, g0,
                      max_diff(got_s, want_s));
                for (int kk : e.triv) CHECK(sch[k]->ptr[kk + 1] == sch[k]->ptr[kk], "%s %s: row %d in the trivial list has entries", name, nm[k], kk);
            }
            {   // the 32-bit packing of large bases: same passes, same slots, same result bit for bit
                EllPacked w;
                ell_pack(fs, maskable, &w, true, true, 4);     // (+ compacted right-hand-side copies, rows without entries listed)
                std::vector<double> got_w;
                solve_ell(w, m, fs.rhs_base, b, &got_w);
                CHECK(w.sidx.empty() && w.overflow() == e.overflow() && w.passes.size() <= e.passes.size(),
                      "%s %s cap %d: wide packing has another shape", name, nm[k], cap);
                CHECK(max_diff(got_w, got) <= 1e-12, "%s %s cap %d: wide packing solves differently", name, nm[k], cap);
                std::vector<double> bs(m, 0.0), want_s, got_s;      // hyper-sparse start on the list / compact variant
                for (int t = 0; t < 3; ++t) bs[rng() % m] = (double)((int)(rng() % 9) + 1);
                int g0 = 0x7fffffff;
                for (int i = 0; i < m; ++i) if (bs[i] != 0.0) g0 = std::min(g0, w.reach[i]);
                want_s = bs;
                solve_plain(*sch[k], want_s);
                solve_ell(w, m, fs.rhs_base, bs, &got_s, g0);
                CHECK(max_diff(got_s, want_s) <= 1e-9, "%s %s cap %d: wide sweep started at group %d differs by %.3e", name, nm[k], cap, g0,
                      max_diff(got_s, want_s));
            }
            CHECK(fs.s.level_ptr.size() <= sch[k]->level_ptr.size(), "%s %s: more groups than levels", name, nm[k]);
            for (const EllPassHost& ps : e.passes) CHECK(ps.lanes <= 256 && ps.lanes > 0, "%s %s: pass of %d lanes", name, nm[k], ps.lanes);
            if (!maskable || m < 2) continue;
            // mask a few pivots the way ft_update does: 1 / diagonal := 0, the listed entries := 0, x[p] := 0 on entry
            std::vector<char> masked(m, 0);
            for (int t = 0; t < std::min(m, 6); ++t) {
                const int p = (int)(rng() % m);
                if (masked[p]) continue;
                masked[p] = 1;
                e.rdiag[p] = 0.0;
                for (int v = e.via_ptr[p]; v < e.via_ptr[p + 1]; ++v) { CHECK(e.via_pos[v] >= 0, "%s %s: via entry in an overflow list", name, nm[k]); if (e.via_pos[v] >= 0) e.sval[e.via_pos[v]] = 0.0; }
                std::vector<double> bm = b;
                for (int i = 0; i < m; ++i) if (masked[i]) bm[i] = 0.0;
                want = bm;
                solve_plain(*sch[k], want, &masked);
                solve_ell(e, m, fs.rhs_base, bm, &got);
                CHECK(max_diff(got, want) <= 1e-9, "%s %s cap %d: packed solve with %d masked pivots differs by %.3e", name, nm[k], cap,
                      t + 1, max_diff(got, want));
            }
        }
    }
}

int main() {
    std::mt19937_64 rng(20250104);
    // the reference's factorisation cases, column by column (decomposition/mod.rs:301-491 gives them row by row)
    check_matrix("identity 2", 2, {{{0, 1.0}}, {{1, 1.0}}}, rng);
    check_matrix("identity 3", 3, {{{0, 1.0}}, {{1, 1.0}}, {{2, 1.0}}}, rng);
    check_matrix("offdiagonal upper", 2, {{{0, 1.0}}, {{0, 1.0}, {1, 1.0}}}, rng);
    check_matrix("offdiagonal lower", 2, {{{0, 1.0}, {1, 1.0}}, {{1, 1.0}}}, rng);
    check_matrix("offdiagonal swapped", 2, {{{0, 1.0}, {1, 1.0}}, {{0, 1.0}}}, rng);
    check_matrix("wikipedia 1", 2, {{{0, 4.0}, {1, 6.0}}, {{0, 3.0}, {1, 3.0}}}, rng);
    check_matrix("wikipedia 2", 2, {{{0, -1.0}, {1, 1.0}}, {{0, 1.5}, {1, -1.0}}}, rng);
    {   // wikipedia 2's known answers (decomposition/mod.rs:470-489): B^-1 e_0 = (2, 2), B^-1 e_1 = (3, 2)
        Cols c = {{{0, -1.0}, {1, 1.0}}, {{0, 1.5}, {1, -1.0}}};
        LUFactors f; std::string err;
        CHECK(lu_factor(2, c, &f, &err), "wikipedia 2: %s", err.c_str());
        std::vector<double> x;
        lu_ftran_host(f, {1.0, 0.0}, &x);
        CHECK(std::fabs(x[0] - 2.0) < 1e-14 && std::fabs(x[1] - 2.0) < 1e-14, "wikipedia 2: B^-1 e_0 = (%g, %g)", x[0], x[1]);
        lu_ftran_host(f, {0.0, 1.0}, &x);
        CHECK(std::fabs(x[0] - 3.0) < 1e-14 && std::fabs(x[1] - 2.0) < 1e-14, "wikipedia 2: B^-1 e_1 = (%g, %g)", x[0], x[1]);
    }
    // the 5 x 5 upper factor of lower_upper/mod.rs:779-867 (Elble & Sahinidis) as a basis of its own
    check_matrix("elble-sahinidis U", 5, {{{0, 11.0}}, {{0, 12.0}, {1, 22.0}}, {{0, 13.0}, {1, 23.0}, {2, 33.0}},
                                          {{0, 14.0}, {1, 24.0}, {2, 34.0}, {3, 44.0}}, {{0, 15.0}, {1, 25.0}, {2, 35.0}, {3, 45.0}, {4, 55.0}}}, rng);
    // a singular matrix is reported, not factorised
    {
        LUFactors f; std::string err;
        Cols c = {{{0, 1.0}, {1, 2.0}}, {{0, 2.0}, {1, 4.0}}};
        CHECK(!lu_factor(2, c, &f, &err) && !err.empty(), "a singular matrix passed");
    }
    // LP-like random bases: a permuted diagonal, a sparse bump, a few dense columns, long dependency chains
    for (int trial = 0; trial < 60; ++trial) {
        const int m = 2 + (int)(rng() % (trial < 40 ? 40 : 260));
        Cols cols(m);
        for (int j = 0; j < m; ++j) {
            std::vector<int> rows_{(int)(((long long)j * 7 + 3) % m)};
            if (m % 7 == 0) rows_.push_back(j);
            const int extra = (int)(rng() % 4);
            for (int e = 0; e < extra; ++e) rows_.push_back((int)(rng() % m));
            if (rng() % 5 == 0 && j > 0) rows_.push_back((int)(((long long)(j - 1) * 7 + 3) % m));     // chains: deep level structure
            if (rng() % 50 == 0) for (int e = 0; e < 70 && e < m; ++e) rows_.push_back((int)(rng() % m));
            std::sort(rows_.begin(), rows_.end());
            rows_.erase(std::unique(rows_.begin(), rows_.end()), rows_.end());
            for (int r : rows_) { const int q = (int)(rng() % 9) - 4; cols[j].emplace_back(r, q == 0 ? 1.0 : (rng() % 3 == 0 ? q * 0.5 : (double)q)); }
        }
        LUFactors f; std::string err;
        if (!lu_factor(m, cols, &f, &err)) continue;                 // (a random matrix may be singular: nothing to check)
        char nm[64];
        std::snprintf(nm, sizeof nm, "random %d (m = %d)", trial, m);
        check_matrix(nm, m, cols, rng);
    }
    std::printf("test_lu_host: %d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
