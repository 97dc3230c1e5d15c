// The device-side LU factorisation (rust-lp_amd/csrc/relp_lu_factor_core.h) compiled for the HOST: the
// very code the kernel k_lu_factor runs, with its parallel loops as plain loops.  Checked on the reference's factorisation
// cases (decomposition/mod.rs:301-491) and on seeded LP-like bases -- with structural, slack, bound and artificial columns --
// against P B Q = L U, against dense solves and against the host factorisation lu_factor.  No GPU.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "relp_lu.hpp"
#include "relp_lu_factor_core.h"
#include "relp_lu_schedule_core.h"

using namespace relp;
using Cols = std::vector<std::vector<std::pair<int32_t, double>>>;

static int g_checks = 0, g_failed = 0;
#define CHECK(cond, ...) do { ++g_checks; if (!(cond)) { ++g_failed; std::printf("FAILED %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

// A provider in the engine's terms: structural columns (CSC over the constraint rows) with optional bound rows, virtual
// (slack) columns, artificial columns; and a basis = m column ids.
struct Provider {
    int32_t mc = 0, m = 0, na = 0, nn = 0, nv = 0;
    std::vector<int64_t> cptr; std::vector<int32_t> cidx; std::vector<double> cval;
    std::vector<int32_t> bound_row, vrow0, vrow1, vsign, column_to_row;
    std::vector<double> cost;
    // row-major copy of the provider columns
    std::vector<int32_t> rptr, rcol, art_of_row; std::vector<double> rval;
    std::vector<int32_t> wrapped_row;

    std::vector<std::pair<int32_t, double>> column_of(int32_t j) const {      // one column as (row, value) pairs
        std::vector<std::pair<int32_t, double>> c;
        if (j < na) { c.emplace_back(column_to_row[j], 1.0); return c; }
        const int32_t p = j - na;
        if (p < nn) {
            for (int64_t e = cptr[p]; e < cptr[p + 1]; ++e) c.emplace_back(cidx[e], cval[e]);
            if (bound_row[p] >= 0) c.emplace_back(bound_row[p], 1.0);
        } else {
            const int32_t v = p - nn;
            if (vrow0[v] >= 0) c.emplace_back(vrow0[v], (double)vsign[v]);
            if (vrow1[v] >= 0) c.emplace_back(vrow1[v], 1.0);
        }
        return c;
    }
    void finish() {
        std::vector<std::vector<std::pair<int32_t, double>>> rows(m);
        for (int32_t p = 0; p < nn + nv; ++p) { const auto col = column_of(na + p); for (auto& e : col) rows[e.first].emplace_back(p, e.second); }
        rptr.assign(m + 1, 0);
        for (int32_t i = 0; i < m; ++i) { rptr[i + 1] = rptr[i] + (int32_t)rows[i].size(); for (auto& e : rows[i]) { rcol.push_back(e.first); rval.push_back(e.second); } }
        art_of_row.assign(m, -1);
        for (int32_t a = 0; a < na; ++a) art_of_row[column_to_row[a]] = a;
        cost.assign(std::max(nn, 1), 0.0);
    }
    LufMatrix view() const {
        LufMatrix M{};
        M.m = m; M.na = na; M.n_provider = nn + nv;
        M.csc = DeviceCSC{cptr.data(), cidx.data(), cval.data()};
        M.ct = ColumnTable{na, nn, nv, mc, column_to_row.data(), bound_row.data(), vrow0.data(), vrow1.data(), vsign.data(), cost.data()};
        M.rptr = rptr.data(); M.rcol = rcol.data(); M.rval = rval.data(); M.art_of_row = art_of_row.data();
        M.wrapped_na = 0; M.wrapped_row = wrapped_row.data();
        return M;
    }
};

struct Buffers {
    std::vector<int32_t> ints; std::vector<double> dbl; std::vector<unsigned long long> red, u64;
    std::vector<double> dense, utv; std::vector<int32_t> dint, uti;
    LufWork W{}; LufOut O{};
    static int& dense_cap() { static int c = 64; return c; }       // (the tests run with the dense finish on and off)
    void setup(const Provider& P, int32_t nb_cap, int32_t cap, int32_t arena_cap = -1) {
        const int32_t m = P.m;
        if (arena_cap < 0) arena_cap = cap;
        auto need = (size_t)(P.nn + P.nv) + P.na + 16 * (size_t)m + 14 * (size_t)nb_cap + 512 + 8 * (size_t)(m + 1) + 4 * (size_t)cap + 64 + (size_t)arena_cap +
                    3 * (size_t)cap;
        ints.assign(need, 0); dbl.assign((size_t)nb_cap + m + 4 * (size_t)cap + 16 + (size_t)arena_cap + (size_t)cap, 0.0); red.assign(8, 0);
        u64.assign(4 * (size_t)nb_cap + 4, 0);
        int32_t* ip = ints.data(); double* dp = dbl.data();
        auto ti = [&](size_t n) { int32_t* r = ip; ip += n; return r; };
        auto td = [&](size_t n) { double* r = dp; dp += n; return r; };
        W.pos_p = ti(P.nn + P.nv + 1); W.pos_a = ti(P.na + 1); W.wrow_pos = ti(m); W.rcount = ti(m); W.ccount = ti(m);
        W.claim = ti(m); W.claim2 = ti(m); W.list = ti(m); W.list2 = ti(m); W.piv = ti(m); W.part = ti(66);
        W.brow = ti(m); W.bcol = ti(m); W.lrow = ti(m); W.lcol = ti(m);
        W.nb_cap = nb_cap;
        W.rbeg = ti(nb_cap); W.rlen = ti(nb_cap); W.rcap = ti(nb_cap); W.ract = ti(nb_cap); W.cact = ti(nb_cap); W.bcc = ti(nb_cap);
        W.bstep_row = ti(nb_cap); W.bstep_col = ti(nb_cap); W.cpiv = ti(nb_cap); W.prank = ti(nb_cap); W.acc = ti(nb_cap); W.pval = td(nb_cap);
        W.cmax = u64.data(); W.rowmark = u64.data() + nb_cap; W.colbest = u64.data() + 2 * (size_t)nb_cap; W.cprio = u64.data() + 3 * (size_t)nb_cap;
        W.ecol = ti(arena_cap); W.eval = td(arena_cap); W.arena_cap = arena_cap;
        W.lt_row = ti(cap); W.lt_step = ti(cap); W.lt_val = td(cap); W.lt_cap = cap; W.lt_ptr = ti(nb_cap + 1); W.lt_ord = ti(cap);
        W.counters = ti(128); W.red = red.data(); W.scalars = ti(16);
        const int dc = dense_cap();
        dense.assign((size_t)dc * dc + 1, 0.0); dint.assign(8 * (size_t)dc + 1, 0);
        W.dense = dense.data(); W.dint = dint.data(); W.dense_cap = dc;
        utv.assign((size_t)cap + 1, 0.0); uti.assign(3 * (size_t)cap + m + 4, 0);
        W.ut_row = uti.data(); W.ut_col = uti.data() + cap; W.ut_val = utv.data(); W.vtmp = uti.data() + 2 * (size_t)cap; W.vw = uti.data() + 3 * (size_t)cap;
        W.vtmp_cap = cap;
        O.status = ti(8); O.rowperm = ti(m); O.colperm = ti(m); O.row_step = ti(m); O.col_step = ti(m); O.diag = td(m);
        LufTriangle* tri[4] = {&O.Lf, &O.Uf, &O.Ub, &O.Lb};
        for (auto* t : tri) { t->ptr = ti(m + 1); t->idx = ti(cap); t->val = td(cap); }
        O.cap = cap;
    }
};

static std::vector<double> dense_basis(const Provider& P, const std::vector<int32_t>& basis) {
    std::vector<double> a((size_t)P.m * P.m, 0.0);
    for (int32_t c = 0; c < P.m; ++c) { const auto col = P.column_of(basis[c]); for (auto& e : col) a[(size_t)e.first * P.m + c] += e.second; }
    return a;
}

// x = B^-1 a through the device factors (forward with the rows of L, backward with the rows of U)
static std::vector<double> ftran(const LufOut& O, int32_t m, const std::vector<double>& a) {
    std::vector<double> y(m), x(m);
    for (int32_t k = 0; k < m; ++k) { double s = a[O.rowperm[k]]; for (int32_t e = O.Lf.ptr[k]; e < O.Lf.ptr[k + 1]; ++e) s -= O.Lf.val[e] * y[O.Lf.idx[e]]; y[k] = s; }
    for (int32_t k = m - 1; k >= 0; --k) { double s = y[k]; for (int32_t e = O.Uf.ptr[k]; e < O.Uf.ptr[k + 1]; ++e) s -= O.Uf.val[e] * y[O.Uf.idx[e]]; y[k] = s / O.diag[k]; }
    for (int32_t k = 0; k < m; ++k) x[O.colperm[k]] = y[k];
    return x;
}
// z' = c' B^-1 through the column views (U' forward, L' backward)
static std::vector<double> btran(const LufOut& O, int32_t m, const std::vector<double>& c) {
    std::vector<double> t(m), z(m);
    for (int32_t k = 0; k < m; ++k) { double s = c[O.colperm[k]]; for (int32_t e = O.Ub.ptr[k]; e < O.Ub.ptr[k + 1]; ++e) s -= O.Ub.val[e] * t[O.Ub.idx[e]]; t[k] = s / O.diag[k]; }
    for (int32_t k = m - 1; k >= 0; --k) { double s = t[k]; for (int32_t e = O.Lb.ptr[k]; e < O.Lb.ptr[k + 1]; ++e) s -= O.Lb.val[e] * t[O.Lb.idx[e]]; t[k] = s; }
    for (int32_t k = 0; k < m; ++k) z[O.rowperm[k]] = t[k];
    return z;
}

static std::vector<double> dense_solve(int m, std::vector<double> a, std::vector<double> b, bool transposed) {
    if (transposed) for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) std::swap(a[(size_t)i * m + j], a[(size_t)j * m + i]);
    for (int k = 0; k < m; ++k) {
        int p = k;
        for (int i = k + 1; i < m; ++i) if (std::fabs(a[(size_t)i * m + k]) > std::fabs(a[(size_t)p * m + k])) p = i;
        if (p != k) { for (int j = 0; j < m; ++j) std::swap(a[(size_t)k * m + j], a[(size_t)p * m + j]); std::swap(b[k], b[p]); }
        for (int i = k + 1; i < m; ++i) {
            const double f = a[(size_t)i * m + k] / a[(size_t)k * m + k];
            if (f == 0.0) continue;
            for (int j = k; j < m; ++j) a[(size_t)i * m + j] -= f * a[(size_t)k * m + j];
            b[i] -= f * b[k];
        }
    }
    for (int k = m - 1; k >= 0; --k) { double s = b[k]; for (int j = k + 1; j < m; ++j) s -= a[(size_t)k * m + j] * b[j]; b[k] = s / a[(size_t)k * m + k]; }
    return b;
}

static double max_rel(const std::vector<double>& a, const std::vector<double>& b) {
    double d = 0.0, n = 1.0;
    for (size_t i = 0; i < a.size(); ++i) { d = std::max(d, std::fabs(a[i] - b[i])); n = std::max(n, std::fabs(b[i])); }
    return d / n;
}

static void check_schedules(const char* name, const LufOut& O, const LufWork& W, int32_t m, std::mt19937_64& rng);

// factorise `basis` of provider P with the device code and check it every way there is
static void check(const char* name, const Provider& P, const std::vector<int32_t>& basis, std::mt19937_64& rng, bool expect_singular = false) {
    const int32_t m = P.m;
    Buffers B;
    size_t nnz = 0;
    for (int32_t c = 0; c < m; ++c) nnz += P.column_of(basis[c]).size();
    B.setup(P, m, (int32_t)(nnz + (size_t)m * m + 16));
    const LufMatrix M = P.view();
    luf_factor(M, basis.data(), B.W, B.O);
    Cols cols(m);
    for (int32_t c = 0; c < m; ++c) { cols[c] = P.column_of(basis[c]); std::sort(cols[c].begin(), cols[c].end()); }
    LUFactors hf; std::string err;
    const bool host_ok = lu_factor(m, cols, &hf, &err);
    if (expect_singular) { CHECK(B.O.status[0] == LUF_SINGULAR && !host_ok, "%s: a singular basis passed (status %d)", name, B.O.status[0]); return; }
    if (!host_ok) return;                                 // (a random basis may be singular: nothing to compare with)
    CHECK(B.O.status[0] == LUF_OK, "%s: status %d (bump %d, peeled %d)", name, B.O.status[0], B.O.status[1], B.O.status[2]);
    if (B.O.status[0] != LUF_OK) return;
    const LufOut& O = B.O;
    // permutations are permutations, the inverses agree
    std::vector<int> seen_r(m, 0), seen_c(m, 0);
    for (int32_t k = 0; k < m; ++k) { ++seen_r[O.rowperm[k]]; ++seen_c[O.colperm[k]]; CHECK(O.row_step[O.rowperm[k]] == k && O.col_step[O.colperm[k]] == k, "%s: step %d", name, k); }
    for (int32_t i = 0; i < m; ++i) CHECK(seen_r[i] == 1 && seen_c[i] == 1, "%s: not a permutation at %d", name, i);
    // P B Q = L U, entry by entry
    const std::vector<double> a = dense_basis(P, basis);
    std::vector<double> L((size_t)m * m, 0.0), U((size_t)m * m, 0.0);
    for (int32_t k = 0; k < m; ++k) {
        L[(size_t)k * m + k] = 1.0; U[(size_t)k * m + k] = O.diag[k];
        for (int32_t e = O.Lf.ptr[k]; e < O.Lf.ptr[k + 1]; ++e) { CHECK(O.Lf.idx[e] < k, "%s: L entry above the diagonal", name); L[(size_t)k * m + O.Lf.idx[e]] = O.Lf.val[e]; }
        for (int32_t e = O.Uf.ptr[k]; e < O.Uf.ptr[k + 1]; ++e) { CHECK(O.Uf.idx[e] > k, "%s: U entry below the diagonal", name); U[(size_t)k * m + O.Uf.idx[e]] = O.Uf.val[e]; }
    }
    double worst = 0.0, scale = 1.0;
    for (int32_t k = 0; k < m; ++k)
        for (int32_t l = 0; l < m; ++l) {
            double s = 0.0;
            for (int32_t q = 0; q <= std::min(k, l); ++q) s += L[(size_t)k * m + q] * U[(size_t)q * m + l];
            const double want = a[(size_t)O.rowperm[k] * m + O.colperm[l]];
            worst = std::max(worst, std::fabs(s - want)); scale = std::max(scale, std::fabs(want));
        }
    CHECK(worst <= 1e-12 * scale * std::max(1, m), "%s: |P B Q - L U| = %.3e", name, worst);
    // the column views hold the same entries as the row views
    {
        std::vector<double> Lc((size_t)m * m, 0.0), Uc((size_t)m * m, 0.0);
        for (int32_t l = 0; l < m; ++l) {
            for (int32_t e = O.Lb.ptr[l]; e < O.Lb.ptr[l + 1]; ++e) Lc[(size_t)O.Lb.idx[e] * m + l] = O.Lb.val[e];
            for (int32_t e = O.Ub.ptr[l]; e < O.Ub.ptr[l + 1]; ++e) Uc[(size_t)O.Ub.idx[e] * m + l] = O.Ub.val[e];
        }
        bool same = true;
        for (int32_t k = 0; k < m && same; ++k)
            for (int32_t l = 0; l < m; ++l) {
                if (k > l && Lc[(size_t)k * m + l] != L[(size_t)k * m + l]) same = false;
                if (k < l && Uc[(size_t)k * m + l] != U[(size_t)k * m + l]) same = false;
            }
        CHECK(same, "%s: row and column views differ", name);
        CHECK(O.Lf.ptr[m] == O.Lb.ptr[m] && O.Uf.ptr[m] == O.Ub.ptr[m] && O.status[3] == O.Lf.ptr[m] && O.status[4] == O.Uf.ptr[m], "%s: entry counts", name);
    }
    // FTRAN / BTRAN: against dense elimination and against the host factorisation
    std::vector<double> rhs(m), xh, zh;
    for (auto& v : rhs) v = (double)((int)(rng() % 19) - 9);
    const std::vector<double> x = ftran(O, m, rhs), z = btran(O, m, rhs);
    CHECK(max_rel(x, dense_solve(m, a, rhs, false)) <= 1e-9, "%s: FTRAN differs from the dense solve by %.3e", name, max_rel(x, dense_solve(m, a, rhs, false)));
    CHECK(max_rel(z, dense_solve(m, a, rhs, true)) <= 1e-9, "%s: BTRAN differs from the dense solve by %.3e", name, max_rel(z, dense_solve(m, a, rhs, true)));
    lu_ftran_host(hf, rhs, &xh); lu_btran_host(hf, rhs, &zh);
    CHECK(max_rel(x, xh) <= 1e-9 && max_rel(z, zh) <= 1e-9, "%s: solves differ from lu_factor's by %.3e / %.3e", name, max_rel(x, xh), max_rel(z, zh));
    check_schedules(name, O, B.W, m, rng);
    // no more fill than the host factorisation by a wide margin (both pivot for sparsity)
    CHECK(O.Lf.ptr[m] + O.Uf.ptr[m] <= 3 * (hf.nnz_l + hf.nnz_u) + 4 * m, "%s: %d + %d entries against the host's %lld + %lld", name, O.Lf.ptr[m], O.Uf.ptr[m],
          (long long)hf.nnz_l, (long long)hf.nnz_u);
}

// ---- the device-side schedule builder (relp_lu_schedule_core.h) on the factors above ------------------------------------------
static void plain_solve(const LufTriangle& T, const double* diag, bool ascending, int32_t m, std::vector<double>& x) {
    for (int32_t q = 0; q < m; ++q) {
        const int32_t k = ascending ? q : m - 1 - q;
        double s = x[k];
        for (int32_t e = T.ptr[k]; e < T.ptr[k + 1]; ++e) s -= T.val[e] * x[T.idx[e]];
        x[k] = diag ? s / diag[k] : s;
    }
}
// executes an image the way ell_solve_pp does: right-hand-side copies behind x, rows without entries first, then pass by pass from
// group `first`; rdiag / sval may have been edited by a mask (the caller's copies of the image)
static void image_solve(const char* img, const int32_t* desc, int32_t m, bool wide, const int32_t* triv, const int32_t* rhs_src, std::vector<double>& xm, int first) {
    const LufImageLayout L = luf_image_layout(m, desc[LUF_D_PASSES], desc[LUF_D_LEVELS], desc[LUF_D_LANES], desc[LUF_D_OVF], wide);
    const EllPass* passes = reinterpret_cast<const EllPass*>(img + L.passes);
    const double* rdiag = reinterpret_cast<const double*>(img + L.rdiag);
    const double* sval = reinterpret_cast<const double*>(img + L.sval);
    const double* oval = reinterpret_cast<const double*>(img + L.oval);
    const int32_t* rovf = reinterpret_cast<const int32_t*>(img + L.rovf);
    const int shift = wide ? kEllLgWide : kEllLg;
    auto sidx = [&](int64_t at) { return wide ? (int)reinterpret_cast<const uint32_t*>(img + L.sidx)[at] : (int)reinterpret_cast<const uint16_t*>(img + L.sidx)[at]; };
    auto oidx = [&](int64_t at) { return wide ? (int)reinterpret_cast<const uint32_t*>(img + L.oidx)[at] : (int)reinterpret_cast<const uint16_t*>(img + L.oidx)[at]; };
    std::vector<double> x((size_t)2 * m + 2, 0.0);
    for (int32_t i = 0; i < m; ++i) x[i] = xm[i];
    if (desc[LUF_D_USES_RHS]) {
        if (desc[LUF_D_NRHS] < 0) for (int32_t i = 0; i < m; ++i) x[m + 1 + i] = xm[i];                  // a copy per pivot
        else for (int32_t i = 0; i < desc[LUF_D_NRHS]; ++i) x[m + 1 + i] = xm[rhs_src[i]];               // the compacted copies
    }
    for (int32_t i = 0; i < desc[LUF_D_TRIV]; ++i) x[triv[i]] *= rdiag[triv[i]];
    for (int32_t p = 0; p < desc[LUF_D_PASSES]; ++p) {
        const EllPass& ps = passes[p];
        if (ps.level < first) continue;
        std::vector<std::pair<int, double>> stores;
        for (int lane = 0; lane < ps.lanes;) {
            const int iv = sidx(ps.lane0 + lane), lg = iv >> shift, k = iv & ((1 << shift) - 1);
            double sum = 0.0;
            for (int j = 0; j < (1 << lg); ++j) sum += -sval[ps.lane0 + lane + j] * x[sidx(ps.lane0 + lane + j) & ((1 << shift) - 1)];
            if (desc[LUF_D_OVF] > 0) for (int o = rovf[2 * k]; o < rovf[2 * k + 1]; ++o) sum += -oval[o] * x[oidx(o)];
            stores.emplace_back(k, sum * rdiag[k]);
            lane += 1 << lg;
        }
        for (auto& st : stores) x[st.first] = st.second;
    }
    for (int32_t i = 0; i < m; ++i) xm[i] = x[i];
}

struct SchedBuffers {
    std::vector<int32_t> wi, oi; std::vector<double> wd; std::vector<uint32_t> wb; std::vector<char> image;
    LufSchedWork S{}; LufSchedOut SO{};
    void setup(int32_t m, int32_t nnz) {
        const int32_t xcap = 4 * nnz + 2 * m + 64, pcap = 6 * nnz + 4 * m + 64;
        wi.assign(12 * (size_t)m + 6 * ((size_t)m + 3) + 3 * (size_t)xcap + pcap + 256, 0); wd.assign((size_t)xcap + 8, 0.0); wb.assign(2 * ((size_t)m / 32 + 3), 0u);
        int32_t* ip = wi.data();
        auto ti = [&](size_t n) { int32_t* r = ip; ip += n; return r; };
        S.indeg = ti(m); S.lev = ti(m); S.order = ti(m); S.nlev_cap = m + 1;
        S.lvl_ptr = ti(m + 3); S.lvl_grp = ti(m + 3); S.grp_lvl0 = ti(m + 3); S.grp_lane0 = ti(m + 3); S.grp_pass0 = ti(m + 3); S.grp_lanes = ti(m + 3);
        S.bits0 = wb.data(); S.bits1 = wb.data() + m / 32 + 3;
        S.grp = ti(m); S.xbeg = ti(m); S.xlen = ti(m);
        S.x_src = ti(xcap); S.x_coef = wd.data(); S.x_v0 = ti(xcap); S.x_vn = ti(xcap); S.x_cap = xcap; S.pool = ti(pcap); S.pool_cap = pcap;
        S.lg = ti(m); S.loff = ti(m); S.tmp = ti(m + 2); S.tmp2 = ti(m + 2); S.ovf_off = ti(m + 1); S.rhs_id = ti(m); S.sc = ti(32);
        const int64_t cap = 64 + 40 * ((int64_t)xcap + 2 * m + 64) + 16 * ((int64_t)m + 8);
        image.assign((size_t)cap, 0);
        oi.assign(6 * (size_t)m + LUF_D_WORDS + 2 + pcap + (size_t)m / 32 + 4, 0);
        int32_t* op = oi.data();
        auto to = [&](size_t n) { int32_t* r = op; op += n; return r; };
        SO.image = image.data(); SO.image_cap = cap; SO.desc = to(LUF_D_WORDS); SO.triv = to(m); SO.reach = to(m); SO.level_of = to(m);
        SO.via_ptr = to(m + 1); SO.via_pos = to(pcap); SO.via_cap = pcap; SO.rhs_src = to(m); SO.rhs_pos = to(m);
        SO.triv_bits = reinterpret_cast<uint32_t*>(to(m / 32 + 3));
    }
};

static void check_schedules(const char* name, const LufOut& O, const LufWork& W, int32_t m, std::mt19937_64& rng) {
    (void)W;
    const LufTriangle* tri[4] = {&O.Lf, &O.Uf, &O.Ub, &O.Lb};
    const LufTriangle* trt[4] = {&O.Lb, &O.Ub, &O.Uf, &O.Lf};                // the transposed pattern of each
    const bool asc[4] = {true, false, true, false};
    const char* nm[4] = {"L", "U", "U'", "L'"};
    for (int q = 0; q < 4; ++q)
        for (int variant = 0; variant < 5; ++variant) {
            // 0: 16-bit slots, unfused; 1: 16-bit, fused (a copy per pivot); 2: wide, unfused; 3: wide, fused, compacted copies, rows
            // without entries listed; 4: wide, fused with a small lane budget
            const bool wide = variant >= 2, maskable = q == 1 || q == 2;
            const int fuse = variant == 1 || variant == 3 ? 256 : variant == 4 ? 64 : 0;
            if (!wide && 2 * m + 2 >= (1 << kEllLg)) continue;
            LufSchedIn T{m, tri[q]->ptr, tri[q]->idx, tri[q]->val, trt[q]->ptr, trt[q]->idx, maskable ? O.diag : nullptr, maskable ? 1 : 0, wide ? 1 : 0,
                         variant == 3 ? 3 : 0x7fffffff, fuse, m, 1};
            SchedBuffers SB;
            SB.setup(m, tri[q]->ptr[m]);
            luf_build_schedule(T, SB.S, SB.SO);
            const LufSchedOut& SO = SB.SO;
            CHECK(SO.desc[LUF_D_STATUS] == LUF_OK, "%s %s variant %d: schedule status %d", name, nm[q], variant, SO.desc[LUF_D_STATUS]);
            if (SO.desc[LUF_D_STATUS] != LUF_OK) continue;
            std::vector<double> b(m), want, got;
            for (auto& v : b) v = (rng() % 3 == 0) ? (double)((int)(rng() % 13) - 6) : 0.0;
            want = b; got = b;
            plain_solve(*tri[q], T.diag, asc[q], m, want);
            image_solve(SB.image.data(), SO.desc, m, wide, SO.triv, SO.rhs_src, got, 0);
            CHECK(max_rel(got, want) <= 1e-9, "%s %s variant %d: image solve differs by %.3e", name, nm[q], variant, max_rel(got, want));
            CHECK(SO.desc[LUF_D_LEVELS] <= SO.desc[LUF_D_KAHN_LEVELS], "%s %s variant %d: more groups than levels", name, nm[q], variant);
            // hyper-sparse start from the reach array
            std::vector<double> bs(m, 0.0);
            for (int t = 0; t < 3; ++t) bs[rng() % m] = (double)((int)(rng() % 9) + 1);
            int g0 = 0x7fffffff;
            for (int32_t i = 0; i < m; ++i) if (bs[i] != 0.0) g0 = std::min(g0, SO.reach[i]);
            want = bs; got = bs;
            plain_solve(*tri[q], T.diag, asc[q], m, want);
            image_solve(SB.image.data(), SO.desc, m, wide, SO.triv, SO.rhs_src, got, g0);
            CHECK(max_rel(got, want) <= 1e-9, "%s %s variant %d: sweep from group %d differs by %.3e", name, nm[q], variant, g0, max_rel(got, want));
            // level_of (start_after of fuse_levels): a right-hand side that is zero on p and on everything solved no later than p lets
            // the sweep start at group level_of[p] + 1
            {
                std::vector<int> lev(m, 0);
                for (int32_t qq = 0; qq < m; ++qq) {
                    const int32_t k = asc[q] ? qq : m - 1 - qq;
                    int lv = 0;
                    for (int32_t e = tri[q]->ptr[k]; e < tri[q]->ptr[k + 1]; ++e) lv = std::max(lv, lev[tri[q]->idx[e]] + 1);
                    lev[k] = lv;
                }
                const int32_t p = (int32_t)(rng() % m);
                std::vector<double> bz = b;
                for (int32_t k = 0; k < m; ++k) if (lev[k] <= lev[p]) bz[k] = 0.0;
                want = bz; got = bz;
                plain_solve(*tri[q], T.diag, asc[q], m, want);
                image_solve(SB.image.data(), SO.desc, m, wide, SO.triv, SO.rhs_src, got, SO.level_of[p] + 1);
                CHECK(max_rel(got, want) <= 1e-9, "%s %s variant %d: sweep behind pivot %d (group %d) differs by %.3e", name, nm[q], variant, p, SO.level_of[p] + 1,
                      max_rel(got, want));
            }
            if (!maskable || m < 2) continue;
            // mask a few pivots the way ft_update does: 1 / diagonal := 0, the listed slots := 0, x[p] := 0 on entry
            const LufImageLayout L = luf_image_layout(m, SO.desc[LUF_D_PASSES], SO.desc[LUF_D_LEVELS], SO.desc[LUF_D_LANES], SO.desc[LUF_D_OVF], wide);
            double* rdiag = reinterpret_cast<double*>(SB.image.data() + L.rdiag);
            double* sval = reinterpret_cast<double*>(SB.image.data() + L.sval);
            std::vector<char> masked(m, 0);
            for (int t = 0; t < std::min(m, 6); ++t) {
                const int p = (int)(rng() % m);
                if (masked[p]) continue;
                masked[p] = 1;
                rdiag[p] = 0.0;
                for (int v = SO.via_ptr[p]; v < SO.via_ptr[p + 1]; ++v) sval[SO.via_pos[v]] = 0.0;
                std::vector<double> bm = b;
                for (int i = 0; i < m; ++i) if (masked[i]) bm[i] = 0.0;
                want = bm; got = bm;
                // the plain solve with the pivots masked: x[p] = 0
                for (int32_t qq = 0; qq < m; ++qq) {
                    const int32_t k = asc[q] ? qq : m - 1 - qq;
                    double s2 = want[k];
                    for (int32_t e = tri[q]->ptr[k]; e < tri[q]->ptr[k + 1]; ++e) s2 -= tri[q]->val[e] * want[tri[q]->idx[e]];
                    want[k] = masked[k] ? 0.0 : s2 / T.diag[k];
                }
                image_solve(SB.image.data(), SO.desc, m, wide, SO.triv, SO.rhs_src, got, 0);
                CHECK(max_rel(got, want) <= 1e-9, "%s %s variant %d: image solve with %d masked pivots differs by %.3e", name, nm[q], variant, t + 1, max_rel(got, want));
            }
        }
}

// a provider that is just the given square matrix as structural columns; basis = all of them
static Provider square(int m, const Cols& cols) {
    Provider P;
    P.mc = P.m = m; P.nn = m;
    P.cptr.assign(1, 0);
    for (auto c : cols) { std::sort(c.begin(), c.end()); for (auto& e : c) { P.cidx.push_back(e.first); P.cval.push_back(e.second); } P.cptr.push_back((int64_t)P.cidx.size()); }
    P.bound_row.assign(m, -1);
    P.finish();
    return P;
}
static void check_square(const char* name, int m, const Cols& cols, std::mt19937_64& rng, bool singular = false) {
    const Provider P = square(m, cols);
    std::vector<int32_t> basis(m);
    for (int c = 0; c < m; ++c) basis[c] = c;
    check(name, P, basis, rng, singular);
}

// optional: `test_lu_device_model BASIS.txt` (a RELP_DUMP_BASIS text dump) -- bump, rounds and fill of the device algorithm beside
// the host factorisation's, for DESIGN.md
static int probe(const char* path) {
    FILE* f = std::fopen(path, "r");
    if (!f) return 2;
    int m;
    if (std::fscanf(f, "%d", &m) != 1) return 2;
    Cols cols(m);
    size_t nnz = 0;
    for (auto& c : cols) {
        int n; if (std::fscanf(f, "%d", &n) != 1) return 2;
        c.resize(n);
        for (auto& e : c) if (std::fscanf(f, "%d %lf", &e.first, &e.second) != 2) return 2;
        std::sort(c.begin(), c.end());
        nnz += c.size();
    }
    std::fclose(f);
    const Provider P = square(m, cols);
    std::vector<int32_t> basis(m);
    for (int c = 0; c < m; ++c) basis[c] = c;
    Buffers B;
    B.setup(P, m, (int32_t)(8 * nnz + 16 * (size_t)m));
    luf_factor(P.view(), basis.data(), B.W, B.O);
    LUFactors hf; std::string err;
    const bool ok = lu_factor(m, cols, &hf, &err);
    std::printf("m %d, nnz(B) %zu: device status %d, bump %d, peeled %d, rounds %d (dense finish on %d rows, arena top %d), nnz(L) %d, nnz(U) %d (off-diagonal)  |  host %s: nnz(L) %lld, nnz(U) %lld\n",
                m, nnz, B.O.status[0], B.O.status[1], B.O.status[2], B.W.counters[2], B.W.counters[4], B.W.counters[0], B.O.status[3], B.O.status[4], ok ? "ok" : err.c_str(),
                (long long)hf.nnz_l, (long long)hf.nnz_u - m);
    auto levels = [&](const LufTriangle& T, bool asc) {
        std::vector<int> lev(m, 0); int nl = 0;
        for (int q = 0; q < m; ++q) { const int k = asc ? q : m - 1 - q; int l = 0; for (int e = T.ptr[k]; e < T.ptr[k + 1]; ++e) l = std::max(l, lev[T.idx[e]] + 1); lev[k] = l; nl = std::max(nl, l + 1); }
        return nl;
    };
    std::printf("levels device: L %d, U %d, U' %d, L' %d  |  host: L %zu, U %zu, U' %zu, L' %zu\n", levels(B.O.Lf, true), levels(B.O.Uf, false),
                levels(B.O.Ub, true), levels(B.O.Lb, false), hf.Lf.level_ptr.size() - 1, hf.Uf.level_ptr.size() - 1, hf.Ub.level_ptr.size() - 1, hf.Lb.level_ptr.size() - 1);
    // passes per sweep: the device's factors through the device's fusion, the host's through fuse_levels / ell_pack
    const LufTriangle* tri[4] = {&B.O.Lf, &B.O.Uf, &B.O.Ub, &B.O.Lb};
    const LufTriangle* trt[4] = {&B.O.Lb, &B.O.Ub, &B.O.Uf, &B.O.Lf};
    const TriangularSchedule* hs[4] = {&hf.Lf, &hf.Uf, &hf.Ub, &hf.Lb};
    const char* nm[4] = {"L", "U", "U'", "L'"};
    int dev_total = 0, host_total = 0;
    for (int q = 0; q < 4; ++q) {
        const bool maskable = q == 1 || q == 2;
        LufSchedIn T{m, tri[q]->ptr, tri[q]->idx, tri[q]->val, trt[q]->ptr, trt[q]->idx, maskable ? B.O.diag : nullptr, maskable ? 1 : 0, 0, 0x7fffffff, 256, m, 0};
        SchedBuffers SB;
        SB.setup(m, tri[q]->ptr[m]);
#if defined(LUF_COUNT)
        g_luf_count[0] = g_luf_count[1] = g_luf_count[2] = 0;
#endif
        luf_build_schedule(T, SB.S, SB.SO);
#if defined(LUF_COUNT)
        std::printf("   %s: term-search steps %lld, levels %lld, rows visited %lld\n", nm[q], g_luf_count[0], g_luf_count[1], g_luf_count[2]);
#endif
        FusedSchedule fs; EllPacked e;
        fuse_levels(*hs[q], maskable, maskable, 256, &fs);
        ell_pack(fs, maskable, &e);
        std::printf("%s: device %d levels -> %d groups, %d passes, %d lanes, image %d bytes, via %d (status %d)  |  host %zu levels -> %zu groups, %zu passes, %zu lanes\n",
                    nm[q], SB.SO.desc[LUF_D_KAHN_LEVELS], SB.SO.desc[LUF_D_LEVELS], SB.SO.desc[LUF_D_PASSES], SB.SO.desc[LUF_D_LANES], SB.SO.desc[LUF_D_BYTES],
                    SB.SO.desc[LUF_D_VIA], SB.SO.desc[LUF_D_STATUS], hs[q]->level_ptr.size() - 1, fs.s.level_ptr.size() - 1, e.passes.size(), e.lanes());
        dev_total += SB.SO.desc[LUF_D_PASSES]; host_total += (int)e.passes.size();
    }
    std::printf("passes of the four sweeps: device %d, host %d\n", dev_total, host_total);
    return 0;
}

static void all_checks() {
    std::mt19937_64 rng(20250611);
    // the reference's factorisation cases (decomposition/mod.rs:301-491), column by column
    check_square("identity 2", 2, {{{0, 1.0}}, {{1, 1.0}}}, rng);
    check_square("identity 3", 3, {{{0, 1.0}}, {{1, 1.0}}, {{2, 1.0}}}, rng);
    check_square("offdiagonal upper", 2, {{{0, 1.0}}, {{0, 1.0}, {1, 1.0}}}, rng);
    check_square("offdiagonal lower", 2, {{{0, 1.0}, {1, 1.0}}, {{1, 1.0}}}, rng);
    check_square("offdiagonal swapped", 2, {{{0, 1.0}, {1, 1.0}}, {{0, 1.0}}}, rng);
    check_square("wikipedia 1", 2, {{{0, 4.0}, {1, 6.0}}, {{0, 3.0}, {1, 3.0}}}, rng);
    check_square("wikipedia 2", 2, {{{0, -1.0}, {1, 1.0}}, {{0, 1.5}, {1, -1.0}}}, rng);
    check_square("elble-sahinidis U", 5, {{{0, 11.0}}, {{0, 12.0}, {1, 22.0}}, {{0, 13.0}, {1, 23.0}, {2, 33.0}},
                                          {{0, 14.0}, {1, 24.0}, {2, 34.0}, {3, 44.0}}, {{0, 15.0}, {1, 25.0}, {2, 35.0}, {3, 45.0}, {4, 55.0}}}, rng);
    check_square("dense 3", 3, {{{0, 2.0}, {1, 1.0}, {2, 4.0}}, {{0, 1.0}, {1, 3.0}, {2, 1.0}}, {{0, 5.0}, {1, 1.0}, {2, 2.0}}}, rng);
    check_square("singular", 2, {{{0, 1.0}, {1, 2.0}}, {{0, 2.0}, {1, 4.0}}}, rng, true);
    // LP-like random bases (the generator of test_lu_host.cpp): a permuted diagonal, a sparse bump, dense columns, chains
    for (int trial = 0; trial < 80; ++trial) {
        const int m = 2 + (int)(rng() % (trial < 50 ? 40 : 300));
        Cols cols(m);
        for (int j = 0; j < m; ++j) {
            std::vector<int> rows_{(int)(((long long)j * 7 + 3) % m)};
            if (m % 7 == 0) rows_.push_back(j);
            const int extra = (int)(rng() % 4);
            for (int e = 0; e < extra; ++e) rows_.push_back((int)(rng() % m));
            if (rng() % 5 == 0 && j > 0) rows_.push_back((int)(((long long)(j - 1) * 7 + 3) % m));
            if (rng() % 50 == 0) for (int e = 0; e < 70 && e < m; ++e) rows_.push_back((int)(rng() % m));
            std::sort(rows_.begin(), rows_.end());
            rows_.erase(std::unique(rows_.begin(), rows_.end()), rows_.end());
            for (int r : rows_) { const int q = (int)(rng() % 9) - 4; cols[j].emplace_back(r, q == 0 ? 1.0 : (rng() % 3 == 0 ? q * 0.5 : (double)q)); }
        }
        char nm[64];
        std::snprintf(nm, sizeof nm, "random %d (m = %d)", trial, m);
        check_square(nm, m, cols, rng);
    }
    // providers with every kind of column: mc constraint rows, bound rows, slacks of both signs, artificial columns; random bases
    for (int trial = 0; trial < 60; ++trial) {
        Provider P;
        P.mc = 3 + (int)(rng() % 30);
        P.nn = 2 + (int)(rng() % 40);
        std::vector<int32_t> bounded;
        P.bound_row.assign(P.nn, -1);
        for (int p = 0; p < P.nn; ++p) if (rng() % 3 == 0) { P.bound_row[p] = P.mc + (int)bounded.size(); bounded.push_back(p); }
        P.m = P.mc + (int)bounded.size();
        P.cptr.assign(1, 0);
        for (int p = 0; p < P.nn; ++p) {
            std::vector<int> rows_;
            const int n = 1 + (int)(rng() % 4);
            for (int e = 0; e < n; ++e) rows_.push_back((int)(rng() % P.mc));
            std::sort(rows_.begin(), rows_.end()); rows_.erase(std::unique(rows_.begin(), rows_.end()), rows_.end());
            for (int r : rows_) { P.cidx.push_back(r); P.cval.push_back((double)((int)(rng() % 7) + 1) * (rng() % 2 ? 1.0 : -0.5)); }
            P.cptr.push_back((int64_t)P.cidx.size());
        }
        // one slack per constraint row (sign by row), one bound slack per bound row
        for (int i = 0; i < P.mc; ++i) { P.vrow0.push_back(i); P.vrow1.push_back(-1); P.vsign.push_back(i % 3 == 0 ? -1 : 1); }
        for (size_t k = 0; k < bounded.size(); ++k) { P.vrow0.push_back(P.mc + (int)k); P.vrow1.push_back(-1); P.vsign.push_back(1); }
        P.nv = (int)P.vrow0.size();
        // artificial columns on the rows whose slack is negative
        for (int i = 0; i < P.mc; ++i) if (i % 3 == 0) P.column_to_row.push_back(i);
        P.na = (int)P.column_to_row.size();
        P.finish();
        // a basis: start from slacks / artificials (nonsingular), then swap in structural columns while it stays nonsingular
        std::vector<int32_t> basis(P.m);
        {
            int a = 0;
            for (int i = 0; i < P.mc; ++i) basis[i] = (i % 3 == 0) ? a++ : P.na + P.nn + i;
            for (size_t k = 0; k < bounded.size(); ++k) basis[P.mc + k] = P.na + P.nn + P.mc + (int)k;
        }
        for (int swaps = 0; swaps < P.m; ++swaps) {
            const int pos = (int)(rng() % P.m), cand = P.na + (int)(rng() % P.nn);
            if (std::find(basis.begin(), basis.end(), cand) != basis.end()) continue;
            std::vector<int32_t> trial_basis = basis;
            trial_basis[pos] = cand;
            Cols cols(P.m);
            for (int c = 0; c < P.m; ++c) { cols[c] = P.column_of(trial_basis[c]); std::sort(cols[c].begin(), cols[c].end()); }
            LUFactors f; std::string err;
            if (lu_factor(P.m, cols, &f, &err)) basis = trial_basis;
        }
        char nm[64];
        std::snprintf(nm, sizeof nm, "provider %d (m = %d, %d structural, %d artificial)", trial, P.m, P.nn, P.na);
        check(nm, P, basis, rng);
    }
}

int main(int argc, char** argv) {
    if (argc > 2) Buffers::dense_cap() = std::atoi(argv[2]);
    if (argc > 1) return probe(argv[1]);
    for (int dc : {64, 0, 7}) { Buffers::dense_cap() = dc; all_checks(); }     // the dense finish at the kernel's size, off, tiny
    std::printf("test_lu_device_model: %d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
