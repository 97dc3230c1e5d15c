// relp_lu.cpp -- host-side sparse LU (P B Q = L U) with Markowitz-style pivoting and the level
// schedules for the device triangular solves.  See relp_lu.hpp.
#include "relp_lu.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <queue>
#include <tuple>

namespace relp {

namespace {

struct Entry { int32_t row, pos; double val; };   // L entry: (original row, pivot step of the column, factor)

// rows[k] = list of (dependency, value); `ascending`: dependencies have smaller indices (solve 0..m-1),
// otherwise larger (solve m-1..0).
void build_schedule(int32_t m, const std::vector<std::vector<std::pair<int32_t, double>>>& rows,
                    const std::vector<double>& diag, bool ascending, TriangularSchedule* s) {
    s->ptr.assign(m + 1, 0);
    for (int32_t k = 0; k < m; ++k) s->ptr[k + 1] = s->ptr[k] + (int32_t)rows[k].size();
    s->idx.resize(s->ptr[m]);
    s->val.resize(s->ptr[m]);
    for (int32_t k = 0; k < m; ++k) {
        int32_t o = s->ptr[k];
        for (auto& e : rows[k]) { s->idx[o] = e.first; s->val[o] = e.second; ++o; }
    }
    s->diag = diag;
    std::vector<int32_t> lev(m, 0);
    int32_t nlev = 0;
    auto visit = [&](int32_t k) {
        int32_t l = 0;
        for (auto& e : rows[k]) l = std::max(l, lev[e.first] + 1);
        lev[k] = l;
        nlev = std::max(nlev, l + 1);
    };
    if (ascending) for (int32_t k = 0; k < m; ++k) visit(k);
    else for (int32_t k = m - 1; k >= 0; --k) visit(k);
    s->level_ptr.assign(nlev + 1, 0);
    for (int32_t k = 0; k < m; ++k) ++s->level_ptr[lev[k] + 1];
    for (int32_t l = 0; l < nlev; ++l) s->level_ptr[l + 1] += s->level_ptr[l];
    s->level_rows.resize(m);
    std::vector<int32_t> fill(s->level_ptr.begin(), s->level_ptr.end() - 1);
    for (int32_t k = 0; k < m; ++k) s->level_rows[fill[lev[k]]++] = k;
}

}  // namespace

bool lu_factor(int32_t m, const std::vector<std::vector<std::pair<int32_t, double>>>& columns, LUFactors* out,
               std::string* err) {
    std::vector<std::vector<std::pair<int32_t, double>>> rows(m);       // active submatrix, row major: (column, value)
    std::vector<std::vector<int32_t>> colrows(m);                       // rows that (may) hold an entry of the column
    std::vector<int32_t> ccount(m, 0);
    for (int32_t j = 0; j < m; ++j)
        for (auto& e : columns[j]) {
            if (e.second == 0.0) continue;
            rows[e.first].emplace_back(j, e.second);
            colrows[j].push_back(e.first);
            ++ccount[j];
        }
    std::vector<char> row_done(m, 0);
    std::vector<int32_t> step_of_row(m, -1), step_of_col(m, -1);
    out->m = m;
    out->rowperm.assign(m, 0);
    out->colperm.assign(m, 0);
    std::vector<std::vector<std::pair<int32_t, double>>> urows(m);      // U row of step k: (original column, value)
    std::vector<double> udiag(m, 1.0);
    std::vector<Entry> lent;
    std::vector<int32_t> mark(m, -1), pos(m, 0);
    int32_t stamp = 0;
    // Singletons first: most of an LP basis is triangular (slacks, bound rows).  A column with one active
    // entry is a fill-free pivot that always passes the threshold; a row with one active entry is fill-free
    // and is taken when it passes it.  Both are served from stacks, so the O(m) searches below only run on
    // the "bump" that is left.
    std::vector<int32_t> col_single, row_single;
    for (int32_t j = 0; j < m; ++j) if (ccount[j] == 1) col_single.push_back(j);
    for (int32_t i = 0; i < m; ++i) if (rows[i].size() == 1) row_single.push_back(i);
    // lazy min-heaps of (count, index) for the sparsest active row / column of the bump: an entry is stale
    // when the line is done or its count has changed since it was pushed (every change pushes a fresh one)
    using Item = std::pair<int32_t, int32_t>;
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> row_heap, col_heap;
    for (int32_t i = 0; i < m; ++i) row_heap.emplace((int32_t)rows[i].size(), i);
    for (int32_t j = 0; j < m; ++j) col_heap.emplace(ccount[j], j);
    // largest active |entry| of column j; the scan also drops the stale members of colrows[j] (finished rows,
    // rows that lost the entry, duplicates), so the lists stay as short as the columns are
    std::vector<int32_t> seen_stamp(m, -1);
    int32_t seen_tick = 0;
    auto col_max_of = [&](int32_t j) {
        double mx = 0.0;
        ++seen_tick;
        size_t o = 0;
        auto& list = colrows[j];
        for (size_t t = 0; t < list.size(); ++t) {
            const int32_t i = list[t];
            if (row_done[i] || seen_stamp[i] == seen_tick) continue;
            for (auto& e : rows[i]) if (e.first == j) { mx = std::max(mx, std::fabs(e.second)); seen_stamp[i] = seen_tick; list[o++] = i; break; }
        }
        list.resize(o);
        return mx;
    };

    for (int32_t k = 0; k < m; ++k) {
        int32_t spi = -1, spj = -1; double spv = 0.0;
        while (!col_single.empty() && spj < 0) {
            const int32_t j = col_single.back(); col_single.pop_back();
            if (step_of_col[j] >= 0 || ccount[j] != 1) continue;
            for (int32_t i : colrows[j]) {
                if (row_done[i]) continue;
                for (auto& e : rows[i]) if (e.first == j && e.second != 0.0) { spi = i; spj = j; spv = e.second; break; }
                if (spj >= 0) break;
            }
        }
        while (!row_single.empty() && spj < 0) {
            const int32_t i = row_single.back(); row_single.pop_back();
            if (row_done[i] || rows[i].size() != 1) continue;
            const int32_t j = rows[i][0].first; const double v = rows[i][0].second;
            if (v != 0.0 && std::fabs(v) >= 0.1 * col_max_of(j)) { spi = i; spj = j; spv = v; }
        }
        // Markowitz search restricted to the sparsest active row and the sparsest active column
        // (pivoting.rs:45-81 searches every remaining entry): candidate A = the entry of the sparsest row
        // with the lowest column count, candidate B = the entry of the sparsest column with the lowest row
        // count; the lower (r - 1)(c - 1) wins.
        int32_t ra = -1;
        if (spj < 0) {
            while (!row_heap.empty() && (row_done[row_heap.top().second] ||
                                         (int32_t)rows[row_heap.top().second].size() != row_heap.top().first)) row_heap.pop();
            if (row_heap.empty()) { if (err) *err = "singular basis (no active row)"; return false; }
            ra = row_heap.top().second;
            if (rows[ra].empty()) { if (err) *err = "singular basis (empty row during LU)"; return false; }
        }
        // threshold partial pivoting: a pivot must be at least kThreshold of the largest active entry of
        // its column, which bounds every multiplier of L by 1 / kThreshold
        auto& col_max = col_max_of;
        constexpr double kThreshold = 0.1;
        int32_t pi = spi, pj = spj; double pv = spv; int64_t best = spj >= 0 ? 0 : -1;
        if (spj < 0) {
            // candidate A: entries of the sparsest row, by ascending column count
            std::vector<std::pair<int32_t, int32_t>> order;          // (column count, column)
            for (auto& e : rows[ra]) if (e.second != 0.0) order.emplace_back(ccount[e.first], e.first);
            std::sort(order.begin(), order.end());
            for (auto& oc : order) {
                double v = 0.0;
                for (auto& e : rows[ra]) if (e.first == oc.second) { v = e.second; break; }
                if (std::fabs(v) < kThreshold * col_max(oc.second)) continue;
                pi = ra; pj = oc.second; pv = v;
                best = (int64_t)(rows[ra].size() - 1) * (ccount[pj] - 1);
                break;
            }
        }
        if (best != 0) {
            int32_t cb = -1;
            while (!col_heap.empty() && (step_of_col[col_heap.top().second] >= 0 || ccount[col_heap.top().second] != col_heap.top().first ||
                                         col_heap.top().first <= 0)) col_heap.pop();
            if (!col_heap.empty()) cb = col_heap.top().second;
            if (cb >= 0) {
                const double cmax = col_max(cb);
                for (int32_t i : colrows[cb]) {
                    if (row_done[i]) continue;
                    double v = 0.0; bool has = false;
                    for (auto& e : rows[i]) if (e.first == cb) { v = e.second; has = true; break; }
                    if (!has || v == 0.0 || std::fabs(v) < kThreshold * cmax) continue;
                    const int64_t cost = (int64_t)(rows[i].size() - 1) * (ccount[cb] - 1);
                    if (best < 0 || cost < best) { best = cost; pi = i; pj = cb; pv = v; }
                }
            }
        }
        if (pj < 0 && ra >= 0) {
            // no entry of the sparsest row / column passes the threshold: take the largest entry of the
            // sparsest row's best column (always acceptable)
            double bestv = 0.0;
            for (auto& e : rows[ra]) {
                const int32_t j = e.first;
                for (int32_t i : colrows[j]) {
                    if (row_done[i]) continue;
                    for (auto& f : rows[i]) if (f.first == j && std::fabs(f.second) > bestv) { bestv = std::fabs(f.second); pi = i; pj = j; pv = f.second; }
                }
                if (pj >= 0) break;
            }
        }
        if (pj < 0) { if (err) *err = "singular basis (no acceptable pivot)"; return false; }
        out->rowperm[k] = pi; out->colperm[k] = pj;
        step_of_row[pi] = k; step_of_col[pj] = k;
        row_done[pi] = 1;
        udiag[k] = pv;
        for (auto& e : rows[pi]) {
            if (--ccount[e.first] == 1) col_single.push_back(e.first);
            col_heap.emplace(ccount[e.first], e.first);
            if (e.first != pj) urows[k].push_back(e);
        }
        const std::vector<std::pair<int32_t, double>>& prow = urows[k];
        // eliminate column pj from the other active rows
        for (int32_t i : colrows[pj]) {
            if (row_done[i]) continue;
            auto& ri = rows[i];
            double vij = 0.0; bool has = false;
            for (auto& e : ri) if (e.first == pj) { vij = e.second; has = true; break; }
            if (!has) continue;                                   // stale list entry
            const double f = vij / pv;
            lent.push_back(Entry{i, k, f});
            ++stamp;
            std::vector<std::pair<int32_t, double>> nr;
            nr.reserve(ri.size() + prow.size());
            for (auto& e : ri) {
                if (e.first == pj) continue;
                mark[e.first] = stamp; pos[e.first] = (int32_t)nr.size();
                nr.push_back(e);
            }
            for (auto& e : prow) {
                if (mark[e.first] == stamp) nr[pos[e.first]].second -= f * e.second;
                else {
                    mark[e.first] = stamp; pos[e.first] = (int32_t)nr.size();
                    nr.emplace_back(e.first, -f * e.second);
                    colrows[e.first].push_back(i);
                    ++ccount[e.first];
                    col_heap.emplace(ccount[e.first], e.first);
                }
            }
            size_t o = 0;
            for (auto& e : nr) {
                if (e.second == 0.0) {                                       // exact cancellation (decomposition/mod.rs:178)
                    if (--ccount[e.first] == 1) col_single.push_back(e.first);
                    col_heap.emplace(ccount[e.first], e.first);
                    continue;
                }
                nr[o++] = e;
            }
            nr.resize(o);
            ri.swap(nr);
            if (ri.size() == 1) row_single.push_back(i);
            row_heap.emplace((int32_t)ri.size(), i);
        }
        rows[pi].clear();
    }

    // L and U in pivot-step coordinates
    std::vector<std::vector<std::pair<int32_t, double>>> lrows(m), lcols(m), urows_s(m), ucols(m);
    for (auto& e : lent) {
        const int32_t k = step_of_row[e.row];
        lrows[k].emplace_back(e.pos, e.val);          // L[k, l], l < k
        lcols[e.pos].emplace_back(k, e.val);          // column l of L: rows k > l
    }
    for (int32_t k = 0; k < m; ++k)
        for (auto& e : urows[k]) {
            const int32_t l = step_of_col[e.first];
            urows_s[k].emplace_back(l, e.second);     // U[k, l], l > k
            ucols[l].emplace_back(k, e.second);       // column l of U: rows k < l
        }
    for (auto& v : lrows) std::sort(v.begin(), v.end());
    for (auto& v : lcols) std::sort(v.begin(), v.end());
    for (auto& v : urows_s) std::sort(v.begin(), v.end());
    for (auto& v : ucols) std::sort(v.begin(), v.end());
    std::vector<double> ones(m, 1.0);
    build_schedule(m, lrows, ones, true, &out->Lf);
    build_schedule(m, urows_s, udiag, false, &out->Uf);
    build_schedule(m, ucols, udiag, true, &out->Ub);
    build_schedule(m, lcols, ones, false, &out->Lb);
    out->nnz_l = (int64_t)lent.size();
    out->nnz_u = 0;
    for (auto& v : urows) out->nnz_u += (int64_t)v.size() + 1;
    return true;
}

void ell_pack(const TriangularSchedule& t, bool keep_trivial, EllPacked* out) {
    const int32_t m = (int32_t)t.diag.size();
    const int32_t nlev = (int32_t)t.level_ptr.size() - 1;
    *out = EllPacked{};
    out->lvl_pass.assign(nlev + 1, 0);
    out->rdiag.assign(m, 1.0);
    out->rovf.assign(2 * (size_t)m, 0);
    constexpr int32_t kLanes = 256;                      // threads that walk the passes (relp_lu_device.h: ell_solve)
    struct Row { int32_t lg, k; };
    for (int32_t l = 0; l < nlev; ++l) {
        out->lvl_pass[l] = (int32_t)out->passes.size();
        std::vector<Row> rows;
        for (int32_t i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) {
            const int32_t k = t.level_rows[i], n = t.ptr[k + 1] - t.ptr[k];
            if (n == 0 && !keep_trivial && t.diag[k] == 1.0) continue;
            int32_t lg = 0;
            while ((1 << lg) < n + 1 && lg < 6) ++lg;
            rows.push_back(Row{lg, k});
        }
        std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.lg > b.lg; });
        const size_t first_pass = out->passes.size();
        for (size_t i = 0; i < rows.size();) {
            EllPassHost ps{(int32_t)out->sidx.size(), 0, 0, l};
            int32_t pos = 0, max_lg = 0, ovf = 0;
            while (i < rows.size() && pos + (1 << rows[i].lg) <= kLanes) {
                const int32_t k = rows[i].k, lg = rows[i].lg, w = 1 << lg, n = t.ptr[k + 1] - t.ptr[k];
                out->rdiag[k] = 1.0 / t.diag[k];
                out->sidx.push_back((uint16_t)(k | (lg << 12)));             // the row's own unknown: -(-1) x[k]
                out->sval.push_back(-1.0);
                for (int32_t j = 1; j < w; ++j) {
                    const bool has = j - 1 < n;
                    out->sidx.push_back((uint16_t)((has ? t.idx[t.ptr[k] + j - 1] : 0) | (lg << 12)));
                    out->sval.push_back(has ? t.val[t.ptr[k] + j - 1] : 0.0);
                }
                out->rovf[2 * (size_t)k] = (int32_t)out->oidx.size();
                for (int32_t e = t.ptr[k] + w - 1; e < t.ptr[k + 1]; ++e) { out->oidx.push_back((uint16_t)t.idx[e]); out->oval.push_back(t.val[e]); ovf = 1; }
                out->rovf[2 * (size_t)k + 1] = (int32_t)out->oidx.size();
                max_lg = std::max(max_lg, lg);
                pos += w;
                ++i;
            }
            ps.lanes = pos;
            ps.info = max_lg | (ovf << 9);
            out->passes.push_back(ps);
        }
        if (out->passes.size() > first_pass) out->passes.back().info |= 1 << 8;
    }
    out->lvl_pass[nlev] = (int32_t)out->passes.size();
    if (out->oidx.empty()) out->rovf.clear();            // no row has more than 63 entries: the ranges are never read
}

bool lu_from_triangles(int32_t m, const std::vector<std::vector<std::pair<int32_t, double>>>& lcols_in,
                       const std::vector<std::vector<std::pair<int32_t, double>>>& ucols_in, LUFactors* out, std::string* err) {
    std::vector<std::vector<std::pair<int32_t, double>>> lrows(m), lcols(m), urows(m), ucols(m);
    std::vector<double> udiag(m, 0.0), ones(m, 1.0);
    out->nnz_l = 0; out->nnz_u = 0;
    for (int32_t j = 0; j < (int32_t)lcols_in.size() && j < m; ++j)
        for (auto& e : lcols_in[j]) {
            if (e.first <= j || e.first >= m) { if (err) *err = "L entry outside the strict lower triangle"; return false; }
            lrows[e.first].emplace_back(j, e.second); lcols[j].emplace_back(e.first, e.second); ++out->nnz_l;
        }
    for (int32_t j = 0; j < m; ++j)
        for (auto& e : ucols_in[j]) {
            if (e.first > j || e.first < 0) { if (err) *err = "U entry below the diagonal"; return false; }
            ++out->nnz_u;
            if (e.first == j) { udiag[j] = e.second; continue; }
            urows[e.first].emplace_back(j, e.second); ucols[j].emplace_back(e.first, e.second);
        }
    for (int32_t j = 0; j < m; ++j) if (udiag[j] == 0.0) { if (err) *err = "zero on the diagonal of U"; return false; }
    for (auto& v : lrows) std::sort(v.begin(), v.end());
    for (auto& v : urows) std::sort(v.begin(), v.end());
    out->m = m;
    out->rowperm.resize(m); out->colperm.resize(m);
    for (int32_t k = 0; k < m; ++k) { out->rowperm[k] = k; out->colperm[k] = k; }
    build_schedule(m, lrows, ones, true, &out->Lf);
    build_schedule(m, urows, udiag, false, &out->Uf);
    build_schedule(m, ucols, udiag, true, &out->Ub);
    build_schedule(m, lcols, ones, false, &out->Lb);
    return true;
}

namespace {
void solve_schedule_host(const TriangularSchedule& s, std::vector<double>& x) {
    const int32_t nl = (int32_t)s.level_ptr.size() - 1;
    for (int32_t l = 0; l < nl; ++l)
        for (int32_t t = s.level_ptr[l]; t < s.level_ptr[l + 1]; ++t) {
            const int32_t k = s.level_rows[t];
            double sum = x[k];
            for (int32_t e = s.ptr[k]; e < s.ptr[k + 1]; ++e) sum -= s.val[e] * x[s.idx[e]];
            x[k] = sum / s.diag[k];
        }
}
}  // namespace

void lu_ftran_host(const LUFactors& f, const std::vector<double>& a, std::vector<double>* x) {
    std::vector<double> w(f.m);
    for (int32_t k = 0; k < f.m; ++k) w[k] = a[f.rowperm[k]];
    solve_schedule_host(f.Lf, w);
    solve_schedule_host(f.Uf, w);
    x->assign(f.m, 0.0);
    for (int32_t k = 0; k < f.m; ++k) (*x)[f.colperm[k]] = w[k];
}

void lu_btran_host(const LUFactors& f, const std::vector<double>& c, std::vector<double>* z) {
    std::vector<double> w(f.m);
    for (int32_t k = 0; k < f.m; ++k) w[k] = c[f.colperm[k]];
    solve_schedule_host(f.Ub, w);
    solve_schedule_host(f.Lb, w);
    z->assign(f.m, 0.0);
    for (int32_t k = 0; k < f.m; ++k) (*z)[f.rowperm[k]] = w[k];
}

}  // namespace relp
