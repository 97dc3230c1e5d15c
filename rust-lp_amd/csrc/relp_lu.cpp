// relp_lu.cpp -- host-side sparse LU (P B Q = L U) with Markowitz-style pivoting and the level
// schedules for the device triangular solves.  See relp_lu.hpp.
#include "relp_lu.hpp"

#include <thread>

#include <algorithm>
#include <cmath>
#include <functional>
#include <queue>
#include <tuple>

namespace relp {

namespace {

// rows as CSR (ptr / idx / val, entries of a row sorted by index) -> schedule with levels.  `ascending`: dependencies have
// smaller indices (solve 0..m-1), otherwise larger (solve m-1..0).
void finish_schedule(int32_t m, const std::vector<double>& diag, bool ascending, TriangularSchedule* s) {
    s->diag = diag;
    std::vector<int32_t> lev(m, 0);
    int32_t nlev = m > 0 ? 1 : 0;
    auto visit = [&](int32_t k) {
        int32_t l = 0;
        for (int32_t e = s->ptr[k]; e < s->ptr[k + 1]; ++e) l = std::max(l, lev[s->idx[e]] + 1);
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

// (row, index, value) triplets -> the CSR part of a schedule; entries of a row sorted by index
struct Triplet { int32_t row, idx; double val; };
void rows_from_triplets(int32_t m, const std::vector<Triplet>& t, TriangularSchedule* s) {
    s->ptr.assign(m + 1, 0);
    for (const Triplet& e : t) ++s->ptr[e.row + 1];
    for (int32_t k = 0; k < m; ++k) s->ptr[k + 1] += s->ptr[k];
    s->idx.resize(t.size()); s->val.resize(t.size());
    std::vector<int32_t> fill(s->ptr.begin(), s->ptr.end() - 1);
    for (const Triplet& e : t) { const int32_t o = fill[e.row]++; s->idx[o] = e.idx; s->val[o] = e.val; }
    for (int32_t k = 0; k < m; ++k) {                              // insertion sort: rows are short and mostly in order
        const int32_t b = s->ptr[k], e = s->ptr[k + 1];
        for (int32_t i = b + 1; i < e; ++i) {
            const int32_t ci = s->idx[i]; const double cv = s->val[i];
            int32_t j = i - 1;
            while (j >= b && s->idx[j] > ci) { s->idx[j + 1] = s->idx[j]; s->val[j + 1] = s->val[j]; --j; }
            s->idx[j + 1] = ci; s->val[j + 1] = cv;
        }
    }
}

// The active line (row or column) with the smallest (count, index): one bit set per count, lines with 63 or more entries
// share the last set.  Replaces a lazily-deleted binary heap (same answer: the minimum over the lines' current counts).
struct CountBuckets {
    static constexpr int32_t kTop = 63;
    int32_t words = 0;
    std::vector<uint64_t> bits;                  // (kTop + 1) x words
    std::vector<int32_t> members;                // lines per bucket
    void init(int32_t n) { words = (n + 63) / 64; bits.assign((size_t)(kTop + 1) * words, 0); members.assign(kTop + 1, 0); }
    static int32_t bucket(int32_t count) { return count < kTop ? count : kTop; }
    void insert(int32_t line, int32_t count) { const int32_t b = bucket(count); bits[(size_t)b * words + (line >> 6)] |= 1ull << (line & 63); ++members[b]; }
    void erase(int32_t line, int32_t count) { const int32_t b = bucket(count); bits[(size_t)b * words + (line >> 6)] &= ~(1ull << (line & 63)); --members[b]; }
    void move(int32_t line, int32_t from, int32_t to) { if (bucket(from) != bucket(to)) { erase(line, from); insert(line, to); } }
    // smallest (count, line) with count >= min_count; -1 when there is none.  `count_of` is only asked in the shared last set.
    template <class F> int32_t top(int32_t min_count, F count_of) const {
        for (int32_t b = bucket(min_count); b <= kTop; ++b) {
            if (!members[b]) continue;
            const uint64_t* w = &bits[(size_t)b * words];
            if (b < kTop) { for (int32_t i = 0; i < words; ++i) if (w[i]) return i * 64 + __builtin_ctzll(w[i]); continue; }
            int32_t best = -1, bc = 0;
            for (int32_t i = 0; i < words; ++i)
                for (uint64_t v = w[i]; v; v &= v - 1) {
                    const int32_t line = i * 64 + __builtin_ctzll(v), c = count_of(line);
                    if (best < 0 || c < bc) { best = line; bc = c; }
                }
            return best;
        }
        return -1;
    }
};

// growable lists in one arena: list l = store[beg[l] .. beg[l] + len[l]), relocated to the end with doubled room when full
template <class T>
struct ListArena {
    std::vector<T> store;
    std::vector<int32_t> beg, len, cap;
    void init(int32_t lists, size_t reserve) { store.clear(); store.reserve(reserve); beg.assign(lists, 0); len.assign(lists, 0); cap.assign(lists, 0); }
    int32_t alloc(int32_t room) { const int32_t at = (int32_t)store.size(); store.resize(store.size() + room); return at; }
    void push(int32_t l, const T& v) {
        if (len[l] == cap[l]) {
            const int32_t room = cap[l] ? 2 * cap[l] : 4, at = alloc(room);
            for (int32_t i = 0; i < len[l]; ++i) store[at + i] = store[beg[l] + i];
            beg[l] = at; cap[l] = room;
        }
        store[beg[l] + len[l]++] = v;
    }
};

}  // namespace

void lu_levels_from_rows(int32_t m, const std::vector<double>& diag, bool ascending, TriangularSchedule* s) {
    finish_schedule(m, diag, ascending, s);
}

bool lu_factor(int32_t m, const std::vector<std::vector<std::pair<int32_t, double>>>& columns, LUFactors* out,
               std::string* err) {
    std::vector<int64_t> ptr((size_t)m + 1, 0);
    for (int32_t j = 0; j < m; ++j) ptr[(size_t)j + 1] = ptr[j] + (int64_t)columns[j].size();
    std::vector<int32_t> idx((size_t)ptr[m]);
    std::vector<double> val((size_t)ptr[m]);
    for (int32_t j = 0; j < m; ++j) {
        int64_t o = ptr[j];
        for (auto& e : columns[j]) { idx[(size_t)o] = e.first; val[(size_t)o] = e.second; ++o; }
    }
    return lu_factor_csc(m, ptr.data(), idx.data(), val.data(), out, err);
}

// The same on a flat copy of the basis (column j = entries [ptr[j], ptr[j + 1]) of idx / val): what the engine hands over at
// every refactorisation -- 64,000 separately allocated column vectors were a cache miss each, twice per factorisation.
bool lu_factor_csc(int32_t m, const int64_t* cptr, const int32_t* cidx, const double* cval, LUFactors* out, std::string* err) {
    // active submatrix, row major: entries (column, value) of row i at rc / rv [rows.beg[i], + rows.len[i]); colrows: the rows
    // that (may) hold an entry of a column (stale members are dropped whenever the list is scanned for its maximum)
    const size_t nnz = (size_t)cptr[m];
    ListArena<int32_t> rows, colrows;
    std::vector<double> rv;                                            // values parallel to rows.store
    rows.init(m, 4 * nnz + 8 * (size_t)m); colrows.init(m, 4 * nnz + 8 * (size_t)m);
    rv.reserve(4 * nnz + 8 * (size_t)m);
    std::vector<int32_t> ccount(m, 0);
    {
        std::vector<int32_t> rcount(m, 0);
        for (int32_t j = 0; j < m; ++j)
            for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) if (cval[e] != 0.0) { ++rcount[cidx[e]]; ++ccount[j]; }
        // (every list gets its room by one prefix sum and one resize: 128,000 separate allocations at 64,000 rows were a third
        // of a refactorisation that has no bump to eliminate)
        int64_t at = 0;
        for (int32_t i = 0; i < m; ++i) { rows.beg[i] = (int32_t)at; rows.cap[i] = rcount[i] + 4; at += rcount[i] + 4; }
        rows.store.resize((size_t)at);
        rv.resize(rows.store.size());
        at = 0;
        for (int32_t j = 0; j < m; ++j) { colrows.beg[j] = (int32_t)at; colrows.cap[j] = ccount[j] + 4; at += ccount[j] + 4; }
        colrows.store.resize((size_t)at);
        for (int32_t j = 0; j < m; ++j)
            for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) {
                if (cval[e] == 0.0) continue;
                const int32_t i = cidx[e], o = rows.beg[i] + rows.len[i]++;
                rows.store[o] = j; rv[o] = cval[e];
                colrows.store[colrows.beg[j] + colrows.len[j]++] = i;
            }
    }
    // a rewritten row that outgrows its room moves to the end of the arena
    auto row_store = [&](int32_t i, const std::vector<int32_t>& c, const std::vector<double>& v, int32_t n) {
        if (n > rows.cap[i]) { rows.beg[i] = rows.alloc(n + 4); rows.cap[i] = n + 4; rv.resize(rows.store.size()); }
        const int32_t b = rows.beg[i];
        for (int32_t t = 0; t < n; ++t) { rows.store[b + t] = c[t]; rv[b + t] = v[t]; }
        rows.len[i] = n;
    };
    auto find_in_row = [&](int32_t i, int32_t j) {                     // position of column j in row i, or -1
        const int32_t b = rows.beg[i], e = b + rows.len[i];
        for (int32_t t = b; t < e; ++t) if (rows.store[t] == j) return t;
        return -1;
    };
    std::vector<char> row_done(m, 0);
    std::vector<int32_t> step_of_row(m, -1), step_of_col(m, -1);
    out->m = m;
    out->rowperm.assign(m, 0);
    out->colperm.assign(m, 0);
    std::vector<Triplet> lrow_t, lcol_t, urow_t, ucol_t;               // L / U entries in original (row, column) coordinates
    lrow_t.reserve(2 * nnz); urow_t.reserve(2 * nnz);
    std::vector<int32_t> ubeg(m + 1, 0), ucol; std::vector<double> uval;   // U row of step k: (original column, value)
    ucol.reserve(2 * nnz); uval.reserve(2 * nnz);
    std::vector<double> udiag(m, 1.0);
    std::vector<int32_t> mark(m, -1), pos(m, 0);
    int32_t stamp = 0;
    // Singletons first: most of an LP basis is triangular (slacks, bound rows).  A column with one active
    // entry is a fill-free pivot that always passes the threshold; a row with one active entry is fill-free
    // and is taken when it passes it.  Both are served from stacks, so the searches below only run on
    // the "bump" that is left.
    std::vector<int32_t> col_single, row_single;
    col_single.reserve(4 * (size_t)m); row_single.reserve(4 * (size_t)m);
    for (int32_t j = 0; j < m; ++j) if (ccount[j] == 1) col_single.push_back(j);
    for (int32_t i = 0; i < m; ++i) if (rows.len[i] == 1) row_single.push_back(i);
    static const bool peel_stacks = std::getenv("RELP_LU_PEEL_STACKS") && std::atoi(std::getenv("RELP_LU_PEEL_STACKS")) != 0;
    int64_t col_head = 0, col_end = (int64_t)col_single.size(), row_head = 0, row_end = 0;
    bool peel_cols = true;
    // the sparsest active row / column of the bump, ties to the lower index
    // (built when the first pivot has to be SEARCHED for: while singletons last -- a whole multi-commodity basis of 64,000 rows,
    // the triangular part of every other -- the two sets of 64 bit-vectors of m bits each are neither filled nor maintained;
    // the sets then hold exactly what incremental maintenance would have left: the active lines by their current counts)
    CountBuckets row_set, col_set;
    bool buckets_live = false;
    auto buckets_build = [&]() {
        row_set.init(m); col_set.init(m);
        for (int32_t i = 0; i < m; ++i) if (!row_done[i]) row_set.insert(i, rows.len[i]);
        for (int32_t j = 0; j < m; ++j) if (step_of_col[j] < 0) col_set.insert(j, ccount[j]);
        buckets_live = true;
    };
    auto col_count_changed = [&](int32_t j, int32_t from) {
        if (buckets_live && step_of_col[j] < 0) col_set.move(j, from, ccount[j]);
    };
    // largest active |entry| of column j; the scan also drops the stale members of colrows[j] (finished rows,
    // rows that lost the entry, duplicates), so the lists stay as short as the columns are
    std::vector<int32_t> seen_stamp(m, -1);
    int32_t seen_tick = 0;
    auto col_max_of = [&](int32_t j) {
        double mx = 0.0;
        ++seen_tick;
        int32_t o = 0;
        const int32_t b = colrows.beg[j], n = colrows.len[j];
        for (int32_t t = 0; t < n; ++t) {
            const int32_t i = colrows.store[b + t];
            if (row_done[i] || seen_stamp[i] == seen_tick) continue;
            const int32_t at = find_in_row(i, j);
            if (at >= 0) { mx = std::max(mx, std::fabs(rv[at])); seen_stamp[i] = seen_tick; colrows.store[b + o++] = i; }
        }
        colrows.len[j] = o;
        return mx;
    };
    std::vector<std::pair<int32_t, int32_t>> order;                    // (column count, column)
    std::vector<int32_t> nc; std::vector<double> nv;                   // the row being rewritten
    nc.resize(m); nv.resize(m);
    constexpr double kThreshold = 0.1;

    for (int32_t k = 0; k < m; ++k) {
        int32_t spi = -1, spj = -1; double spv = 0.0;
        // Singletons are taken in ROUNDS, column singletons and row singletons in turn (queues, a round = what was queued when it
        // began): a basis that is all singletons -- the multi-commodity bases of 64,000 rows -- is triangular either way, but taken
        // as column singletons only (what the stacks of rounds 1-3 did) it is ONE triangle, U, 115 levels deep; eaten from both
        // ends it is an L and a U of 32 levels each, and the sweeps of a pivot walk 64 levels instead of 115 (measured: 183,000
        // clocks per pivot against 223,000).  RELP_LU_PEEL_STACKS=1 restores the old order (measurements).
        auto next_col_singleton = [&]() {
            while (col_head < col_end && spj < 0) {
                const int32_t j = col_single[col_head++];
                if (step_of_col[j] >= 0 || ccount[j] != 1) continue;
                for (int32_t t = 0; t < colrows.len[j]; ++t) {
                    const int32_t i = colrows.store[colrows.beg[j] + t];
                    if (row_done[i]) continue;
                    const int32_t at = find_in_row(i, j);
                    if (at >= 0 && rv[at] != 0.0) { spi = i; spj = j; spv = rv[at]; break; }
                }
            }
        };
        auto next_row_singleton = [&]() {
            while (row_head < row_end && spj < 0) {
                const int32_t i = row_single[row_head++];
                if (row_done[i] || rows.len[i] != 1) continue;
                const int32_t j = rows.store[rows.beg[i]]; const double v = rv[rows.beg[i]];
                if (v != 0.0 && std::fabs(v) >= 0.1 * col_max_of(j)) { spi = i; spj = j; spv = v; }
            }
        };
        if (peel_stacks) {                                  // columns while there are any, then rows, newest first
            while (!col_single.empty() && spj < 0) { col_head = (int64_t)col_single.size() - 1; col_end = col_head + 1; next_col_singleton(); col_single.pop_back(); }
            while (!row_single.empty() && spj < 0) { row_head = (int64_t)row_single.size() - 1; row_end = row_head + 1; next_row_singleton(); row_single.pop_back(); }
        } else {
            for (int turn = 0; turn < 3 && spj < 0; ++turn) {
                if (peel_cols) { next_col_singleton(); if (spj < 0) { peel_cols = false; row_end = (int64_t)row_single.size(); } }
                else { next_row_singleton(); if (spj < 0) { peel_cols = true; col_end = (int64_t)col_single.size(); } }
            }
        }
        // Markowitz search restricted to the sparsest active row and the sparsest active column
        // (pivoting.rs:45-81 searches every remaining entry): candidate A = the entry of the sparsest row
        // with the lowest column count, candidate B = the entry of the sparsest column with the lowest row
        // count; the lower (r - 1)(c - 1) wins.
        int32_t ra = -1;
        if (spj < 0) {
            if (!buckets_live) buckets_build();
            ra = row_set.top(0, [&](int32_t i) { return rows.len[i]; });
            if (ra < 0) { if (err) *err = "singular basis (no active row)"; return false; }
            if (rows.len[ra] == 0) { if (err) *err = "singular basis (empty row during LU)"; return false; }
        }
        // threshold partial pivoting: a pivot must be at least kThreshold of the largest active entry of
        // its column, which bounds every multiplier of L by 1 / kThreshold
        int32_t pi = spi, pj = spj; double pv = spv; int64_t best = spj >= 0 ? 0 : -1;
        if (spj < 0) {
            // candidate A: entries of the sparsest row, by ascending column count
            order.clear();
            const int32_t b = rows.beg[ra], n = rows.len[ra];
            for (int32_t t = 0; t < n; ++t) if (rv[b + t] != 0.0) order.emplace_back(ccount[rows.store[b + t]], rows.store[b + t]);
            std::sort(order.begin(), order.end());
            for (auto& oc : order) {
                const double v = rv[find_in_row(ra, oc.second)];
                if (std::fabs(v) < kThreshold * col_max_of(oc.second)) continue;
                pi = ra; pj = oc.second; pv = v;
                best = (int64_t)(n - 1) * (ccount[pj] - 1);
                break;
            }
        }
        if (best != 0) {
            const int32_t cb = col_set.top(1, [&](int32_t j) { return ccount[j]; });
            if (cb >= 0) {
                const double cmax = col_max_of(cb);
                for (int32_t t = 0; t < colrows.len[cb]; ++t) {
                    const int32_t i = colrows.store[colrows.beg[cb] + t];
                    if (row_done[i]) continue;
                    const int32_t at = find_in_row(i, cb);
                    if (at < 0) continue;
                    const double v = rv[at];
                    if (v == 0.0 || std::fabs(v) < kThreshold * cmax) continue;
                    const int64_t cost = (int64_t)(rows.len[i] - 1) * (ccount[cb] - 1);
                    if (best < 0 || cost < best) { best = cost; pi = i; pj = cb; pv = v; }
                }
            }
        }
        if (pj < 0 && ra >= 0) {
            // no entry of the sparsest row / column passes the threshold: take the largest entry of the
            // sparsest row's best column (always acceptable)
            double bestv = 0.0;
            for (int32_t t = 0; t < rows.len[ra] && pj < 0; ++t) {
                const int32_t j = rows.store[rows.beg[ra] + t];
                for (int32_t u = 0; u < colrows.len[j]; ++u) {
                    const int32_t i = colrows.store[colrows.beg[j] + u];
                    if (row_done[i]) continue;
                    const int32_t at = find_in_row(i, j);
                    if (at >= 0 && std::fabs(rv[at]) > bestv) { bestv = std::fabs(rv[at]); pi = i; pj = j; pv = rv[at]; }
                }
            }
        }
        if (pj < 0) { if (err) *err = "singular basis (no acceptable pivot)"; return false; }
        out->rowperm[k] = pi; out->colperm[k] = pj;
        step_of_row[pi] = k; step_of_col[pj] = k;
        row_done[pi] = 1;
        if (buckets_live) { row_set.erase(pi, rows.len[pi]); col_set.erase(pj, ccount[pj]); }
        udiag[k] = pv;
        ubeg[k] = (int32_t)ucol.size();
        {
            const int32_t b = rows.beg[pi], n = rows.len[pi];
            for (int32_t t = 0; t < n; ++t) {
                const int32_t c = rows.store[b + t];
                const int32_t before = ccount[c]--;
                if (ccount[c] == 1) col_single.push_back(c);
                col_count_changed(c, before);
                if (c != pj) { ucol.push_back(c); uval.push_back(rv[b + t]); }
            }
        }
        const int32_t pb = ubeg[k], pn = (int32_t)ucol.size() - pb;
        // eliminate column pj from the other active rows
        for (int32_t t = 0; t < colrows.len[pj]; ++t) {
            const int32_t i = colrows.store[colrows.beg[pj] + t];
            if (row_done[i]) continue;
            const int32_t at = find_in_row(i, pj);
            if (at < 0) continue;                                  // stale list entry
            const double f = rv[at] / pv;
            lrow_t.push_back(Triplet{i, k, f});
            ++stamp;
            const int32_t b = rows.beg[i], n = rows.len[i];
            int32_t nn = 0;
            if ((int32_t)nc.size() < n + pn) { nc.resize(n + pn); nv.resize(n + pn); }
            for (int32_t u = 0; u < n; ++u) {
                const int32_t c = rows.store[b + u];
                if (c == pj) continue;
                mark[c] = stamp; pos[c] = nn;
                nc[nn] = c; nv[nn] = rv[b + u]; ++nn;
            }
            for (int32_t u = 0; u < pn; ++u) {
                const int32_t c = ucol[pb + u];
                if (mark[c] == stamp) nv[pos[c]] -= f * uval[pb + u];
                else {
                    mark[c] = stamp; pos[c] = nn;
                    nc[nn] = c; nv[nn] = -f * uval[pb + u]; ++nn;
                    colrows.push(c, i);
                    const int32_t before = ccount[c]++;
                    col_count_changed(c, before);
                }
            }
            int32_t o = 0;
            for (int32_t u = 0; u < nn; ++u) {
                if (nv[u] == 0.0) {                                          // exact cancellation (decomposition/mod.rs:178)
                    const int32_t c = nc[u];
                    const int32_t before = ccount[c]--;
                    if (ccount[c] == 1) col_single.push_back(c);
                    col_count_changed(c, before);
                    continue;
                }
                nc[o] = nc[u]; nv[o] = nv[u]; ++o;
            }
            if (buckets_live) row_set.move(i, n, o);
            row_store(i, nc, nv, o);
            if (o == 1) row_single.push_back(i);
        }
        rows.len[pi] = 0;
    }
    ubeg[m] = (int32_t)ucol.size();

    // L and U in pivot-step coordinates: rows of L (FTRAN), rows of U, columns of U (BTRAN), columns of L
    for (Triplet& e : lrow_t) e.row = step_of_row[e.row];             // L[k, l], l < k
    lcol_t.reserve(lrow_t.size());
    for (const Triplet& e : lrow_t) lcol_t.push_back(Triplet{e.idx, e.row, e.val});      // column l of L: rows k > l
    urow_t.clear(); ucol_t.reserve(ucol.size());
    for (int32_t k = 0; k < m; ++k)
        for (int32_t t = ubeg[k]; t < ubeg[k + 1]; ++t) {
            const int32_t l = step_of_col[ucol[t]];
            urow_t.push_back(Triplet{k, l, uval[t]});                  // U[k, l], l > k
            ucol_t.push_back(Triplet{l, k, uval[t]});                  // column l of U: rows k < l
        }
    std::vector<double> ones(m, 1.0);
    // (the four schedules are independent; beyond a few thousand rows the two of U, the larger ones, go to a second thread)
    auto forward = [&]() {
        rows_from_triplets(m, urow_t, &out->Uf); finish_schedule(m, udiag, false, &out->Uf);
        rows_from_triplets(m, lrow_t, &out->Lf); finish_schedule(m, ones, true, &out->Lf);
    };
    auto backward = [&]() {
        rows_from_triplets(m, ucol_t, &out->Ub); finish_schedule(m, udiag, true, &out->Ub);
        rows_from_triplets(m, lcol_t, &out->Lb); finish_schedule(m, ones, false, &out->Lb);
    };
    if (m >= 8192) {
        struct Joined {                                    // (joins on every path out: an exception in forward() must not
            std::thread t;                                 // reach std::terminate through a joinable thread)
            ~Joined() { if (t.joinable()) t.join(); }
        } helper{std::thread(backward)};
        forward();
    } else {
        forward(); backward();
    }
    out->nnz_l = (int64_t)lrow_t.size();
    out->nnz_u = (int64_t)ucol.size() + m;
    return true;
}

void fuse_levels(const TriangularSchedule& t, bool maskable, bool keep_trivial, int32_t lane_cap, FusedSchedule* out) {
    const int32_t m = (int32_t)t.diag.size();
    const int32_t nlev = (int32_t)t.level_ptr.size() - 1;
    *out = FusedSchedule{};
    out->rhs_base = m + 1;
    const int32_t rhs_base = m + 1;
    // expanded rows, flat: row k = entries [ebeg[k], ebeg[k] + elen[k]) of (esrc, ecoef); maskable: the pivots an entry's
    // substitution path runs through = pool[ev0[e], ev0[e] + evn[e])
    std::vector<int32_t> esrc, ev0, evn, ebeg(m, 0), elen(m, 0), pool, grp(m, -1), group_first, level_group(nlev, 0);
    std::vector<double> ecoef;
    const size_t room = 3 * t.idx.size() + (size_t)m + 64;
    esrc.reserve(room); ecoef.reserve(room);
    if (maskable) { ev0.reserve(room); evn.reserve(room); pool.reserve(room); }
    std::vector<int32_t> seen(2 * (size_t)m + 2, -1), where(2 * (size_t)m + 2, 0);     // combining the duplicates of a row
    int32_t seen_tick = 0;
    auto lanes_of = [](int32_t n) { int32_t lg = 0; while ((1 << lg) < n + 1 && lg < 6) ++lg; return 1 << lg; };
    auto packed = [&](int32_t k) { return keep_trivial || t.ptr[k + 1] != t.ptr[k] || t.diag[k] != 1.0; };
    auto push = [&](int32_t src, double coef, int32_t v0, int32_t vn) {
        esrc.push_back(src); ecoef.push_back(coef);
        if (maskable) { ev0.push_back(v0); evn.push_back(vn); }
    };
    auto truncate = [&](size_t entries, size_t pooled) {
        esrc.resize(entries); ecoef.resize(entries);
        if (maskable) { ev0.resize(entries); evn.resize(entries); pool.resize(pooled); }
    };
    int32_t g = -1, lanes = 0;
    for (int32_t l = 0; l < nlev; ++l) {
        const int32_t r0 = t.level_ptr[l], r1 = t.level_ptr[l + 1];
        const size_t mark_e = esrc.size(), mark_p = pool.size();
        bool ok = g >= 0 && lane_cap > 0;
        int32_t add = 0;
        if (ok) {                                                    // the rows as they are already overflow the pass: do not try
            int32_t least = 0;
            for (int32_t i = r0; i < r1 && lanes + least <= lane_cap; ++i) {
                const int32_t k = t.level_rows[i];
                if (packed(k)) least += lanes_of(t.ptr[k + 1] - t.ptr[k]);
            }
            ok = lanes + least <= lane_cap;
        }
        for (int32_t i = r0; ok && i < r1; ++i) {                    // try the level as part of the open group
            const int32_t k = t.level_rows[i];
            const int32_t b = (int32_t)esrc.size();
            ++seen_tick;
            auto term = [&](int32_t src, double coef, int32_t v0, int32_t vn) {
                if (!maskable) {                                     // one entry per index, summed in the order met
                    if (seen[src] == seen_tick) { ecoef[where[src]] += coef; return; }
                    seen[src] = seen_tick; where[src] = (int32_t)esrc.size();
                }
                push(src, coef, v0, vn);
            };
            for (int32_t e = t.ptr[k]; e < t.ptr[k + 1] && (int32_t)esrc.size() - b <= 64; ++e) {
                const int32_t j = t.idx[e];
                const double v = t.val[e];
                if (grp[j] != g) { term(j, v, 0, 0); continue; }
                const double f = v / t.diag[j];
                int32_t v0 = (int32_t)pool.size();
                if (maskable) pool.push_back(j);
                term(rhs_base + j, f, v0, 1);
                for (int32_t u = ebeg[j]; u < ebeg[j] + elen[j]; ++u) {
                    v0 = (int32_t)pool.size();
                    if (maskable) { for (int32_t a = 0; a < evn[u]; ++a) { const int32_t w = pool[ev0[u] + a]; pool.push_back(w); } pool.push_back(j); }
                    term(esrc[u], -f * ecoef[u], v0, maskable ? evn[u] + 1 : 0);
                }
            }
            const int32_t n = (int32_t)esrc.size() - b;
            if (n > 63) { ok = false; break; }
            ebeg[k] = b; elen[k] = n;                                 // (overwritten below when the level is rejected)
            if (packed(k)) add += lanes_of(n);
            if (lanes + add > lane_cap) { ok = false; break; }
        }
        if (ok) {
            for (int32_t i = r0; i < r1; ++i) { const int32_t k = t.level_rows[i]; if (packed(k)) grp[k] = g; }
            lanes += add;
        } else {                                                    // the level opens a new group with its rows as they are
            truncate(mark_e, mark_p);
            ++g; lanes = 0;
            group_first.push_back(l);
            for (int32_t i = r0; i < r1; ++i) {
                const int32_t k = t.level_rows[i];
                ebeg[k] = (int32_t)esrc.size(); elen[k] = t.ptr[k + 1] - t.ptr[k];
                for (int32_t e = t.ptr[k]; e < t.ptr[k + 1]; ++e) push(t.idx[e], t.val[e], 0, 0);
                if (packed(k)) { grp[k] = g; lanes += lanes_of(elen[k]); }
            }
        }
        level_group[l] = g;
    }
    const int32_t ngroups = g + 1;
    TriangularSchedule& s = out->s;
    s.diag = t.diag;
    s.level_rows = t.level_rows;                                    // rows stay in level order: groups are runs of levels
    s.level_ptr.assign(ngroups + 1, m);
    for (int32_t q = 0; q < ngroups; ++q) s.level_ptr[q] = t.level_ptr[group_first[q]];
    if (ngroups == 0) s.level_ptr.assign(1, 0);
    s.ptr.assign(m + 1, 0);
    for (int32_t k = 0; k < m; ++k) s.ptr[k + 1] = s.ptr[k] + elen[k];
    s.idx.resize(s.ptr[m]); s.val.resize(s.ptr[m]);
    for (int32_t k = 0; k < m; ++k)
        for (int32_t u = 0; u < elen[k]; ++u) { s.idx[s.ptr[k] + u] = esrc[ebeg[k] + u]; s.val[s.ptr[k] + u] = ecoef[ebeg[k] + u]; }
    if (maskable) {
        out->via_ptr.assign(m + 1, 0);
        for (int32_t k = 0; k < m; ++k)
            for (int32_t u = ebeg[k]; u < ebeg[k] + elen[k]; ++u)
                for (int32_t a = 0; a < evn[u]; ++a) ++out->via_ptr[pool[ev0[u] + a] + 1];
        for (int32_t k = 0; k < m; ++k) out->via_ptr[k + 1] += out->via_ptr[k];
        out->via_ent.resize(out->via_ptr[m]);
        std::vector<int32_t> fill(out->via_ptr.begin(), out->via_ptr.end() - 1);
        for (int32_t k = 0; k < m; ++k)
            for (int32_t u = 0; u < elen[k]; ++u)
                for (int32_t a = 0; a < evn[ebeg[k] + u]; ++a) out->via_ent[fill[pool[ev0[ebeg[k] + u] + a]]++] = s.ptr[k] + u;
    }
    // where a sweep may start when everything up to and including pivot p's level is zero
    out->start_after.assign(m, -1);
    for (int32_t l = 0; l < nlev; ++l) {
        const int32_t q = level_group[l];
        const bool last_of_group = l + 1 == nlev || level_group[l + 1] != q;
        for (int32_t i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) out->start_after[t.level_rows[i]] = last_of_group ? q : q - 1;
    }
}

void ell_pack(const FusedSchedule& f, bool keep_trivial, EllPacked* out, bool wide, bool compact_rhs, int32_t triv_min) {
    const TriangularSchedule& t = f.s;
    const int32_t m = (int32_t)t.diag.size();
    const int32_t nlev = (int32_t)t.level_ptr.size() - 1;
    *out = EllPacked{};
    out->lvl_pass.assign(nlev + 1, 0);
    out->rdiag.assign((size_t)m + 1, 1.0);
    out->rovf.assign(2 * (size_t)m, 0);
    std::vector<int32_t> lane_of(t.idx.size(), -1);      // entry -> its slot (entries in overflow lists: -1)
    constexpr int32_t kLanes = 256;                      // threads that walk the passes (relp_lu_device.h: ell_solve)
    struct Row { int32_t lg, k; };
    // compact the right-hand-side copies: pivots that are read through their copy, ascending -> rhs_base + rank
    std::vector<int32_t> rhs_id;
    int32_t n_trivial = 0;
    if (keep_trivial) for (int32_t k = 0; k < m; ++k) if (t.ptr[k + 1] == t.ptr[k]) ++n_trivial;
    const bool list_trivial = keep_trivial && n_trivial >= triv_min;
    if (f.rhs_base > 0 && compact_rhs) {
        std::vector<char> used((size_t)m, 0);
        bool any = false;
        for (int32_t v : t.idx) if (v >= f.rhs_base) { used[v - f.rhs_base] = 1; any = true; }
        if (any) {
            rhs_id.assign((size_t)m, -1);
            for (int32_t j = 0; j < m; ++j) if (used[j]) { rhs_id[j] = (int32_t)out->rhs_src.size(); out->rhs_src.push_back(j); }
        }
    }
    auto slot_index = [&](int32_t v) { return (!rhs_id.empty() && v >= f.rhs_base) ? f.rhs_base + rhs_id[v - f.rhs_base] : v; };
    auto n_slots = [&]() { return wide ? out->sidx32.size() : out->sidx.size(); };
    auto push_slot = [&](int32_t index, int32_t lg) {
        if (wide) out->sidx32.push_back((uint32_t)index | ((uint32_t)lg << kEllLgShiftWide));
        else out->sidx.push_back((uint16_t)(index | (lg << kEllLgShift)));
    };
    for (int32_t l = 0; l < nlev; ++l) {
        out->lvl_pass[l] = (int32_t)out->passes.size();
        std::vector<Row> rows;
        for (int32_t i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) {
            const int32_t k = t.level_rows[i], n = t.ptr[k + 1] - t.ptr[k];
            if (n == 0 && !keep_trivial && t.diag[k] == 1.0) continue;
            if (n == 0 && list_trivial) {                // (such a row has no dependency: it sits in the first level)
                out->rdiag[k] = 1.0 / t.diag[k];
                // (a unit diagonal leaves x[k] as it is; a pivot that is masked later is zeroed by the kernel before every
                // sweep -- ft_ftran, ft_ut_solve -- so its row need not be visited for that either)
                if (t.diag[k] != 1.0) out->triv.push_back(k);
                continue;
            }
            int32_t lg = 0;
            while ((1 << lg) < n + 1 && lg < 6) ++lg;
            rows.push_back(Row{lg, k});
        }
        std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.lg > b.lg; });
        const size_t first_pass = out->passes.size();
        for (size_t i = 0; i < rows.size();) {
            EllPassHost ps{(int32_t)n_slots(), 0, 0, l};
            int32_t pos = 0, max_lg = 0, ovf = 0;
            while (i < rows.size() && pos + (1 << rows[i].lg) <= kLanes) {
                const int32_t k = rows[i].k, lg = rows[i].lg, w = 1 << lg, n = t.ptr[k + 1] - t.ptr[k];
                out->rdiag[k] = 1.0 / t.diag[k];
                push_slot(k, lg);                               // the row's own unknown: -(-1) x[k]
                out->sval.push_back(-1.0);
                for (int32_t j = 1; j < w; ++j) {
                    const bool has = j - 1 < n;
                    if (has) lane_of[t.ptr[k] + j - 1] = (int32_t)n_slots();
                    push_slot(has ? slot_index(t.idx[t.ptr[k] + j - 1]) : 0, lg);
                    out->sval.push_back(has ? t.val[t.ptr[k] + j - 1] : 0.0);
                }
                out->rovf[2 * (size_t)k] = (int32_t)out->oval.size();
                for (int32_t e = t.ptr[k] + w - 1; e < t.ptr[k + 1]; ++e) {
                    if (wide) out->oidx32.push_back((uint32_t)slot_index(t.idx[e])); else out->oidx.push_back((uint16_t)slot_index(t.idx[e]));
                    out->oval.push_back(t.val[e]); ovf = 1;
                }
                out->rovf[2 * (size_t)k + 1] = (int32_t)out->oval.size();
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
    out->reach.assign((size_t)m, 0x7fffffff);
    for (int32_t l = 0; l < nlev; ++l)
        for (int32_t i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) {
            const int32_t k = t.level_rows[i], n = t.ptr[k + 1] - t.ptr[k];
            if (n == 0 && (list_trivial || (!keep_trivial && t.diag[k] == 1.0))) continue;   // (unit rows that are left out, or rows of the `triv` loop)
            out->reach[k] = std::min(out->reach[k], l);
            for (int32_t e = t.ptr[k]; e < t.ptr[k + 1]; ++e) {
                const int32_t p = t.idx[e] >= f.rhs_base && f.rhs_base > 0 ? t.idx[e] - f.rhs_base : t.idx[e];
                out->reach[p] = std::min(out->reach[p], l);
            }
        }
    if (out->oval.empty()) out->rovf.clear();            // no row has more than 63 entries: the ranges are never read
    if (!f.via_ptr.empty()) {                            // (substituted entries never overflow: fuse_levels caps rows at 63)
        out->via_ptr = f.via_ptr;
        out->via_pos.resize(f.via_ent.size());
        for (size_t a = 0; a < f.via_ent.size(); ++a) out->via_pos[a] = lane_of[f.via_ent[a]];
    }
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
    auto build = [&](const std::vector<std::vector<std::pair<int32_t, double>>>& r, const std::vector<double>& d, bool asc,
                     TriangularSchedule* sch) {
        std::vector<Triplet> t;
        for (int32_t k = 0; k < m; ++k) for (auto& e : r[k]) t.push_back(Triplet{k, e.first, e.second});
        rows_from_triplets(m, t, sch);
        finish_schedule(m, d, asc, sch);
    };
    build(lrows, ones, true, &out->Lf);
    build(urows, udiag, false, &out->Uf);
    build(ucols, udiag, true, &out->Ub);
    build(lcols, ones, false, &out->Lb);
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
