// relp_lu_factor_core.h -- P B Q = L U of the basis ON THE DEVICE: one workgroup, everything it
// touches in global memory or LDS, no host round trip.  Reference: the right-looking elimination with Markowitz pivoting of
// carry/lower_upper/decomposition/mod.rs:27-138 and decomposition/pivoting.rs:45-81 -- like the host factorisation
// (relp_lu.cpp: lu_factor) with singletons first and a threshold of 0.1 on the bump that is left.
//
// How it maps to a GPU.  A basis of an LP is mostly triangular: columns / rows with ONE active entry are fill-free pivots,
// and all singletons of a round are independent of each other, so a round finds and numbers them in parallel
// (`luf_select`: an ordered compaction, so the numbering is deterministic).  What survives the peeling -- the "bump" -- is
// eliminated on a dense nb x nb working copy D: per step one reduction for the sparsest row, one for the pivot among its
// entries (lowest column count that passes the threshold against the column maximum), the two patterns as ordered lists and
// the rank-1 update over |I| x |J| pairs by the whole workgroup.  Exact zeros are never stored as entries (the reference
// drops them, decomposition/mod.rs:178).  Nothing is recorded during the peeling: no entry of a peeled row or column is
// ever modified, so afterwards L and U outside the bump are read off the basis columns by the step numbers alone --
// entry (i, c) is U[k_i, k_c] when k_i < k_c, the multiplier L[k_i, k_c] = v / d_{k_c} when k_i > k_c, the diagonal when
// equal -- and the bump block off D.
//
// The file compiles twice: for the device (RELP_LUF_DEVICE: PAR_FOR = a thread-strided loop ending in a barrier) and for the
// host (plain loops), where tests/cpp/test_lu_device_model.cpp runs the very same code against lu_factor and dense solves.
// The parallel loops therefore contain nothing whose result depends on the order in which their iterations run.
#pragma once
#include <stdint.h>

#include "relp_kernels.h"

namespace relp {

enum LufStatus : int32_t { LUF_OK = 0, LUF_SINGULAR = 1, LUF_BUMP_TOO_LARGE = 2, LUF_NO_ROOM = 3 };

// What a basis column / row looks like (static per engine; the row-major copy is built by the host once)
struct LufMatrix {
    int32_t m, na, n_provider;      // rows; artificial columns (phase 1); provider columns (structural + virtual)
    DeviceCSC csc; ColumnTable ct;  // column-major: structural columns + bound rows, virtual columns, artificial columns
    const int32_t* rptr;            // m + 1: row-major copy of the provider columns: row i -> (provider column, value)
    const int32_t* rcol; const double* rval;
    const int32_t* art_of_row;      // m: artificial column that is e_row, or -1
    int32_t wrapped_na;             // artificial variables that survived phase 1 (Engine::switch_to_phase_two)
    const int32_t* wrapped_row;     // their rows by original artificial index
};

// L and U in pivot coordinates, the four views the solves use (relp_lu.hpp: TriangularSchedule without the levels)
struct LufTriangle { int32_t* ptr; int32_t* idx; double* val; };     // ptr: m + 1
struct LufOut {
    int32_t* status;                // [0] LufStatus, [1] bump size, [2] peeled pivots, [3] entries of L, [4] of U (off-diagonal)
    int32_t* rowperm; int32_t* colperm;      // step -> row, step -> basis position
    int32_t* row_step; int32_t* col_step;    // the inverses
    double* diag;                   // m: U[k, k]
    LufTriangle Lf, Uf, Ub, Lb;     // rows of L (l < k), rows of U (l > k), columns of U (k < l), columns of L (k > l)
    int32_t cap;                    // entries each of the four arrays can take
};

struct LufWork {
    int32_t* pos_p; int32_t* pos_a; int32_t* wrow_pos;     // provider column / artificial / row of a wrapped artificial -> basis position
    int32_t* rcount; int32_t* ccount;                      // active entries per row / basis position
    int32_t* claim; int32_t* claim2;                       // m each: a row / a basis position taken as pivot
    int32_t* list; int32_t* list2; int32_t* piv;           // m each
    int32_t* part;                                         // threads + 1: partial counts of luf_select
    int32_t* brow; int32_t* bcol; int32_t* lrow; int32_t* lcol;      // bump: local -> row / position, and back (m each)
    int32_t* brc; int32_t* bcc; int32_t* ract; int32_t* cact;        // nb each
    int32_t* bstep_row; int32_t* bstep_col;                          // nb: elimination step of a local row / column
    int32_t* I; int32_t* J; double* fmul;                            // nb each
    unsigned long long* red;                                         // 8 words: reductions, 64 words: column maxima of the candidates
    double* D; int32_t nb_cap;                                       // dense bump, nb_cap x nb_cap at most
    int32_t* scalars;                                                // 16 ints of uniform state
};

#if defined(RELP_LUF_DEVICE)
#define LUF_FN __device__ __forceinline__
#define LUF_NT ((int)blockDim.x)
#define LUF_TID ((int)threadIdx.x)
#define PAR_FOR(i, n) for (int i = LUF_TID; i < (n); i += LUF_NT)
#define PAR_END __syncthreads();
#define LUF_SINGLE if (LUF_TID == 0)
// Global atomics execute in L2 and leave the CU's L1 alone: a word that is ever touched by an atomic is read and written
// through L2 as well (agent-scope atomic load / store), never by a plain access that could hit a stale L1 line.
LUF_FN void luf_add(int32_t* p, int32_t v) { atomicAdd(p, v); }
LUF_FN int32_t luf_cas(int32_t* p, int32_t expect, int32_t v) { return atomicCAS(p, expect, v); }
LUF_FN void luf_min64(unsigned long long* p, unsigned long long v) { atomicMin(p, v); }
LUF_FN void luf_max64(unsigned long long* p, unsigned long long v) { atomicMax(p, v); }
LUF_FN void luf_max32(int32_t* p, int32_t v) { atomicMax(p, v); }
LUF_FN void luf_min32(int32_t* p, int32_t v) { atomicMin(p, v); }
LUF_FN void luf_or(int32_t* p, int32_t v) { atomicOr(p, v); }
LUF_FN int32_t luf_ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LUF_FN void luf_st(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LUF_FN unsigned long long luf_ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LUF_FN void luf_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
#define LUF_FN inline
#define LUF_NT 64
#define LUF_TID 0
#define PAR_FOR(i, n) for (int i = 0; i < (n); ++i)
#define PAR_END
#define LUF_SINGLE
LUF_FN void luf_add(int32_t* p, int32_t v) { *p += v; }
LUF_FN int32_t luf_cas(int32_t* p, int32_t expect, int32_t v) { const int32_t o = *p; if (o == expect) *p = v; return o; }
LUF_FN void luf_min64(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
LUF_FN void luf_max64(unsigned long long* p, unsigned long long v) { if (v > *p) *p = v; }
LUF_FN void luf_max32(int32_t* p, int32_t v) { if (v > *p) *p = v; }
LUF_FN void luf_min32(int32_t* p, int32_t v) { if (v < *p) *p = v; }
LUF_FN void luf_or(int32_t* p, int32_t v) { *p |= v; }
LUF_FN int32_t luf_ld(const int32_t* p) { return *p; }
LUF_FN void luf_st(int32_t* p, int32_t v) { *p = v; }
LUF_FN unsigned long long luf_ld64(const unsigned long long* p) { return *p; }
LUF_FN void luf_st64(unsigned long long* p, unsigned long long v) { *p = v; }
#endif

LUF_FN unsigned long long luf_bits(double v) {      // |v| as an integer that orders like the value
    union { double d; unsigned long long u; } x;
    x.d = v < 0 ? -v : v;
    return x.u;
}
LUF_FN double luf_abs(double v) { return v < 0 ? -v : v; }
LUF_FN double luf_from_bits(unsigned long long u) { union { double d; unsigned long long u; } x; x.u = u; return x.d; }

// entries of basis column j (engine numbering) -> f(row, value)
template <class F>
LUF_FN void luf_col_entries(const LufMatrix& M, int32_t j, F f) {
    if (j >= kWrappedArtificialBase) { f(M.wrapped_row[M.wrapped_na - 1 - (INT32_MAX - j)], 1.0); return; }
    if (j < M.na) { f(M.ct.column_to_row[j], 1.0); return; }
    const int32_t p = j - M.na;
    if (p < M.ct.nr_normal) {
        for (int64_t e = M.csc.col_ptr[p]; e < M.csc.col_ptr[p + 1]; ++e) f(M.csc.row_idx[e], M.csc.values[e]);
        if (M.ct.bound_row[p] >= 0) f(M.ct.bound_row[p], 1.0);
    } else {
        const int32_t v = p - M.ct.nr_normal;
        if (M.ct.vrow0[v] >= 0) f(M.ct.vrow0[v], (double)M.ct.vsign[v]);
        if (M.ct.vrow1[v] >= 0) f(M.ct.vrow1[v], 1.0);
    }
}
// entries of row i of the basis matrix -> f(basis position, value), in the order of the static row-major copy
template <class F>
LUF_FN void luf_row_entries(const LufMatrix& M, const LufWork& W, int32_t i, F f) {
    for (int32_t e = M.rptr[i]; e < M.rptr[i + 1]; ++e) { const int32_t c = W.pos_p[M.rcol[e]]; if (c >= 0) f(c, M.rval[e]); }
    const int32_t a = M.art_of_row[i];
    if (a >= 0 && W.pos_a[a] >= 0) f(W.pos_a[a], 1.0);
    if (W.wrow_pos[i] >= 0) f(W.wrow_pos[i], 1.0);
}

#if defined(RELP_LUF_DEVICE)
// exclusive prefix sum of one value per thread over the workgroup (<= 1,024 threads); *total = the sum.  Two barriers.
LUF_FN int32_t luf_block_exscan(int32_t v, int32_t* total) {
    __shared__ int32_t s_wave[17];
    const int lane = LUF_TID & 63, wave = LUF_TID >> 6, nw = (LUF_NT + 63) >> 6;
    int32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    __syncthreads();                                   // (s_wave may still be read by the previous call)
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int32_t before = 0, all = 0;
    for (int w = 0; w < nw; ++w) { const int32_t c = s_wave[w]; if (w < wave) before += c; all += c; }
    *total = all;
    return before + inc - v;
}
#endif

// Ordered compaction: out = { i in [0, n) : pred(i) } ascending; returns the count (uniform).  Every thread owns a contiguous
// chunk, so the order does not depend on the execution.
template <class P>
LUF_FN int32_t luf_select(int32_t n, P pred, int32_t* out, const LufWork& W) {
    (void)W;
#if defined(RELP_LUF_DEVICE)
    const int32_t nt = LUF_NT, chunk = (n + nt - 1) / nt;
    const int32_t t = LUF_TID, lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    int32_t cnt = 0;
    for (int32_t i = lo; i < hi; ++i) cnt += pred(i) ? 1 : 0;
    int32_t total = 0;
    int32_t at = luf_block_exscan(cnt, &total);
    for (int32_t i = lo; i < hi; ++i) if (pred(i)) out[at++] = i;
    __syncthreads();
    return total;
#else
    int32_t at = 0;
    for (int32_t i = 0; i < n; ++i) if (pred(i)) out[at++] = i;
    return at;
#endif
}

// counts a[1 .. n] -> offsets: a[0] = 0, a[k + 1] = a[1] + .. + a[k + 1]; returns the total (uniform).  In place.
LUF_FN int32_t luf_offsets_from_counts(int32_t* a, int32_t n) {
#if defined(RELP_LUF_DEVICE)
    const int32_t nt = LUF_NT, chunk = (n + nt - 1) / nt;
    const int32_t t = LUF_TID, lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    int32_t sum = 0;
    for (int32_t i = lo; i < hi; ++i) sum += a[i + 1];
    int32_t total = 0;
    int32_t run = luf_block_exscan(sum, &total);
    for (int32_t i = lo; i < hi; ++i) { run += a[i + 1]; a[i + 1] = run; }
    if (t == 0) a[0] = 0;
    __syncthreads();
    return total;
#else
    a[0] = 0;
    for (int32_t i = 0; i < n; ++i) a[i + 1] += a[i];
    return a[n];
#endif
}

LUF_FN void luf_fail(const LufOut& O, int32_t why) { LUF_SINGLE { O.status[0] = why; } PAR_END }

// ---------------------------------------------------------------------------------------------------------------------
// The factorisation.  `basis`: m column ids (engine numbering).
// ---------------------------------------------------------------------------------------------------------------------
LUF_FN void luf_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O) {
    const int32_t m = M.m;
    constexpr double kThreshold = 0.1;
    // ---- maps: who is basic where ----------------------------------------------------------------------------------------
    PAR_FOR(p, M.n_provider) W.pos_p[p] = -1; PAR_END
    PAR_FOR(a, M.na) W.pos_a[a] = -1; PAR_END
    PAR_FOR(i, m) {
        W.wrow_pos[i] = -1; O.row_step[i] = -1; O.col_step[i] = -1; W.lrow[i] = -1; W.lcol[i] = -1;
        luf_st(&W.claim[i], -1); luf_st(&W.claim2[i], -1);
    } PAR_END
    LUF_SINGLE { O.status[0] = LUF_OK; O.status[1] = 0; O.status[2] = 0; O.status[3] = 0; O.status[4] = 0; } PAR_END
    PAR_FOR(c, m) {
        const int32_t j = basis[c];
        if (j >= kWrappedArtificialBase) W.wrow_pos[M.wrapped_row[M.wrapped_na - 1 - (INT32_MAX - j)]] = c;
        else if (j < M.na) W.pos_a[j] = c;
        else W.pos_p[j - M.na] = c;
    } PAR_END
    PAR_FOR(c, m) { int32_t n = 0; luf_col_entries(M, basis[c], [&](int32_t, double) { ++n; }); luf_st(&W.ccount[c], n); } PAR_END
    PAR_FOR(i, m) { int32_t n = 0; luf_row_entries(M, W, i, [&](int32_t, double) { ++n; }); luf_st(&W.rcount[i], n); } PAR_END

    // ---- singletons, round by round ----------------------------------------------------------------------------------------
    int32_t k = 0;
    for (;;) {
        int32_t made = 0;
        // columns with one active entry: fill-free, no multipliers, always acceptable
        const int32_t n1 = luf_select(m, [&](int32_t c) { return O.col_step[c] < 0 && luf_ld(&W.ccount[c]) == 1; }, W.list, W);
        if (n1 > 0) {
            PAR_FOR(t, n1) {
                const int32_t c = W.list[t];
                int32_t row = -1; double val = 0.0;
                luf_col_entries(M, basis[c], [&](int32_t i, double v) { if (O.row_step[i] < 0) { row = i; val = v; } });
                W.piv[t] = row;
                if (row >= 0) {
                    if (luf_cas(&W.claim[row], -1, c) != -1) W.piv[t] = -2;        // two columns on one row: singular
                    else O.diag[k + t] = val;
                }
            } PAR_END
            PAR_FOR(t, n1) { if (W.piv[t] < 0) O.status[0] = LUF_SINGULAR; } PAR_END
            if (O.status[0] != LUF_OK) return;
            PAR_FOR(t, n1) {
                const int32_t c = W.list[t], i = W.piv[t];
                O.row_step[i] = k + t; O.col_step[c] = k + t; O.rowperm[k + t] = i; O.colperm[k + t] = c;
            } PAR_END
            PAR_FOR(t, n1) {                                 // the pivot rows leave: their other columns lose an active entry
                luf_row_entries(M, W, W.piv[t], [&](int32_t c, double) { if (O.col_step[c] < 0) luf_add(&W.ccount[c], -1); });
            } PAR_END
            k += n1; made += n1;
        }
        // rows with one active entry: fill-free; the other active rows of the column become multipliers, so the entry must
        // pass the threshold against the column's largest active entry
        const int32_t n2 = luf_select(m, [&](int32_t i) { return O.row_step[i] < 0 && luf_ld(&W.rcount[i]) == 1; }, W.list, W);
        if (n2 > 0) {
            PAR_FOR(t, n2) {
                const int32_t i = W.list[t];
                int32_t col = -1; double val = 0.0;
                luf_row_entries(M, W, i, [&](int32_t c, double v) { if (O.col_step[c] < 0) { col = c; val = v; } });
                W.piv[t] = -1;
                if (col >= 0) {
                    double cmax = 0.0;
                    luf_col_entries(M, basis[col], [&](int32_t r, double v) { if (O.row_step[r] < 0 && luf_abs(v) > cmax) cmax = luf_abs(v); });
                    if (luf_abs(val) >= kThreshold * cmax && val != 0.0) W.piv[t] = col;
                }
            } PAR_END
            const int32_t n2a = luf_select(n2, [&](int32_t t) { return W.piv[t] >= 0; }, W.list2, W);
            if (n2a > 0) {
                PAR_FOR(u, n2a) {
                    const int32_t t = W.list2[u], i = W.list[t], c = W.piv[t];
                    if (luf_cas(&W.claim2[c], -1, i) != -1) O.status[0] = LUF_SINGULAR;             // two rows on one column
                } PAR_END
                if (O.status[0] != LUF_OK) return;
                PAR_FOR(u, n2a) {
                    const int32_t t = W.list2[u], i = W.list[t], c = W.piv[t];
                    double val = 0.0;
                    luf_row_entries(M, W, i, [&](int32_t cc, double v) { if (cc == c) val = v; });
                    O.row_step[i] = k + u; O.col_step[c] = k + u; O.rowperm[k + u] = i; O.colperm[k + u] = c; O.diag[k + u] = val;
                } PAR_END
                PAR_FOR(u, n2a) {                            // the pivot columns leave: their other rows lose an active entry
                    const int32_t c = W.piv[W.list2[u]];
                    luf_col_entries(M, basis[c], [&](int32_t r, double) { if (O.row_step[r] < 0) luf_add(&W.rcount[r], -1); });
                } PAR_END
                k += n2a; made += n2a;
            }
        }
        if (made == 0) break;
    }
    const int32_t k_peel = k;

    // ---- the bump: dense working copy ---------------------------------------------------------------------------------------
    const int32_t nbr = luf_select(m, [&](int32_t i) { return O.row_step[i] < 0; }, W.brow, W);
    const int32_t nbc = luf_select(m, [&](int32_t c) { return O.col_step[c] < 0; }, W.bcol, W);
    if (nbr != nbc) { luf_fail(O, LUF_SINGULAR); return; }
    const int32_t nb = nbr;
    LUF_SINGLE { O.status[1] = nb; O.status[2] = k_peel; } PAR_END
    if (nb > W.nb_cap) { luf_fail(O, LUF_BUMP_TOO_LARGE); return; }
    double* const D = W.D;
    if (nb > 0) {
        PAR_FOR(t, nb) { W.lrow[W.brow[t]] = t; W.lcol[W.bcol[t]] = t; W.ract[t] = 1; W.cact[t] = 1; } PAR_END
        PAR_FOR(e, nb * nb) D[e] = 0.0; PAR_END
        PAR_FOR(t, nb) {
            int32_t n = 0;
            luf_row_entries(M, W, W.brow[t], [&](int32_t c, double v) {
                const int32_t u = W.lcol[c];
                if (u >= 0 && v != 0.0) { D[(int64_t)t * nb + u] += v; }
            });
            for (int32_t u = 0; u < nb; ++u) if (D[(int64_t)t * nb + u] != 0.0) ++n;
            luf_st(&W.brc[t], n);
        } PAR_END
        PAR_FOR(u, nb) { int32_t n = 0; for (int32_t t = 0; t < nb; ++t) if (D[(int64_t)t * nb + u] != 0.0) ++n; luf_st(&W.bcc[u], n); } PAR_END
    }
    for (int32_t s = 0; s < nb; ++s) {
        // the sparsest active row (ties: the lower index)
        LUF_SINGLE { luf_st64(&W.red[0], ~0ull); luf_st64(&W.red[1], ~0ull); } PAR_END
        PAR_FOR(t, nb) { if (W.ract[t]) luf_min64(&W.red[0], ((unsigned long long)(uint32_t)luf_ld(&W.brc[t]) << 32) | (uint32_t)t); } PAR_END
        const unsigned long long rkey = luf_ld64(&W.red[0]);
        if (rkey == ~0ull || (rkey >> 32) == 0) { luf_fail(O, LUF_SINGULAR); return; }
        const int32_t ra = (int32_t)(rkey & 0xffffffffu);
        // its entries, by ascending column count, against the threshold
        const int32_t nc = luf_select(nb, [&](int32_t u) { return W.cact[u] && D[(int64_t)ra * nb + u] != 0.0; }, W.J, W);
        const int32_t ncc = nc < 56 ? nc : 56;              // (column maxima of the first 56 candidates; a sparsest row is short)
        LUF_SINGLE { for (int32_t q = 0; q < ncc; ++q) luf_st64(&W.red[8 + q], 0ull); } PAR_END
        PAR_FOR(e, ncc * nb) {
            const int32_t q = e / nb, t = e % nb;
            if (W.ract[t]) { const double v = D[(int64_t)t * nb + W.J[q]]; if (v != 0.0) luf_max64(&W.red[8 + q], luf_bits(v)); }
        } PAR_END
        PAR_FOR(q, ncc) {
            const int32_t u = W.J[q];
            if (luf_bits(D[(int64_t)ra * nb + u]) >= luf_bits(kThreshold * luf_from_bits(luf_ld64(&W.red[8 + q]))))
                luf_min64(&W.red[1], ((unsigned long long)(uint32_t)luf_ld(&W.bcc[u]) << 32) | (uint32_t)u);
        } PAR_END
        int32_t pr = ra, pc;
        const unsigned long long ckey = luf_ld64(&W.red[1]);
        if (ckey != ~0ull) {
            pc = (int32_t)(ckey & 0xffffffffu);
        } else {
            // nothing in the row passes: the largest entry of the row's first column (always acceptable)
            pc = W.J[0];
            const unsigned long long cmax0 = luf_ld64(&W.red[8]);
            PAR_FOR(t, nb) { if (W.ract[t] && luf_bits(D[(int64_t)t * nb + pc]) == cmax0) luf_min64(&W.red[1], (unsigned long long)(uint32_t)t); } PAR_END
            pr = (int32_t)(luf_ld64(&W.red[1]) & 0xffffffffu);
        }
        const double pv = D[(int64_t)pr * nb + pc];
        const int32_t nj = luf_select(nb, [&](int32_t u) { return W.cact[u] && u != pc && D[(int64_t)pr * nb + u] != 0.0; }, W.J, W);
        const int32_t ni = luf_select(nb, [&](int32_t t) { return W.ract[t] && t != pr && D[(int64_t)t * nb + pc] != 0.0; }, W.I, W);
        PAR_FOR(a, ni) { const int64_t at = (int64_t)W.I[a] * nb + pc; const double f = D[at] / pv; D[at] = f; W.fmul[a] = f; } PAR_END
        PAR_FOR(e, ni * nj) {
            const int32_t a = e / nj, b = e % nj, t = W.I[a], u = W.J[b];
            const int64_t at = (int64_t)t * nb + u;
            const double old = D[at], nw = old - W.fmul[a] * D[(int64_t)pr * nb + u];
            D[at] = nw;
            if (old == 0.0 && nw != 0.0) { luf_add(&W.brc[t], 1); luf_add(&W.bcc[u], 1); }
            else if (old != 0.0 && nw == 0.0) { luf_add(&W.brc[t], -1); luf_add(&W.bcc[u], -1); }      // exact cancellation
        } PAR_END
        PAR_FOR(b, nj) luf_add(&W.bcc[W.J[b]], -1); PAR_END          // the pivot row leaves ...
        PAR_FOR(a, ni) luf_add(&W.brc[W.I[a]], -1); PAR_END          // ... and the pivot column
        LUF_SINGLE {
            W.ract[pr] = 0; W.cact[pc] = 0; W.bstep_row[pr] = s; W.bstep_col[pc] = s;
            const int32_t i = W.brow[pr], c = W.bcol[pc];
            O.row_step[i] = k_peel + s; O.col_step[c] = k_peel + s; O.rowperm[k_peel + s] = i; O.colperm[k_peel + s] = c;
            O.diag[k_peel + s] = pv;
        } PAR_END
    }

    // ---- L and U in pivot coordinates, row-wise and column-wise; every row / column is written by ONE thread in a fixed order --
    // (two passes: count, then fill behind a running sum)
    for (int pass = 0; pass < 2; ++pass) {
        PAR_FOR(kk, m) {
            const int32_t i = O.rowperm[kk], t = W.lrow[i];
            int32_t nl = 0, nu = 0;
            const int32_t bl = pass ? O.Lf.ptr[kk] : 0, bu = pass ? O.Uf.ptr[kk] : 0;
            luf_row_entries(M, W, i, [&](int32_t c, double v) {
                const int32_t kc = O.col_step[c];
                if (t >= 0 && W.lcol[c] >= 0) return;                      // bump x bump: from D below
                if (kc > kk) { if (pass) { O.Uf.idx[bu + nu] = kc; O.Uf.val[bu + nu] = v; } ++nu; }
                else if (kc < kk) { if (pass) { O.Lf.idx[bl + nl] = kc; O.Lf.val[bl + nl] = v / O.diag[kc]; } ++nl; }
            });
            if (t >= 0) {
                for (int32_t u = 0; u < nb; ++u) {
                    const double v = D[(int64_t)t * nb + u];
                    if (v == 0.0) continue;
                    const int32_t kc = k_peel + W.bstep_col[u];
                    if (kc > kk) { if (pass) { O.Uf.idx[bu + nu] = kc; O.Uf.val[bu + nu] = v; } ++nu; }
                    else if (kc < kk) { if (pass) { O.Lf.idx[bl + nl] = kc; O.Lf.val[bl + nl] = v; } ++nl; }
                }
            }
            if (!pass) { O.Lf.ptr[kk + 1] = nl; O.Uf.ptr[kk + 1] = nu; }
        } PAR_END
        PAR_FOR(kk, m) {
            const int32_t c = O.colperm[kk], u = W.lcol[c];
            int32_t nl = 0, nu = 0;
            const int32_t bl = pass ? O.Lb.ptr[kk] : 0, bu = pass ? O.Ub.ptr[kk] : 0;
            luf_col_entries(M, basis[c], [&](int32_t i, double v) {
                const int32_t kr = O.row_step[i];
                if (u >= 0 && W.lrow[i] >= 0) return;
                if (kr < kk) { if (pass) { O.Ub.idx[bu + nu] = kr; O.Ub.val[bu + nu] = v; } ++nu; }
                else if (kr > kk) { if (pass) { O.Lb.idx[bl + nl] = kr; O.Lb.val[bl + nl] = v / O.diag[kk]; } ++nl; }
            });
            if (u >= 0) {
                for (int32_t t = 0; t < nb; ++t) {
                    const double v = D[(int64_t)t * nb + u];
                    if (v == 0.0) continue;
                    const int32_t kr = k_peel + W.bstep_row[t];
                    if (kr < kk) { if (pass) { O.Ub.idx[bu + nu] = kr; O.Ub.val[bu + nu] = v; } ++nu; }
                    else if (kr > kk) { if (pass) { O.Lb.idx[bl + nl] = kr; O.Lb.val[bl + nl] = v; } ++nl; }
                }
            }
            if (!pass) { O.Lb.ptr[kk + 1] = nl; O.Ub.ptr[kk + 1] = nu; }
        } PAR_END
        if (!pass) {
            int32_t* const ptrs[4] = {O.Lf.ptr, O.Uf.ptr, O.Ub.ptr, O.Lb.ptr};
            int32_t tot[4];
            for (int q = 0; q < 4; ++q) tot[q] = luf_offsets_from_counts(ptrs[q], m);
            LUF_SINGLE {
                for (int q = 0; q < 4; ++q) if (tot[q] > O.cap) O.status[0] = LUF_NO_ROOM;
                O.status[3] = tot[0]; O.status[4] = tot[1];
            } PAR_END
            if (O.status[0] != LUF_OK) return;
        }
    }
}

}  // namespace relp

#if !defined(RELP_LUF_DEVICE)
namespace relp {
// relp_kernels_luf.hip: the same algorithm as one workgroup of luf_threads() threads on stream s
void launch_lu_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O, hipStream_t s);
int32_t luf_threads();
}  // namespace relp
#endif
