// relp_lu_factor_core.h -- P B Q = L U of the basis ON THE DEVICE: one workgroup, everything it
// touches in global memory or LDS, no host round trip.  Reference: the right-looking elimination with Markowitz pivoting of
// carry/lower_upper/decomposition/mod.rs:27-138 and decomposition/pivoting.rs:45-81 -- like the host factorisation
// (relp_lu.cpp: lu_factor) with singletons first and a threshold of 0.1 on the bump that is left.
//
// How it maps to a GPU.  A basis of an LP is mostly triangular: columns / rows with ONE active entry are fill-free pivots,
// and all singletons of a round are independent of each other, so a round finds and numbers them in parallel
// (`luf_select`: an ordered compaction, so the numbering is deterministic).  What survives the peeling -- the "bump" -- is
// eliminated on SPARSE rows (lists of (column, value) in one arena, a row that outgrows its room moves to the arena's end) in
// ROUNDS OF MUTUALLY INDEPENDENT PIVOTS [r4]: every round prices every active entry with the Markowitz count
// (r - 1)(c - 1) of pivoting.rs:45-81 (all entries, not the sparsest row only) under the threshold test against the column
// maximum; every row proposes its best entry; all proposals that do not touch each other
// -- pivots (i1, j1), (i2, j2) with a[i1, j2] = a[i2, j1] = 0, found by two atomic minima per column over proposal
// priorities -- are eliminated together: a row of the active submatrix is rewritten by ONE thread, which applies the round's
// pivots that reach it in ascending order, so the arithmetic does not depend on the execution.  A mid-solve basis of Netlib
// 25FV47 (bump 455 of 790) takes ~40 rounds instead of 455 sequential steps.  Exact zeros are never stored (the reference drops
// them, decomposition/mod.rs:178).  Nothing is recorded during the peeling: no entry of a peeled row or column is ever
// modified, so afterwards L and U outside the bump are read off the basis columns by the step numbers alone -- entry (i, c)
// is U[k_i, k_c] when k_i < k_c, the multiplier L[k_i, k_c] = v / d_{k_c} when k_i > k_c, the diagonal when equal -- and the
// bump block off the rows as the elimination left them plus the list of multipliers.
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
    // the bump as sparse rows: row t = entries [rbeg[t], rbeg[t] + rlen[t]) of (ecol, eval), room for rcap[t]
    int32_t nb_cap;                                                  // local rows / columns the arrays below can take
    int32_t* rbeg; int32_t* rlen; int32_t* rcap;                     // nb each
    int32_t* ract; int32_t* cact;                                    // nb: still active
    int32_t* bcc;                                                    // nb: active entries per local column (atomic)
    int32_t* bstep_row; int32_t* bstep_col;                          // nb: elimination step of a local row / column
    int32_t* cpiv;                                                   // nb: local column the row proposes as its pivot (-1: none / not accepted)
    int32_t* prank;                                                  // nb, by local column: rank of the pivot accepted in this round, or -1
    int32_t* acc;                                                    // nb: the accepted rows of the round, ascending
    double* pval;                                                    // nb: their pivot values, by rank
    unsigned long long* cmax;                                        // nb, by column: bits of the largest active |entry|
    unsigned long long* rowmark;                                     // nb, by column: best priority among the proposing rows with an entry there
    unsigned long long* colbest;                                     // nb, by column: best priority among the rows proposing it
    unsigned long long* cprio;                                       // nb, by row: Markowitz count << 32 | row (~0: nothing to propose)
    int32_t* ecol; double* eval; int32_t arena_cap;                  // the arena of the rows
    int32_t* lt_row; int32_t* lt_step; double* lt_val; int32_t lt_cap;     // multipliers (local row, step, value), in no particular order
    int32_t* lt_ptr; int32_t* lt_ord;                                // nb + 1 / lt_cap: room sizes of the bump's rows (setup) / spare
    int32_t* ut_row; int32_t* ut_col; double* ut_val;                // lt_cap each: the triplets of U (those of L join the multipliers)
    int32_t* vw;                                                     // m + 2: counts / cursors of a view
    int32_t* vtmp; int32_t vtmp_cap;                                 // the buckets of a view (a factor with more entries: LUF_NO_ROOM)
    int32_t* counters;                                               // [0] arena top, [1] multipliers, [2] rounds of the bump
    unsigned long long* red;                                         // 8 words of reductions
    int32_t* scalars;                                                // 16 ints of uniform state
    // the dense finish: once <= dense_cap (<= 64) rows are active they go into a dense block (the tail of a bump is a small
    // dense corner that yields one pivot per round); 0 = off
    double* dense; int32_t* dint; int32_t dense_cap;                 // dense_cap^2 doubles; 8 * dense_cap ints
};

#if defined(RELP_LUF_DEVICE)
#define LUF_FN __device__ __forceinline__
#define LUF_NT ((int)blockDim.x)
#define LUF_TID ((int)threadIdx.x)
#define PAR_FOR(i, n) for (int i = LUF_TID; i < (n); i += LUF_NT)
// (i, j) over ni x nj with nj <= 64: a wave per row, a lane per column
#define PAR_FOR2(i, j, ni, nj) for (int i = LUF_TID >> 6; i < (ni); i += LUF_NT >> 6) for (int j = LUF_TID & 63; j < (nj); j += 64)
// Global atomics execute in L2 and leave the CU's L1 alone: a word that is ever touched by an atomic is read and written
// through L2 as well (luf_ld / luf_st below), never by a plain access that could hit a stale L1 line.  RELP_LUF_FENCES wraps
// every barrier in agent-scope fences instead (L1 invalidated at every phase: a debugging aid, several times slower).
#if defined(RELP_LUF_FENCES)
#define PAR_END __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
#define PAR_END __syncthreads();
#endif
#define LUF_SINGLE if (LUF_TID == 0)
LUF_FN void luf_add(int32_t* p, int32_t v) { atomicAdd(p, v); }
LUF_FN int32_t luf_fetch_add(int32_t* p, int32_t v) { return atomicAdd(p, v); }
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
#define PAR_FOR2(i, j, ni, nj) for (int i = 0; i < (ni); ++i) for (int j = 0; j < (nj); ++j)
#define PAR_END
#define LUF_SINGLE
LUF_FN void luf_add(int32_t* p, int32_t v) { *p += v; }
LUF_FN int32_t luf_fetch_add(int32_t* p, int32_t v) { const int32_t o = *p; *p += v; return o; }
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

// phase clocks (thread 0, device only): counters[8 + 2 i], [9 + 2 i] = low / high word of the clocks of phase i
#if defined(RELP_LUF_DEVICE)
#define LUF_LAP_AT(base, i) do { if (LUF_TID == 0) { const long long now_ = (long long)__builtin_readcyclecounter(); const long long d_ = now_ - lap_; lap_ = now_; \
    unsigned long long* p_ = reinterpret_cast<unsigned long long*>(base) + (i); *p_ += (unsigned long long)d_; } } while (0)
#define LUF_LAP(i) LUF_LAP_AT(W.counters + 8, i)
#define LUF_LAP_BEGIN long long lap_ = (long long)__builtin_readcyclecounter();
#else
#define LUF_LAP(i) do {} while (0)
#define LUF_LAP_AT(base, i) do {} while (0)
#define LUF_LAP_BEGIN
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
LUF_FN int32_t luf_select(int32_t n, P pred, int32_t* out) {
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

// minimum over the workgroup of one value per thread, into *dst (preset to ~0 behind a barrier): a shuffle tree per wave, then one
// atomic per wave
LUF_FN void luf_block_min64(unsigned long long v, unsigned long long* dst) {
#if defined(RELP_LUF_DEVICE)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int lo = __shfl_xor((unsigned int)v, off, 64), hi = __shfl_xor((unsigned int)(v >> 32), off, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        if (o < v) v = o;
    }
    if ((LUF_TID & 63) == 0 && v != ~0ull) atomicMin(dst, v);
#else
    if (v < *dst) *dst = v;
#endif
}

// inside PAR_FOR2(i, j, ..): the number of columns j of row i with `nz`, into *dst (zero before the loop).  On the device the lanes
// of the wave that holds row i count by ballot (64 lanes adding to ONE LDS word serialise: 13,000 clocks per step of the dense
// finish, measured).
LUF_FN void luf_row_tally(int32_t* dst, bool nz) {
#if defined(RELP_LUF_DEVICE)
    const unsigned long long in = __ballot(1), b = __ballot(nz);
    if ((LUF_TID & 63) == __ffsll(in) - 1) luf_st(dst, (int32_t)__popcll(b));
#else
    if (nz) ++*dst;
#endif
}

LUF_FN void luf_fail(const LufOut& O, int32_t why) { LUF_SINGLE { O.status[0] = why; } PAR_END }

// ---------------------------------------------------------------------------------------------------------------------
// The factorisation.  `basis`: m column ids (engine numbering).
// ---------------------------------------------------------------------------------------------------------------------
// Part 1: maps, counts, singletons.  Returns the number of peeled pivots, or -1 when the basis is singular (status set).
LUF_FN int32_t luf_peel(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O) {
    const int32_t m = M.m;
    constexpr double kThreshold = 0.1;
    LUF_LAP_BEGIN
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

    LUF_LAP(0);
    // ---- singletons, round by round ----------------------------------------------------------------------------------------
    int32_t k = 0;
    for (;;) {
        int32_t made = 0;
        // columns with one active entry: fill-free, no multipliers, always acceptable
        const int32_t n1 = luf_select(m, [&](int32_t c) { return O.col_step[c] < 0 && luf_ld(&W.ccount[c]) == 1; }, W.list);
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
            if (O.status[0] != LUF_OK) return -1;
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
        const int32_t n2 = luf_select(m, [&](int32_t i) { return O.row_step[i] < 0 && luf_ld(&W.rcount[i]) == 1; }, W.list);
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
            const int32_t n2a = luf_select(n2, [&](int32_t t) { return W.piv[t] >= 0; }, W.list2);
            if (n2a > 0) {
                PAR_FOR(u, n2a) {
                    const int32_t t = W.list2[u], i = W.list[t], c = W.piv[t];
                    if (luf_cas(&W.claim2[c], -1, i) != -1) O.status[0] = LUF_SINGULAR;             // two rows on one column
                } PAR_END
                if (O.status[0] != LUF_OK) return -1;
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
    LUF_LAP(1);
    return k;
}

// The dense finish of the bump.  What is left after a few rounds is a small corner that fills in completely and yields one pivot
// per round (25FV47 mid-solve, bump 401: 64 rows after 11 rounds, then 25 rounds for them): with <= dense_cap rows active the
// block goes into a dense array and is eliminated step by step, every step the entry of lowest Markowitz count under the same
// threshold test (ties: the lower column, then the lower row), a wave per row.  Afterwards the multipliers join the list and
// every row's part of U goes back into the arena, so the views below read one format.
LUF_FN int32_t luf_dense_finish(const LufWork& W, const LufOut& O, const int32_t k_peel, const int32_t nb, const int32_t done) {
    constexpr double kThreshold = 0.1;
    const int32_t cap = W.dense_cap;
    int32_t* drow = W.dint; int32_t* dcol = W.dint + cap; int32_t* drc = W.dint + 2 * cap; int32_t* dcc = W.dint + 3 * cap;
    int32_t* dsr = W.dint + 4 * cap; int32_t* dsc = W.dint + 5 * cap;      // step of a dense row / column, -1 while active
    int32_t* dpi = W.dint + 6 * cap; int32_t* dpj = W.dint + 7 * cap;      // dense row / column of a step
    const int32_t na = luf_select(nb, [&](int32_t t) { return W.ract[t] != 0; }, drow);
    const int32_t nc = luf_select(nb, [&](int32_t u) { return W.cact[u] != 0; }, dcol);
    if (na != nc) return LUF_SINGULAR;
    double* D = W.dense;
    PAR_FOR(j, na) { W.prank[dcol[j]] = j; dsr[j] = -1; dsc[j] = -1; luf_st(&drc[j], 0); luf_st(&dcc[j], 0); luf_st64(&W.cmax[j], 0ull); }
    PAR_FOR(e, na * na) D[e] = 0.0;
    LUF_SINGLE { luf_st64(&W.red[0], ~0ull); luf_st64(&W.red[1], ~0ull); luf_st(&W.scalars[0], 0); } PAR_END
    PAR_FOR(i, na) {
        const int32_t t = drow[i], b = W.rbeg[t], n = W.rlen[t];
        for (int32_t e = b; e < b + n; ++e) D[i * na + W.prank[W.ecol[e]]] = W.eval[e];
    } PAR_END
#if defined(RELP_LUF_DEVICE)
    // The block in REGISTERS: wave w holds rows w, w + 8, .., lane j column j.  A step is three barriers: the waves' partial column
    // counts / maxima through LDS, the best entry (a shuffle tree per wave, one atomic per wave), the pivot row through LDS; the
    // multipliers are read across the wave's lanes.  (The loop on the LDS copy below took 14,000 clocks per step: every one of
    // its eight rows per wave a chain of six dependent LDS reads.)
    if ((LUF_NT >> 6) == 8 && W.nb_cap >= 512) {
        const int lane = LUF_TID & 63, wave = LUF_TID >> 6;
        constexpr int R = 8;
        int32_t* pc_i = reinterpret_cast<int32_t*>(W.rowmark);             // [8][64] partial counts
        double* pm_d = reinterpret_cast<double*>(W.cmax);                   // [8][64] partial maxima
        double* prow = reinterpret_cast<double*>(W.colbest);                // [64] the pivot row
        double v[R];
        uint32_t rmask = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int i = wave + 8 * k;
            v[k] = (i < na && lane < na) ? D[i * na + lane] : 0.0;
            if (i < na) rmask |= 1u << k;
        }
        bool cact = lane < na;
        __syncthreads();                                                   // (cmax was zeroed above: now the partial maxima live there)
        for (int32_t s = 0; s < na; ++s) {
            int32_t pcnt = 0; double pmax = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) if ((rmask >> k) & 1u) { const double a = luf_abs(v[k]); pcnt += v[k] != 0.0 ? 1 : 0; pmax = a > pmax ? a : pmax; }
            pc_i[wave * 64 + lane] = pcnt; pm_d[wave * 64 + lane] = pmax;
            __syncthreads();
            int32_t cc = 0; double cmx = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2) { cc += pc_i[w2 * 64 + lane]; const double a = pm_d[w2 * 64 + lane]; cmx = a > cmx ? a : cmx; }
            const double thr = kThreshold * cmx;
            uint32_t best = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < R; ++k) if ((rmask >> k) & 1u) {
                const bool nz = cact && v[k] != 0.0;
                const int32_t rc = (int32_t)__popcll(__ballot(nz));
                if (nz && !(luf_abs(v[k]) < thr)) {
                    const uint32_t key = ((uint32_t)((rc - 1) * (cc - 1)) << 12) | ((uint32_t)lane << 6) | (uint32_t)(wave + 8 * k);
                    best = key < best ? key : best;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int32_t)best, off, 64); best = o < best ? o : best; }
            if (lane == 0 && best != 0xffffffffu) atomicMin(&W.red[s & 1], (unsigned long long)best);
            __syncthreads();
            const unsigned long long key64 = W.red[s & 1];
            if (key64 == ~0ull) return LUF_SINGULAR;
            const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)key64);
            const int pi = (int)(key & 63u), pj = (int)((key >> 6) & 63u), wp = pi & 7, kp = pi >> 3;
            if (wave == wp) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) if (k == kp) x = v[k];
                prow[lane] = x;
                rmask &= ~(1u << kp);
            }
            if (LUF_TID == 0) { W.red[(s + 1) & 1] = ~0ull; dsr[pi] = done + s; dsc[pj] = done + s; dpi[s] = pi; dpj[s] = pj; }
            __syncthreads();
            const double pw = prow[lane], pv = prow[pj];
            if (lane == pj) cact = false;
#pragma unroll
            for (int k = 0; k < R; ++k) if ((rmask >> k) & 1u) {
                union { double d; int32_t w[2]; } u; u.d = v[k];
                u.w[0] = __builtin_amdgcn_readlane(u.w[0], pj); u.w[1] = __builtin_amdgcn_readlane(u.w[1], pj);
                if (u.d != 0.0) {
                    const double f = u.d / pv;
                    if (lane == pj) v[k] = f;                                  // the multipliers stay where the column was
                    else if (cact && pw != 0.0) v[k] -= f * pw;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < R; ++k) { const int i = wave + 8 * k; if (i < na && lane < na) D[i * na + lane] = v[k]; }
        __syncthreads();
    } else
#endif
    {
    PAR_FOR2(i, j, na, na) {
        const double v = D[i * na + j];
        luf_row_tally(&drc[i], v != 0.0);
        if (v != 0.0) { luf_add(&dcc[j], 1); luf_max64(&W.cmax[j], luf_bits(v)); }
    } PAR_END
    for (int32_t s = 0; s < na; ++s) {
        unsigned long long best = ~0ull;
        PAR_FOR2(i, j, na, na) {
            if (dsr[i] >= 0 || dsc[j] >= 0) continue;
            const double v = D[i * na + j];
            if (v == 0.0) continue;
            if (luf_bits(v) < luf_bits(kThreshold * luf_from_bits(luf_ld64(&W.cmax[j])))) continue;
            const unsigned long long cost = (unsigned long long)(uint32_t)(luf_ld(&drc[i]) - 1) * (unsigned long long)(uint32_t)(luf_ld(&dcc[j]) - 1);
            const unsigned long long key = (cost << 12) | ((unsigned long long)(uint32_t)j << 6) | (uint32_t)i;
            if (key < best) best = key;
        }
        luf_block_min64(best, &W.red[s & 1]);
        PAR_END
        const unsigned long long key = luf_ld64(&W.red[s & 1]);
        if (key == ~0ull) return LUF_SINGULAR;
        const int32_t pi = (int32_t)(key & 63u), pj = (int32_t)((key >> 6) & 63u);
        const double pv = D[pi * na + pj];
        LUF_SINGLE { luf_st64(&W.red[(s + 1) & 1], ~0ull); dsr[pi] = done + s; dsc[pj] = done + s; dpi[s] = pi; dpj[s] = pj; }
        PAR_FOR(j, na) { luf_st(&drc[j], 0); luf_st(&dcc[j], 0); luf_st64(&W.cmax[j], 0ull); }
        PAR_FOR(i, na) {                                                       // the multipliers stay where the column was
            if (i == pi || dsr[i] >= 0) continue;
            const double v = D[i * na + pj];
            if (v != 0.0) D[i * na + pj] = v / pv;
        } PAR_END
        PAR_FOR2(i, j, na, na) {                                               // rank-1 update; the counts of the next step
            if (dsr[i] >= 0 || dsc[j] >= 0) continue;
            double v = D[i * na + j];
            const double f = D[i * na + pj];
            if (f != 0.0) { const double pw = D[pi * na + j]; if (pw != 0.0) { v -= f * pw; D[i * na + j] = v; } }
            luf_row_tally(&drc[i], v != 0.0);
            if (v != 0.0) { luf_add(&dcc[j], 1); luf_max64(&W.cmax[j], luf_bits(v)); }
        } PAR_END
    }
    }
    // the steps on record; the multipliers join the list, every row's part of U goes back into the arena
    PAR_FOR(s, na) {
        const int32_t pi = dpi[s], pj = dpj[s], t = drow[pi], u = dcol[pj];
        W.ract[t] = 0; W.cact[u] = 0; W.bstep_row[t] = done + s; W.bstep_col[u] = done + s; W.cpiv[t] = u;
        const int32_t i = W.brow[t], c = W.bcol[u], k = k_peel + done + s;
        O.row_step[i] = k; O.col_step[c] = k; O.rowperm[k] = i; O.colperm[k] = c; O.diag[k] = D[pi * na + pj];
    }
    PAR_FOR2(i, j, na, na) {
        if (dsc[j] >= dsr[i]) continue;
        const double v = D[i * na + j];
        if (v == 0.0) continue;
        const int32_t at = luf_fetch_add(&W.counters[1], 1);
        if (at < W.lt_cap) { W.lt_row[at] = drow[i]; W.lt_step[at] = dsc[j]; W.lt_val[at] = v; }
    }
    PAR_FOR(i, na) {
        const int32_t t = drow[i];
        int32_t n = 0;
        for (int32_t j = 0; j < na; ++j) if (dsc[j] >= dsr[i] && D[i * na + j] != 0.0) ++n;
        int32_t b = W.rbeg[t];
        if (n > W.rcap[t]) {
            b = luf_fetch_add(&W.counters[0], n);
            if (b + n > W.arena_cap) { luf_st(&W.scalars[0], 2); continue; }
            W.rbeg[t] = b; W.rcap[t] = n;
        }
        n = 0;
        for (int32_t j = 0; j < na; ++j) if (dsc[j] >= dsr[i] && D[i * na + j] != 0.0) { W.ecol[b + n] = dcol[j]; W.eval[b + n] = D[i * na + j]; ++n; }
        W.rlen[t] = n;
    }
    PAR_FOR(j, na) W.prank[dcol[j]] = -1;
    LUF_SINGLE { ++W.counters[2]; W.counters[4] += na; } PAR_END
    if (luf_ld(&W.scalars[0]) || luf_ld(&W.counters[1]) > W.lt_cap) return LUF_NO_ROOM;
    return LUF_OK;
}

// Part 2: the bump and the four views.  `W` may carry the bump's arrays (rbeg .. cprio, ecol / eval, counters, red, scalars) in
// LDS while everything else stays in global memory (relp_kernels_luf.hip: k_lu_factor).
LUF_FN void luf_bump(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O, const int32_t k_peel) {
    const int32_t m = M.m;
    constexpr double kThreshold = 0.1;
    LUF_LAP_BEGIN
    (void)basis;
    // ---- the bump: sparse rows in an arena -------------------------------------------------------------------------------------
    const int32_t nbr = luf_select(m, [&](int32_t i) { return O.row_step[i] < 0; }, W.brow);
    const int32_t nbc = luf_select(m, [&](int32_t c) { return O.col_step[c] < 0; }, W.bcol);
    if (nbr != nbc) { luf_fail(O, LUF_SINGULAR); return; }
    const int32_t nb = nbr;
    LUF_SINGLE { O.status[1] = nb; O.status[2] = k_peel; luf_st(&W.counters[0], 0); luf_st(&W.counters[1], 0); W.counters[2] = 0; } PAR_END
    if (nb > W.nb_cap) { luf_fail(O, LUF_BUMP_TOO_LARGE); return; }
    if (nb > 0) {
        PAR_FOR(t, nb) {
            W.lrow[W.brow[t]] = t; W.lcol[W.bcol[t]] = t; W.ract[t] = 1; W.cact[t] = 1; luf_st(&W.bcc[t], 0);
            W.bstep_row[t] = -1; W.bstep_col[t] = -1; W.prank[t] = -1;
        } PAR_END
        // room per row: what it holds + a quarter (+ 2); a row that outgrows it moves to the end of the arena with half as much again
        // (the arena of the LDS variant is 9,216 entries: tight rooms, cheap moves)
        PAR_FOR(t, nb) {
            int32_t n = 0;
            luf_row_entries(M, W, W.brow[t], [&](int32_t c, double v) { if (W.lcol[c] >= 0 && v != 0.0) ++n; });
            W.rlen[t] = n; W.rcap[t] = n + n / 4 + 2; W.lt_ptr[t + 1] = n + n / 4 + 2;
        } PAR_END
        const int32_t room = luf_offsets_from_counts(W.lt_ptr, nb);
        if (room > W.arena_cap) { luf_fail(O, LUF_NO_ROOM); return; }
        LUF_SINGLE { luf_st(&W.counters[0], room); } PAR_END
        PAR_FOR(t, nb) {
            const int32_t b = W.lt_ptr[t];
            W.rbeg[t] = b;
            int32_t n = 0;
            luf_row_entries(M, W, W.brow[t], [&](int32_t c, double v) {
                const int32_t u = W.lcol[c];
                if (u >= 0 && v != 0.0) { W.ecol[b + n] = u; W.eval[b + n] = v; ++n; luf_add(&W.bcc[u], 1); }
            });
        } PAR_END
    }
    LUF_LAP(2);
    int32_t done = 0;
    while (done < nb) {
        if (W.dense_cap > 0 && nb - done <= W.dense_cap) {
            const int32_t rc = luf_dense_finish(W, O, k_peel, nb, done);
            if (rc != LUF_OK) { luf_fail(O, rc); return; }
            LUF_LAP(7);
            break;
        }
        // (1) per column: the largest active entry (threshold test), and the marks of this round
        PAR_FOR(u, nb) { if (W.cact[u]) { luf_st64(&W.cmax[u], 0ull); luf_st64(&W.rowmark[u], ~0ull); luf_st64(&W.colbest[u], ~0ull); } }
        LUF_SINGLE { luf_st64(&W.red[0], ~0ull); luf_st(&W.scalars[0], 0); } PAR_END
        PAR_FOR(t, nb) {
            if (!W.ract[t]) continue;
            const int32_t b = W.rbeg[t], n = W.rlen[t];
            if (n == 0) luf_st(&W.scalars[0], 1);                                  // an active row without entries: singular
            for (int32_t e = b; e < b + n; ++e) luf_max64(&W.cmax[W.ecol[e]], luf_bits(W.eval[e]));
        } PAR_END
        if (luf_ld(&W.scalars[0])) { luf_fail(O, LUF_SINGULAR); return; }
        LUF_LAP(3);
        // (2) every row proposes its entry of lowest Markowitz count (r - 1)(c - 1) among those that pass the threshold
        //     (pivoting.rs:45-81 takes the minimum over all entries; ties: the lower column, then the lower row)
        PAR_FOR(t, nb) {
            if (!W.ract[t]) continue;
            const int32_t b = W.rbeg[t], n = W.rlen[t];
            unsigned long long best = ~0ull;
            for (int32_t e = b; e < b + n; ++e) {
                const int32_t u = W.ecol[e];
                if (luf_bits(W.eval[e]) < luf_bits(kThreshold * luf_from_bits(luf_ld64(&W.cmax[u])))) continue;
                const unsigned long long cost = (unsigned long long)(uint32_t)(n - 1) * (unsigned long long)(uint32_t)(luf_ld(&W.bcc[u]) - 1);
                const unsigned long long key = ((cost < 0xffffffffull ? cost : 0xfffffffeull) << 32) | (uint32_t)u;
                if (key < best) best = key;
            }
            W.cpiv[t] = best == ~0ull ? -1 : (int32_t)(best & 0xffffffffu);
            const unsigned long long prio = best == ~0ull ? ~0ull : ((best >> 32) << 32) | (uint32_t)t;
            W.cprio[t] = prio;
            if (prio != ~0ull) luf_min64(&W.red[0], prio);
        } PAR_END
        const unsigned long long lead = luf_ld64(&W.red[0]);
        if (lead == ~0ull) { luf_fail(O, LUF_SINGULAR); return; }                  // (every non-empty column has an entry that passes)
        LUF_LAP(4);
        // (3) independence: EVERY proposal takes part (measured on a mid-solve basis of 25FV47, bump 401: 36 rounds and 4,136
        //     entries of fill-in, against 181 rounds / 4,162 entries when only the proposals of the round's minimum count may
        //     stand -- the priorities already favour the low counts -- and 4,190 entries of the host's lu_factor);
        //     proposals mark every column they have an entry in, and the column they propose
        PAR_FOR(t, nb) {
            if (!W.ract[t]) continue;
            const unsigned long long prio = W.cprio[t];
            if (prio == ~0ull) continue;                                           // (cpiv is -1)
            const int32_t b = W.rbeg[t], n = W.rlen[t];
            for (int32_t e = b; e < b + n; ++e) luf_min64(&W.rowmark[W.ecol[e]], prio);
            luf_min64(&W.colbest[W.cpiv[t]], prio);
        } PAR_END
        //     a proposal stands when no better one has an entry in its pivot column and it has no entry in a better one's pivot column
        PAR_FOR(t, nb) {
            if (!W.ract[t] || W.cpiv[t] < 0) continue;
            const unsigned long long prio = W.cprio[t];
            const int32_t pc = W.cpiv[t], b = W.rbeg[t], n = W.rlen[t];
            bool ok = luf_ld64(&W.rowmark[pc]) == prio;
            for (int32_t e = b; e < b + n && ok; ++e) { const int32_t u = W.ecol[e]; if (u != pc && luf_ld64(&W.colbest[u]) < prio) ok = false; }
            if (!ok) W.cprio[t] = ~0ull;                                           // (cpiv stays: the compaction below reads cprio)
        } PAR_END
        const int32_t n_acc = luf_select(nb, [&](int32_t t) { return W.ract[t] && W.cpiv[t] >= 0 && W.cprio[t] != ~0ull; }, W.acc);
        // (the best proposal of the round always stands)
        PAR_FOR(a, n_acc) {
            const int32_t t = W.acc[a], pc = W.cpiv[t], b = W.rbeg[t], n = W.rlen[t];
            double pv = 0.0;
            for (int32_t e = b; e < b + n; ++e) if (W.ecol[e] == pc) pv = W.eval[e];
            W.pval[a] = pv; W.prank[pc] = a;
            W.ract[t] = 0; W.cact[pc] = 0; W.bstep_row[t] = done + a; W.bstep_col[pc] = done + a;
            const int32_t i = W.brow[t], c = W.bcol[pc], k = k_peel + done + a;
            O.row_step[i] = k; O.col_step[c] = k; O.rowperm[k] = i; O.colperm[k] = c; O.diag[k] = pv;
        } PAR_END
        LUF_LAP(5);
        // (4) elimination: every other active row is rewritten by ONE wave (device) / one thread (host model), the pivots that reach
        //     it in ascending rank, so the arithmetic does not depend on the execution
        auto eliminate_serial = [&](int32_t t) {
            for (;;) {
                int32_t b = W.rbeg[t], n = W.rlen[t], hit = -1, rank = 0x7fffffff;
                for (int32_t e = b; e < b + n; ++e) { const int32_t r = W.prank[W.ecol[e]]; if (r >= 0 && r < rank) { rank = r; hit = e; } }
                if (hit < 0) break;
                const int32_t pc = W.ecol[hit];
                const double f = W.eval[hit] / W.pval[rank];
                const int32_t at = luf_fetch_add(&W.counters[1], 1);
                if (at < W.lt_cap) { W.lt_row[at] = t; W.lt_step[at] = done + rank; W.lt_val[at] = f; }
                W.ecol[hit] = W.ecol[b + n - 1]; W.eval[hit] = W.eval[b + n - 1]; --n;       // the entry of the pivot column leaves the row
                const int32_t pr = W.acc[rank], pb = W.rbeg[pr], pn = W.rlen[pr];
                for (int32_t q = pb; q < pb + pn; ++q) {
                    const int32_t w = W.ecol[q];
                    if (w == pc) continue;
                    const double pw = W.eval[q];
                    int32_t at_w = -1;
                    for (int32_t e = b; e < b + n; ++e) if (W.ecol[e] == w) { at_w = e; break; }
                    if (at_w >= 0) {
                        const double nv = W.eval[at_w] - f * pw;
                        if (nv == 0.0) {                                   // exact cancellation (decomposition/mod.rs:178)
                            W.ecol[at_w] = W.ecol[b + n - 1]; W.eval[at_w] = W.eval[b + n - 1]; --n;
                            luf_add(&W.bcc[w], -1);
                        } else {
                            W.eval[at_w] = nv;
                        }
                    } else {                                               // fill
                        if (n == W.rcap[t]) {
                            const int32_t want = n + n / 2 + 4, nbeg = luf_fetch_add(&W.counters[0], want);
                            if (nbeg + want > W.arena_cap) { luf_st(&W.scalars[0], 2); continue; }      // (reported behind the round)
                            for (int32_t e = 0; e < n; ++e) { W.ecol[nbeg + e] = W.ecol[b + e]; W.eval[nbeg + e] = W.eval[b + e]; }
                            b = nbeg; W.rbeg[t] = nbeg; W.rcap[t] = want;
                        }
                        W.ecol[b + n] = w; W.eval[b + n] = -f * pw; ++n;
                        luf_add(&W.bcc[w], 1);
                    }
                }
                W.rlen[t] = n;
            }
        };
#if defined(RELP_LUF_DEVICE)
        // A wave per row: the row sits in registers (one entry per lane, rows of <= 64 entries), a pivot row is read one entry per
        // lane and walked by broadcast: `ballot(col == w)` finds the entry to update or says fill-in.  One thread per row (the
        // serial form above) cost 2.5 M clocks per factorisation of a 25FV47 basis: its worst row does hits x |pivot row| x |row|
        // dependent LDS reads while 500 threads wait at the barrier.
        {
            const int lane = LUF_TID & 63, wave = LUF_TID >> 6, nw = LUF_NT >> 6;
            // (values of a lane the whole wave agrees on: v_readlane, not a trip through the LDS crossbar)
            auto rl = [](int32_t x, int l) { return __builtin_amdgcn_readlane(x, l); };
            auto rld = [](double x, int l) {
                union { double d; int32_t w[2]; } u; u.d = x;
                u.w[0] = __builtin_amdgcn_readlane(u.w[0], l); u.w[1] = __builtin_amdgcn_readlane(u.w[1], l);
                return u.d;
            };
            // the rows with an entry in one of this round's pivot columns, listed (a thread per row looks), then dealt to the waves
            // one by one: the rows that keep being hit sit together, a wave that owned a block of rows did most of the round alone
            int32_t* const hitlist = reinterpret_cast<int32_t*>(W.cprio);                      // (the priorities are spent)
            int32_t* const hitgrow = hitlist + W.nb_cap;
            PAR_FOR(tl, nb) {
                int32_t my_grow = 0; bool hit = false;                                       // entries the pivot rows can add
                if (W.ract[tl]) {
                    const int32_t my_b = W.rbeg[tl], my_n = W.rlen[tl];
                    for (int32_t e = my_b; e < my_b + my_n; ++e) { const int32_t r = W.prank[W.ecol[e]]; if (r >= 0) { hit = true; my_grow += W.rlen[W.acc[r]] - 1; } }
                }
                hitgrow[tl] = hit ? my_grow : -1;
            } PAR_END
            const int32_t nhit = luf_select(nb, [&](int32_t t) { return hitgrow[t] >= 0; }, hitlist);
            {
#if defined(LUF_WAVE_CLOCKS)
                unsigned long long* const wc = reinterpret_cast<unsigned long long*>(W.counters + 8) + 12;
                long long wt = (long long)__builtin_readcyclecounter();
#define LUF_WLAP(i) do { const long long now_ = (long long)__builtin_readcyclecounter(); if (wave == 0 && lane == 0) wc[i] += (unsigned long long)(now_ - wt); wt = now_; } while (0)
#define LUF_WCNT(i) do { if (wave == 0 && lane == 0) wc[i] += 1; } while (0)
#else
#define LUF_WLAP(i) do {} while (0)
#define LUF_WCNT(i) do {} while (0)
#endif
                for (int32_t h = wave; h < nhit; h += nw) {
                    LUF_WLAP(0);
                    const int32_t t = hitlist[h];
                    int32_t b = W.rbeg[t], n = W.rlen[t];
                    const int32_t grow = hitgrow[t];
                    if (n + grow > 64) { if (lane == 0) eliminate_serial(t); continue; }          // (what the row can grow to)
                    int32_t col = lane < n ? W.ecol[b + lane] : -1;
                    double val = lane < n ? W.eval[b + lane] : 0.0;
                    int32_t rk = col >= 0 ? W.prank[col] : -1;
                    unsigned long long hits = __ballot(rk >= 0);
                    LUF_WLAP(1); LUF_WCNT(5);
                    while (hits) {
                        int32_t r = 0x7fffffff; int hl = 0;                                     // the lowest rank among the hits
                        for (unsigned long long h = hits; h; h &= h - 1) { const int l = __ffsll(h) - 1; const int32_t x = rl(rk, l); if (x < r) { r = x; hl = l; } }
                        const double f = rld(val, hl) / W.pval[r];
                        const int32_t pc = rl(col, hl);
                        if (lane == 0) {
                            const int32_t at = luf_fetch_add(&W.counters[1], 1);
                            if (at < W.lt_cap) { W.lt_row[at] = t; W.lt_step[at] = done + r; W.lt_val[at] = f; }
                        }
                        {   // the entry of the pivot column leaves the row: the last entry takes its place
                            const int32_t lc = rl(col, n - 1), lr = rl(rk, n - 1); const double lv = rld(val, n - 1);
                            if (lane == hl) { col = lc; val = lv; rk = lr; }
                            if (lane == n - 1) { col = -1; val = 0.0; rk = -1; }
                            --n;
                        }
                        const int32_t pr = W.acc[r], pb = W.rbeg[pr], pn = W.rlen[pr];
                        for (int32_t q0 = 0; q0 < pn; q0 += 64) {
                            const int32_t myw = q0 + lane < pn ? W.ecol[pb + q0 + lane] : -1;
                            const double mypw = q0 + lane < pn ? W.eval[pb + q0 + lane] : 0.0;
                            const int32_t cnt = pn - q0 < 64 ? pn - q0 : 64;
                            LUF_WLAP(2); LUF_WCNT(6);
                            for (int32_t q = 0; q < cnt; ++q) {
                                LUF_WCNT(7);
                                const int32_t w = rl(myw, q);
                                if (w == pc) continue;
                                const double pw = rld(mypw, q);
                                const unsigned long long mt = __ballot(col == w);
                                if (mt) {
                                    const int ml = __ffsll(mt) - 1;
                                    const double nv = val - f * pw;
                                    if (rld(nv, ml) == 0.0) {                                  // exact cancellation (decomposition/mod.rs:178)
                                        const int32_t lc = rl(col, n - 1), lr = rl(rk, n - 1); const double lv = rld(val, n - 1);
                                        if (lane == ml) { col = lc; val = lv; rk = lr; }
                                        if (lane == n - 1) { col = -1; val = 0.0; rk = -1; }
                                        --n;
                                        if (lane == 0) luf_add(&W.bcc[w], -1);
                                    } else if (lane == ml) {
                                        val = nv;
                                    }
                                } else {                                                       // fill (never in a column that pivots this round)
                                    if (lane == n) { col = w; val = -f * pw; rk = -1; }
                                    ++n;
                                    if (lane == 0) luf_add(&W.bcc[w], 1);
                                }
                            }
                        }
                        hits = __ballot(rk >= 0);
                        LUF_WLAP(3);
                    }
                    if (n > W.rcap[t]) {
                        const int32_t want = n + n / 2 + 4;
                        int32_t nbeg = 0;
                        if (lane == 0) nbeg = luf_fetch_add(&W.counters[0], want);
                        nbeg = __builtin_amdgcn_readfirstlane(nbeg);
                        if (nbeg + want > W.arena_cap) { if (lane == 0) luf_st(&W.scalars[0], 2); continue; }     // (reported behind the round)
                        b = nbeg;
                        if (lane == 0) { W.rbeg[t] = nbeg; W.rcap[t] = want; }
                    }
                    if (lane < n) { W.ecol[b + lane] = col; W.eval[b + lane] = val; }
                    if (lane == 0) W.rlen[t] = n;
                    LUF_WLAP(4);
                }
            }
        } PAR_END
#else
#if defined(LUF_ROUND_TRACE)
        { int rows = 0, hits = 0, over = 0, work = 0, maxwork = 0;
          for (int32_t t = 0; t < nb; ++t) { if (!W.ract[t]) continue; int h = 0, grow = 0;
              for (int32_t e = W.rbeg[t]; e < W.rbeg[t] + W.rlen[t]; ++e) { const int32_t r = W.prank[W.ecol[e]]; if (r >= 0) { ++h; grow += W.rlen[W.acc[r]] - 1; } }
              if (h) { ++rows; hits += h; work += grow; if (grow > maxwork) maxwork = grow; if (W.rlen[t] + grow > 64) ++over; } }
          std::printf("   elimination: %d rows hit, %d hits, %d pivot entries walked (most on one row %d), %d rows beyond the 64-entry bound\n", rows, hits, work, maxwork, over); }
#endif
        PAR_FOR(t, nb) { if (W.ract[t]) eliminate_serial(t); } PAR_END
#endif
        LUF_LAP(6);
        // (5) the pivot rows leave: their columns lose an active entry; the round's column marks are taken back
        PAR_FOR(a, n_acc) {
            const int32_t t = W.acc[a], pc = W.cpiv[t], b = W.rbeg[t], n = W.rlen[t];
            for (int32_t e = b; e < b + n; ++e) if (W.ecol[e] != pc) luf_add(&W.bcc[W.ecol[e]], -1);
        } PAR_END
        PAR_FOR(a, n_acc) { W.prank[W.cpiv[W.acc[a]]] = -1; } PAR_END
        LUF_SINGLE { ++W.counters[2]; } PAR_END
        if (luf_ld(&W.scalars[0]) || luf_ld(&W.counters[1]) > W.lt_cap) { luf_fail(O, LUF_NO_ROOM); return; }
#if defined(LUF_ROUND_TRACE)
        { int32_t act = 0, mx = 0; long long tot = 0; for (int32_t t = 0; t < nb; ++t) if (W.ract[t]) { ++act; tot += W.rlen[t]; if (W.rlen[t] > mx) mx = W.rlen[t]; }
          std::printf("round %d: accepted %d, active rows left %d, entries %lld, longest row %d\n", W.counters[2], n_acc, act, tot, mx); }
#endif
        done += n_acc;
        LUF_LAP(7);
    }
    // ---- L and U in pivot coordinates: first as triplets (row step, column step, value), then the four views -----------------------
    // The multipliers of the bump already are triplets of L (local row, local step): into pivot coordinates where they lie.  Every
    // other entry comes from a basis column (outside bump x bump: never modified) or from a bump row as it was when it became the
    // pivot row, appended behind an atomic cursor -- the order of the triplets does not matter, a view places every entry by its
    // coordinates alone.  [r4: one thread per row walking luf_row_entries twice and insertion-sorting its rows in global memory
    // cost 1.9 M clocks of the 3.9 M of a factorisation of 25FV47.]
    const int32_t n_lt = luf_ld(&W.counters[1]);
    LUF_SINGLE { luf_st(&W.counters[5], 0); } PAR_END
    PAR_FOR(q, n_lt) { W.lt_row[q] = k_peel + W.bstep_row[W.lt_row[q]]; W.lt_step[q] += k_peel; }
    PAR_FOR(c, m) {
        const int32_t kc = O.col_step[c];
        const bool cb = W.lcol[c] >= 0;
        luf_col_entries(M, basis[c], [&](int32_t r, double v) {
            if (cb && W.lrow[r] >= 0) return;                              // bump x bump: from the arena
            const int32_t kk = O.row_step[r];
            if (kc > kk) {
                const int32_t at = luf_fetch_add(&W.counters[5], 1);
                if (at < W.lt_cap) { W.ut_row[at] = kk; W.ut_col[at] = kc; W.ut_val[at] = v; }
            } else if (kc < kk) {
                const int32_t at = luf_fetch_add(&W.counters[1], 1);
                if (at < W.lt_cap) { W.lt_row[at] = kk; W.lt_step[at] = kc; W.lt_val[at] = v / O.diag[kc]; }
            }
        });
    }
    PAR_FOR(t, nb) {
        const int32_t pc = W.cpiv[t], kk = k_peel + W.bstep_row[t];
        for (int32_t e = W.rbeg[t]; e < W.rbeg[t] + W.rlen[t]; ++e) {
            if (W.ecol[e] == pc) continue;
            const int32_t at = luf_fetch_add(&W.counters[5], 1);
            if (at < W.lt_cap) { W.ut_row[at] = kk; W.ut_col[at] = k_peel + W.bstep_col[W.ecol[e]]; W.ut_val[at] = W.eval[e]; }
        }
    } PAR_END
    const int32_t n_l = luf_ld(&W.counters[1]), n_u = luf_ld(&W.counters[5]);
    LUF_SINGLE {
        if (n_l > W.lt_cap || n_u > W.lt_cap || n_l > O.cap || n_u > O.cap || n_l > W.vtmp_cap || n_u > W.vtmp_cap) O.status[0] = LUF_NO_ROOM;
        O.status[3] = n_l; O.status[4] = n_u;
    } PAR_END
    if (n_l > W.lt_cap || n_u > W.lt_cap || n_l > O.cap || n_u > O.cap || n_l > W.vtmp_cap || n_u > W.vtmp_cap) return;
    LUF_LAP(8);
    // A view: the triplets bucketed by `bk`, every bucket ascending in `od` -- count, offsets, the `od` of every bucket side by side
    // (any order), then each entry goes to offset + (the number of entries of its bucket with a smaller `od`): coordinates are
    // unique, so the result does not depend on the execution.  (From here on the arena is free: `vw` / `vtmp` may lie in its LDS.)
    auto view = [&](int32_t n, const int32_t* bk, const int32_t* od, const double* val, const LufTriangle& R) {
        int32_t* cnt = W.vw;
        int32_t* tmp = W.vtmp;
        PAR_FOR(k, m + 1) luf_st(&cnt[k], 0); PAR_END
        PAR_FOR(e, n) luf_add(&cnt[bk[e] + 1], 1); PAR_END
        PAR_FOR(k, m + 1) { R.ptr[k] = luf_ld(&cnt[k]); } PAR_END
        (void)luf_offsets_from_counts(R.ptr, m);
        PAR_FOR(k, m + 1) luf_st(&cnt[k], 0); PAR_END
        PAR_FOR(e, n) { const int32_t b = bk[e]; tmp[R.ptr[b] + luf_fetch_add(&cnt[b], 1)] = od[e]; } PAR_END
        PAR_FOR(e, n) {
            const int32_t b = bk[e], mine = od[e], lo = R.ptr[b], hi = R.ptr[b + 1];
            int32_t rank = 0;
            for (int32_t a = lo; a < hi; ++a) rank += tmp[a] < mine ? 1 : 0;
            R.idx[lo + rank] = mine; R.val[lo + rank] = val[e];
        } PAR_END
    };
    view(n_l, W.lt_row, W.lt_step, W.lt_val, O.Lf);
    view(n_u, W.ut_row, W.ut_col, W.ut_val, O.Uf);
    LUF_LAP(9);
    view(n_l, W.lt_step, W.lt_row, W.lt_val, O.Lb);
    view(n_u, W.ut_col, W.ut_row, W.ut_val, O.Ub);
    LUF_LAP(10);
}

LUF_FN void luf_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O) {
    const int32_t k_peel = luf_peel(M, basis, W, O);
    if (k_peel < 0) return;
    luf_bump(M, basis, W, O, k_peel);
}

}  // namespace relp

#if !defined(RELP_LUF_DEVICE)
namespace relp {
// relp_kernels_luf.hip: the same algorithm as one workgroup of luf_threads() threads on stream s
void launch_lu_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O, hipStream_t s, bool lds = true);
int32_t luf_threads();
}  // namespace relp
#endif
