// relp_device_common.h -- device helpers shared by the three kernel files (constants, wavefront and
// workgroup reductions, the PRICE key, the ratio-test body).  Included by relp_kernels_*.hip only.
#pragma once
#include <algorithm>
#include "relp_kernels.h"

#include <math.h>

namespace relp {

static constexpr int kThreads = 256;     // 4 wavefronts
static constexpr int kVecPerBlock = 8;   // vectors (columns of A / rows of B^-1) per workgroup
static constexpr int kSingleBlock = 1024;

// ------------------------------------------------------------------------------------------------
// Wavefront (64 lanes) and workgroup reductions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Dot products of kVecPerBlock contiguous vectors (stride ld) with one shared vector x.
// Thread t streams 16-byte pairs k = 2t, 2t + 512, ...; all 8 loads of one step are independent.
// Vectors beyond `v_hi` are clamped (duplicate loads, results discarded) so there is no branch in
// the stream.  result[v] is valid for threads < kVecPerBlock after the call.
__device__ __forceinline__ void block_multi_dot(const double* __restrict__ M, int64_t ld, int len,
                                                int v0, int v_hi, const double* __restrict__ x,
                                                double* s_partial /* [4][kVecPerBlock] */,
                                                double& result) {
    const int t = threadIdx.x;
    double acc[kVecPerBlock];
    const double* base[kVecPerBlock];
#pragma unroll
    for (int v = 0; v < kVecPerBlock; ++v) {
        acc[v] = 0.0;
        int vi = v0 + v;
        if (vi >= v_hi) vi = v_hi - 1;
        base[v] = M + (int64_t)vi * ld;
    }
    const int len2 = len & ~1;
    for (int k = 2 * t; k < len2; k += 2 * kThreads) {
        const double2 xv = *reinterpret_cast<const double2*>(x + k);
#pragma unroll
        for (int v = 0; v < kVecPerBlock; ++v) {
            const double2 a = *reinterpret_cast<const double2*>(base[v] + k);
            acc[v] = fma(a.x, xv.x, acc[v]);
            acc[v] = fma(a.y, xv.y, acc[v]);
        }
    }
    if ((len & 1) && t == 0) {
        const double xl = x[len - 1];
#pragma unroll
        for (int v = 0; v < kVecPerBlock; ++v) acc[v] = fma(base[v][len - 1], xl, acc[v]);
    }
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int v = 0; v < kVecPerBlock; ++v) {
        const double w = wave_sum(acc[v]);
        if (lane == 0) s_partial[wave * kVecPerBlock + v] = w;
    }
    __syncthreads();
    if (t < kVecPerBlock) {
        result = (s_partial[0 * kVecPerBlock + t] + s_partial[1 * kVecPerBlock + t]) +
                 (s_partial[2 * kVecPerBlock + t] + s_partial[3 * kVecPerBlock + t]);
    }
}

// Selection key of a candidate column (smaller wins, ties by smaller j):
//   SteepestDescent: d_j (pivot_rule.rs:118); FirstProfitable[WithMemory]: position in the search order (:88)
__device__ __forceinline__ double select_key(int rule, int n, const PivotRecord* rec, int j, double d_j) {
    if (rule == 2) return d_j;
    const int last = (rule == 1 && rec) ? rec->last_selected : -1;
    if (last >= 0) return (double)(j >= last ? j - last : j - last + n);
    return (double)j;
}

// ------------------------------------------------------------------------------------------------
// PRICE
// Workgroup-level (key, j) minimum of one candidate per thread -> partial slot `slot`.
__device__ __forceinline__ void block_partial_min(double key, int kj, SelectPartials sp, int slot) {
    __shared__ double s_k[kThreads / 64];
    __shared__ int s_j[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_down(key, off, 64);
        const int oj = __shfl_down(kj, off, 64);
        if (ok < key || (ok == key && oj < kj)) { key = ok; kj = oj; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_k[wave] = key; s_j[wave] = kj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w)
            if (s_k[w] < key || (s_k[w] == key && s_j[w] < kj)) { key = s_k[w]; kj = s_j[w]; }
        sp.k1[slot] = key;
        sp.j[slot] = kj;
    }
}

// PRICE of 256 virtual columns (artificial, slack, bound slack; no matrix data: d_j = cost + (+-)(-pi)_row) and their
// partial argmin; `block` = index of the workgroup among the virtual-column workgroups.
__device__ __forceinline__ void price_virtual_body(const ColumnTable& ct, const double* __restrict__ minus_pi,
                                                   double* __restrict__ d, int cost_mode, const SelectPartials& sp,
                                                   const PivotRecord* rec, int block) {
    const int t = block * kThreads + threadIdx.x;
    int j = -1;
    double val = 0.0;
    if (t < ct.nr_artificial) {
        j = t;
        val = (cost_mode == 1 ? 1.0 : 0.0) + minus_pi[ct.column_to_row[t]];   // Cost::One + (-pi)_row
    } else {
        const int v = t - ct.nr_artificial;
        if (v < ct.nr_virtual) {
            // a slack whose row was removed as redundant (RemoveRows) is an empty column: vrow0 = -1
            const int r0 = ct.vrow0[v];
            double s = r0 >= 0 ? (double)ct.vsign[v] * minus_pi[r0] : 0.0;
            const int r1 = ct.vrow1[v];
            if (r1 >= 0) s += minus_pi[r1];
            j = ct.nr_artificial + ct.nr_normal + v;              // slack cost is None (zero)
            val = s;
        }
    }
    if (j >= 0) d[j] = val;
    if (!sp.k1) return;
    __shared__ double s_k[kThreads / 64];
    __shared__ int s_j[kThreads / 64];
    double key = INFINITY;
    int kj = 0x7fffffff;
    if (j >= 0 && !sp.in_basis[j] && val < -sp.tol_cost) { key = select_key(sp.rule, sp.n, rec, j, val); kj = j; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_down(key, off, 64);
        const int oj = __shfl_down(kj, off, 64);
        if (ok < key || (ok == key && oj < kj)) { key = ok; kj = oj; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_k[wave] = key; s_j[wave] = kj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w)
            if (s_k[w] < key || (s_k[w] == key && s_j[w] < kj)) { key = s_k[w]; kj = s_j[w]; }
        sp.k1[sp.offset + block] = key;
        sp.j[sp.offset + block] = kj;
    }
}

static constexpr int kMaxEta = 128;

// ------------------------------------------------------------------------------------------------
// RATIO TEST (single workgroup; two passes: strict minimum, then Bland tie-break on the leaving
// column among rows within the tie band -- identical to tableau/mod.rs:221-247 for zero tolerances)
// ------------------------------------------------------------------------------------------------
// Tie-break key of a row inside the tie band (smaller wins).  ratio_rule 0 = the reference: the leaving column
// (tableau/mod.rs:229-239).  ratio_rule 1 (an f64 safeguard, relp_engine.h: RELP_RATIO_LARGEST_PIVOT): the pivot element's
// size first -- at float precision, larger = smaller key -- then the leaving column.  Order-independent either way.
typedef unsigned long long tie_key_t;
static constexpr tie_key_t kNoTieKey = ~0ull;
__device__ __forceinline__ tie_key_t tie_key(double a, int leave, int ratio_rule) {
    const unsigned size = ratio_rule ? 0x7fffffffu - __float_as_uint((float)a) : 0u;      // (a > tol.pivot > 0)
    return ((tie_key_t)size << 32) | (unsigned)leave;
}
__device__ __forceinline__ int tie_key_leaving(tie_key_t k) { return (int)(unsigned)(k & 0xffffffffull); }

// Workgroup minimum of (key, row) over BS threads, result in every thread; s_k / s_r: BS / 64 words of LDS each.  `wide`
// (uniform) = the keys use their upper half; otherwise the reduction runs on the 32-bit leaving columns as it always did.
template <int BS>
__device__ __forceinline__ void tie_reduce(tie_key_t& key, int& row, tie_key_t* s_k, int* s_r, bool wide) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wide) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const tie_key_t ok = __shfl_down(key, off, 64);
            const int orow = __shfl_down(row, off, 64);
            if (ok < key) { key = ok; row = orow; }
        }
    } else {
        int lv = (int)(unsigned)key;                   // (kNoTieKey -> -1: mapped to INT_MAX below)
        if (key == kNoTieKey) lv = 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int ol = __shfl_down(lv, off, 64);
            const int orow = __shfl_down(row, off, 64);
            if (ol < lv) { lv = ol; row = orow; }
        }
        key = (unsigned)lv;
    }
    if (lane == 0) { s_k[wave] = key; s_r[wave] = row; }
    __syncthreads();
    // every thread finishes the reduction itself (BS / 64 LDS words): no second barrier
    key = s_k[0]; row = s_r[0];
#pragma unroll
    for (int w = 1; w < BS / 64; ++w)
        if (s_k[w] < key) { key = s_k[w]; row = s_r[w]; }
}

// Last step of the ratio test: every thread brings its best (tie key, row) among the rows inside the
// tie band; the workgroup's minimum wins (Bland on the leaving column, tableau/mod.rs:229-239), the record is
// written and the block bookkeeping of the deferred update (row r of W saved, slot of W chosen) is done.
template <int BS>
__device__ __forceinline__ void ratio_commit(tie_key_t best_key, int best_row, const double* alpha, const double* b,
                                             const DeferredUpdate& du, int p, PivotRecord* rec, bool wide = false, int guard_m = 0,
                                             double guard_rel = 0.0) {
    __shared__ tie_key_t s_cl[BS / 64];
    __shared__ int s_cr[BS / 64];
    // (everything the epilogue reads from memory -- alpha_r, b_r, row r of W, the slot of row r -- goes out in ONE round trip)
    tie_reduce<BS>(best_key, best_row, s_cl, s_cr, wide);
    const int best_leave = tie_key_leaving(best_key);
    const int r = best_row;
    if (guard_m > 0) {
        // Tolerances::pivot_guard: the chosen element against the largest |entry| of the column (one pass of the workgroup
        // over alpha; nothing has been written yet)
        __shared__ double s_gm[BS / 64];
        double mx = 0.0;
        for (int i = threadIdx.x; i < guard_m; i += BS) mx = fmax(mx, fabs(alpha[i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_gm[threadIdx.x >> 6] = mx;
        __syncthreads();
        double amax = s_gm[0];
#pragma unroll
        for (int w = 1; w < BS / 64; ++w) amax = fmax(amax, s_gm[w]);
        if (alpha[r] < guard_rel * amax) {
            if (threadIdx.x == 0) rec->outcome = DEV_NO_ROW;
            return;
        }
    }
    const bool deferred = du.kmax > 0;
    double a_r = 0.0, b_r = 0.0;
    int jt = 0;
    if (threadIdx.x == 0) {
        a_r = alpha[r];
        b_r = b[r];
        if (deferred) jt = du.pos_of_row[r];
    }
    // deferred update bookkeeping (k_eta_prepare): save row r of W, choose the column that receives u
    if (deferred)
        for (int j = threadIdx.x; j < p; j += BS) du.wr[j] = du.W[(int64_t)j * du.ld + r];
    if (threadIdx.x == 0) {
        rec->r = r;
        rec->leaving = best_leave;
        rec->alpha_r = a_r;
        rec->b_r = b_r;
        if (deferred) {
            rec->n_eta_old = p;
            if (jt < 0) { jt = p; du.S[p] = r; du.pos_of_row[r] = p; rec->n_eta = p + 1; }
            rec->eta_target = jt;
        }
    }
}

// Body of the ratio test for a workgroup of BS threads that keeps up to ITEMS rows per thread in
// registers.  `p` = rec->n_eta read by the caller together with the outcome.  Ends with the block
// bookkeeping of the deferred update.
template <int BS, int ITEMS>
__device__ __forceinline__ void ratio_body(const double* alpha, const double* b, const int32_t* basis_indices, int m,
                                           const Tolerances& tol, const DeferredUpdate& du, int p, PivotRecord* rec) {
    __shared__ double s_min[BS / 64];
    __shared__ double s_bcast;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // Each thread keeps its rows' ratios and leaving columns in registers (all loads issued at once, one
    // memory round trip); both passes then run out of registers.  m > 16 * 1024 falls back to re-reading.
    constexpr int kItems = ITEMS;
    const bool cached = m <= kItems * BS;
    double ratio_r[kItems], alpha_r[kItems];
    int leave_r[kItems];
    double mn = INFINITY;
    if (cached) {
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int i = threadIdx.x + k * BS;
            const bool in = i < m;
            const double a = in ? alpha[i] : 0.0;
            double bi = in ? b[i] : 0.0;
            leave_r[k] = in ? basis_indices[i] : 0x7fffffff;
            alpha_r[k] = a;
            if (bi <= tol.zero) bi = 0.0;   // also clamps a b_i that rounding pushed below 0: no negative step
            ratio_r[k] = (in && a > tol.pivot) ? bi / a : INFINITY;
            mn = fmin(mn, ratio_r[k]);
        }
    } else {
#pragma unroll 4
        for (int i = threadIdx.x; i < m; i += BS) {
            const double a = alpha[i];
            double bi = b[i];
            if (bi <= tol.zero) bi = 0.0;   // also clamps a b_i that rounding pushed below 0: no negative step
            const double ratio = (a > tol.pivot) ? bi / a : INFINITY;
            mn = fmin(mn, ratio);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    if (lane == 0) s_min[wave] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double g = s_min[0];
        for (int w = 1; w < BS / 64; ++w) g = fmin(g, s_min[w]);
        s_bcast = g;
    }
    __syncthreads();
    const double gmin = s_bcast;
    if (gmin == INFINITY) {
        if (threadIdx.x == 0) rec->outcome = DEV_NO_ROW;
        return;
    }
    const double bound = gmin + tol.tie * fmax(1.0, fabs(gmin));
    tie_key_t best_key = kNoTieKey;
    int best_row = -1;
    if (cached) {
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            // ratio_r is +inf for rows that do not take part, so `<= bound` excludes them
            if (ratio_r[k] <= bound) {
                const tie_key_t key = tie_key(alpha_r[k], leave_r[k], tol.ratio_rule);
                if (key < best_key) { best_key = key; best_row = threadIdx.x + k * BS; }
            }
        }
    } else {
#pragma unroll 4
        for (int i = threadIdx.x; i < m; i += BS) {
            const double a = alpha[i];
            double bi = b[i];
            const int lv = basis_indices[i];
            if (bi <= tol.zero) bi = 0.0;   // also clamps a b_i that rounding pushed below 0: no negative step
            if (a > tol.pivot && bi / a <= bound) {
                const tie_key_t key = tie_key(a, lv, tol.ratio_rule);
                if (key < best_key) { best_key = key; best_row = i; }
            }
        }
    }
    ratio_commit<BS>(best_key, best_row, alpha, b, du, p, rec, tol.ratio_rule != 0, tol.pivot_guard ? m : 0, tol.guard_rel);
}

// Ratio test of a workgroup of BS threads from the minimum ratio of every block of `rpb` rows (`rmin`, nblk
// entries, written by the kernel that formed alpha): the global minimum is the minimum of the block minima, and
// a row inside the tie band lives in a block whose own minimum is inside the band, so only those blocks' rows
// (usually one or two blocks) are read again.  Same result as ratio_body.  `p` = rec->n_eta read by the caller.
// b_i / alpha_i as the ratio test's first pass forms it (ratio_body), +inf when the row does not qualify
__device__ __forceinline__ double row_ratio(double a, double bi, const Tolerances& tol) {
    if (bi <= tol.zero) bi = 0.0;   // also clamps a b_i that rounding pushed below 0: no negative step
    return a > tol.pivot ? bi / a : INFINITY;
}

template <int BS>
__device__ __forceinline__ void ratio_blocks_body(const double* __restrict__ alpha, const double* __restrict__ b,
                                                  const int32_t* __restrict__ basis_indices, int m, const Tolerances& tol,
                                                  const DeferredUpdate& du, const double* __restrict__ rmin, int nblk, int p,
                                                  PivotRecord* rec, double first = INFINITY, bool have_first = false,
                                                  int rpb = kThreads) {
    __shared__ double s_min[BS / 64];
    __shared__ double s_bcast;
    constexpr int kListMax = 64;
    __shared__ int s_list[kListMax];
    __shared__ int s_cnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // `first` = rmin[threadIdx.x] when the caller loaded it together with the record (have_first)
    double mn = have_first ? first : INFINITY;
    for (int t = threadIdx.x + (have_first ? BS : 0); t < nblk; t += BS) mn = fmin(mn, rmin[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    if (lane == 0) s_min[wave] = mn;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        double g = s_min[0];
        for (int w = 1; w < BS / 64; ++w) g = fmin(g, s_min[w]);
        s_bcast = g;
    }
    __syncthreads();
    const double gmin = s_bcast;
    if (gmin == INFINITY) {
        if (threadIdx.x == 0) rec->outcome = DEV_NO_ROW;
        return;
    }
    const double bound = gmin + tol.tie * fmax(1.0, fabs(gmin));
    for (int t = threadIdx.x; t < nblk; t += BS) {
        if (!(rmin[t] <= bound)) continue;
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < kListMax) s_list[pos] = t;
    }
    __syncthreads();
    const int listed = s_cnt;
    const bool use_list = listed <= kListMax;
    const int total = (use_list ? listed : nblk) * rpb;
    tie_key_t best_key = kNoTieKey;
    int best_row = -1;
    for (int idx = threadIdx.x; idx < total; idx += BS) {
        const int t = use_list ? s_list[idx / rpb] : idx / rpb;
        const int i = t * rpb + idx % rpb;
        if (i >= m) continue;
        const double a = alpha[i];
        double bi = b[i];
        if (bi <= tol.zero) bi = 0.0;   // also clamps a b_i that rounding pushed below 0: no negative step
        if (a > tol.pivot && bi / a <= bound) {
            const tie_key_t key = tie_key(a, basis_indices[i], tol.ratio_rule);
            if (key < best_key) { best_key = key; best_row = i; }
        }
    }
    ratio_commit<BS>(best_key, best_row, alpha, b, du, p, rec, tol.ratio_rule != 0, tol.pivot_guard ? m : 0, tol.guard_rel);
}

// The same choice without the bookkeeping: every thread of the workgroup returns with (row, leaving column), row = -1 when
// no row qualifies.  For launches in which every workgroup needs the pivot row (k_tab_ratio_update_all).
template <int BS>
__device__ __forceinline__ void ratio_blocks_pick(const double* alpha, const double* b, const int32_t* basis_indices, int m,
                                                  const Tolerances& tol, const double* rmin, int nblk, int* row_out, int* leave_out,
                                                  double first = INFINITY, bool have_first = false, int rpb = kThreads) {
    __shared__ double s_min[BS / 64];
    __shared__ double s_bcast;
    constexpr int kListMax = 64;
    __shared__ int s_list[kListMax];
    __shared__ int s_cnt;
    __shared__ tie_key_t s_cl[BS / 64];
    __shared__ int s_cr[BS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // `first` = rmin[threadIdx.x] when the caller loaded it together with the record (have_first)
    double mn = have_first ? first : INFINITY;
    for (int t = threadIdx.x + (have_first ? BS : 0); t < nblk; t += BS) mn = fmin(mn, rmin[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    if (lane == 0) s_min[wave] = mn;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        double g = s_min[0];
        for (int w = 1; w < BS / 64; ++w) g = fmin(g, s_min[w]);
        s_bcast = g;
    }
    __syncthreads();
    const double gmin = s_bcast;
    if (gmin == INFINITY) { *row_out = -1; *leave_out = 0x7fffffff; return; }
    const double bound = gmin + tol.tie * fmax(1.0, fabs(gmin));
    for (int t = threadIdx.x; t < nblk; t += BS) {
        if (!(rmin[t] <= bound)) continue;
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < kListMax) s_list[pos] = t;
    }
    __syncthreads();
    const int listed = s_cnt;
    const bool use_list = listed <= kListMax;
    const int total = (use_list ? listed : nblk) * rpb;
    tie_key_t best_key = kNoTieKey;
    int best_row = -1;
    for (int idx = threadIdx.x; idx < total; idx += BS) {
        const int t = use_list ? s_list[idx / rpb] : idx / rpb;
        const int i = t * rpb + idx % rpb;
        if (i >= m) continue;
        const double a = alpha[i];
        double bi = b[i];
        int lv = basis_indices[i];                     // with alpha and b: one round trip, not two
        asm volatile("" : "+v"(lv));
        if (bi <= tol.zero) bi = 0.0;
        if (a > tol.pivot && bi / a <= bound) {
            const tie_key_t key = tie_key(a, lv, tol.ratio_rule);
            if (key < best_key) { best_key = key; best_row = i; }
        }
    }
    tie_reduce<BS>(best_key, best_row, s_cl, s_cr, tol.ratio_rule != 0);
    *row_out = best_row; *leave_out = tie_key_leaving(best_key);
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// Blocks of 256 threads for a grid-stride loop over `total` elements.  (A HIP launch takes at most 2^32 - 1 threads: a tableau
// of 64,000 x 256,000 has four times as many elements, and a launch beyond the limit fails without running.)
static inline int element_blocks(int64_t total) { return (int)std::min<int64_t>((total + 255) / 256, int64_t(1) << 22); }

}  // namespace relp
