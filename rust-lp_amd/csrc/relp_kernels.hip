// relp_kernels.hip -- hand-written gfx950 kernels of the explicit-inverse revised-simplex engine.
//
// Every hot kernel is an HBM stream (<= 0.25 flop/byte), so the design rules are: 16-byte
// coalesced loads per lane along the contiguous dimension (columns of A, rows of B^-1), several
// independent vectors in flight per thread, 64-wide wavefront shuffle reductions, no host sync:
// per-pivot scalars travel through the device-resident PivotRecord.
//
// Reference rows implemented (SURVEY.md section 8a / 8a'):
//   k_price_structural + k_price_virtual  a2+a3  tableau/mod.rs:102-108, carry/mod.rs:572-577
//   k_select_column                        a2     strategy/pivot_rule.rs:38-126
//   k_build_column + k_ftran               a4     carry/basis_inverse_rows.rs:144-173
//   k_ratio                                a5     tableau/mod.rs:221-247
//   k_compute_rho + k_update_vectors       a6     carry/mod.rs:283-333, 549-570
//   k_update_inverse                       a7     carry/basis_inverse_rows.rs:42-83,131-142
//   k_weighted_column_sums                 a10    carry/mod.rs:214-248
#include "relp_kernels.h"

#include <algorithm>
#include <vector>

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

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_price_structural(
    const double* __restrict__ A, int64_t ld_a, ColumnTable ct, const double* __restrict__ minus_pi,
    double* __restrict__ d, int p_lo, int p_hi, int cost_mode, SelectPartials sp, const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    __shared__ double s_partial[4 * kVecPerBlock];
    const int v0 = p_lo + blockIdx.x * kVecPerBlock;
    double dot = 0.0;
    block_multi_dot(A, ld_a, ct.nr_constraints, v0, p_hi, minus_pi, s_partial, dot);
    const int p = v0 + threadIdx.x;
    double key = INFINITY;
    int kj = 0x7fffffff;
    if (threadIdx.x < kVecPerBlock && p < p_hi) {
        double v = dot;
        const int br = ct.bound_row[p];
        if (br >= 0) v += minus_pi[br];                           // +1 entry in the bound row
        if (cost_mode == 2) v += ct.cost[p];                      // phase 1: Cost::Zero
        const int j = ct.nr_artificial + p;
        d[j] = v;
        if (sp.k1 && !sp.in_basis[j] && v < -sp.tol_cost) { key = select_key(sp.rule, sp.n, rec, j, v); kj = j; }
    }
    if (sp.k1 && threadIdx.x < 64) {                              // lanes 0..7 of wavefront 0 hold the candidates
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            const double ok = __shfl_down(key, off, 64);
            const int oj = __shfl_down(kj, off, 64);
            if (ok < key || (ok == key && oj < kj)) { key = ok; kj = oj; }
        }
        if (threadIdx.x == 0) { sp.k1[sp.offset + blockIdx.x] = key; sp.j[sp.offset + blockIdx.x] = kj; }
    }
}

__global__ void k_price_mask_unowned(ColumnTable ct, double* __restrict__ d, int p_lo, int p_hi,
                                     const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < ct.nr_normal && (p < p_lo || p >= p_hi)) d[ct.nr_artificial + p] = INFINITY;
}

__global__ __launch_bounds__(kThreads) void k_price_virtual(ColumnTable ct, const double* __restrict__ minus_pi,
                                                            double* __restrict__ d, int cost_mode, SelectPartials sp,
                                                            const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int j = -1;
    double val = 0.0;
    if (t < ct.nr_artificial) {
        j = t;
        val = (cost_mode == 1 ? 1.0 : 0.0) + minus_pi[ct.column_to_row[t]];   // Cost::One + (-pi)_row
    } else {
        const int v = t - ct.nr_artificial;
        if (v < ct.nr_virtual) {
            double s = (double)ct.vsign[v] * minus_pi[ct.vrow0[v]];
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
        sp.k1[sp.offset + blockIdx.x] = key;
        sp.j[sp.offset + blockIdx.x] = kj;
    }
}

// Entering column from the PRICE workgroups' partial results, then aq := that column in row space
// (k_select_column + k_build_column in one single-workgroup launch).
__global__ __launch_bounds__(kSingleBlock) void k_select_partials(SelectPartials sp, int count,
                                                                  const double* __restrict__ d,
                                                                  const double* __restrict__ A, int64_t ld_a,
                                                                  DeviceCSC csc, ColumnTable ct, int m,
                                                                  double* __restrict__ aq, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_k1[kSingleBlock / 64];
    __shared__ int s_j[kSingleBlock / 64];
    __shared__ int s_q;
    double k1 = INFINITY;
    int bj = 0x7fffffff;
    for (int t = threadIdx.x; t < count; t += kSingleBlock) {
        const double key = sp.k1[t];
        const int j = sp.j[t];
        if (key < k1 || (key == k1 && j < bj)) { k1 = key; bj = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_down(k1, off, 64);
        const int oj = __shfl_down(bj, off, 64);
        if (ok < k1 || (ok == k1 && oj < bj)) { k1 = ok; bj = oj; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_k1[wave] = k1; s_j[wave] = bj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kSingleBlock / 64; ++w)
            if (s_k1[w] < k1 || (s_k1[w] == k1 && s_j[w] < bj)) { k1 = s_k1[w]; bj = s_j[w]; }
        s_k1[0] = k1;
        s_j[0] = bj;
    }
    __syncthreads();
    k1 = s_k1[0];
    bj = s_j[0];
    __syncthreads();
    if (bj != 0x7fffffff && sp.rule == 2 && sp.tol_tie > 0.0) {
        // Dantzig ties: lowest index among the columns within the tie band of the minimum.  A column
        // inside the band lives in a workgroup whose own minimum is inside the band, so only those
        // workgroups' columns are re-read (8 structural columns, or 256 virtual ones).
        const double bound = k1 + sp.tol_tie * fmax(1.0, fabs(k1));
        int lowest = 0x7fffffff;
        // slots whose own minimum lies inside the band (normally one or two): listed, then scanned by
        // the whole workgroup; a long list falls back to one thread per slot
        constexpr int kListMax = 64;
        __shared__ int s_list[kListMax];
        __shared__ int s_cnt;
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < count; t += kSingleBlock) {
            if (!(sp.k1[t] <= bound)) continue;
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < kListMax) s_list[pos] = t;
        }
        __syncthreads();
        const int listed = s_cnt;
        auto scan_slot = [&](int t, int u0, int ustep) {
            if (t < sp.nb_struct) {
                const int p0 = sp.p_lo + t * sp.cols_per_slot;
                for (int u = u0; u < sp.cols_per_slot; u += ustep) {
                    const int p = p0 + u;
                    if (p >= ct.nr_normal) break;
                    const int j = ct.nr_artificial + p;
                    const double v = d[j];
                    if (!sp.in_basis[j] && v < -sp.tol_cost && v <= bound && j < lowest) lowest = j;
                }
            } else {
                const int v0 = (t - sp.nb_struct) * kThreads;
                for (int u = u0; u < kThreads; u += ustep) {
                    const int vt = v0 + u;
                    if (vt >= ct.nr_artificial + ct.nr_virtual) break;
                    const int j = vt < ct.nr_artificial ? vt : ct.nr_artificial + ct.nr_normal + (vt - ct.nr_artificial);
                    const double v = d[j];
                    if (!sp.in_basis[j] && v < -sp.tol_cost && v <= bound && j < lowest) lowest = j;
                }
            }
        };
        if (listed <= kListMax) {
            for (int i = 0; i < listed; ++i) scan_slot(s_list[i], threadIdx.x, kSingleBlock);
        } else {
            for (int t = threadIdx.x; t < count; t += kSingleBlock)
                if (sp.k1[t] <= bound) scan_slot(t, 0, 1);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lowest = min(lowest, __shfl_down(lowest, off, 64));
        if (lane == 0) s_j[wave] = lowest;
        __syncthreads();
        if (threadIdx.x == 0) {
            int low = 0x7fffffff;
            for (int w = 0; w < kSingleBlock / 64; ++w) low = min(low, s_j[w]);
            bj = low;
        }
    }
    if (threadIdx.x == 0) {
        if (bj == 0x7fffffff) {
            rec->outcome = DEV_NO_CANDIDATE;
            if (sp.rule == 1) rec->last_selected = -1;
            s_q = -1;
        } else {
            rec->q = bj;
            rec->d_q = d[bj];
            rec->key1 = k1;
            if (sp.rule == 1) rec->last_selected = bj;
            s_q = bj;
        }
    }
    __syncthreads();
    const int q = s_q;
    if (q < 0) return;
    // build the entering column (k_build_column)
    int kind = 0, p = 0, r0 = -1, r1 = -1;
    double sgn = 1.0;
    if (q < ct.nr_artificial) { kind = 1; r0 = ct.column_to_row[q]; }
    else {
        p = q - ct.nr_artificial;
        if (p < ct.nr_normal) { kind = 0; r0 = ct.bound_row[p]; }
        else { kind = 1; const int vv = p - ct.nr_normal; r0 = ct.vrow0[vv]; r1 = ct.vrow1[vv]; sgn = (double)ct.vsign[vv]; }
    }
    for (int i = threadIdx.x; i < m; i += kSingleBlock) {
        double v = 0.0;
        if (kind == 0) {
            if (i < ct.nr_constraints) v = A ? A[(int64_t)p * ld_a + i] : 0.0;
            else if (i == r0) v = 1.0;
        } else {
            if (i == r0) v = sgn;
            else if (i == r1) v = 1.0;
        }
        aq[i] = v;
    }
    if (kind == 0 && !A) {
        // sparse structural column: scatter its CSC entries over the zero fill
        __syncthreads();
        const int64_t s0 = csc.col_ptr[p], s1 = csc.col_ptr[p + 1];
        for (int64_t e = s0 + threadIdx.x; e < s1; e += kSingleBlock) aq[csc.row_idx[e]] = csc.values[e];
    }
}

// aq := tableau column rec->q built from the CSC matrix (k_build_column for the LU engine)
__global__ void k_build_column_csc(DeviceCSC csc, ColumnTable ct, int m, double* __restrict__ aq, const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    // single workgroup: zero fill, then scatter
    const int q = rec->q;
    int kind = 1, p = 0, r0 = -1, r1 = -1;
    double sgn = 1.0;
    if (q < ct.nr_artificial) r0 = ct.column_to_row[q];
    else {
        p = q - ct.nr_artificial;
        if (p < ct.nr_normal) { kind = 0; r0 = ct.bound_row[p]; }
        else { const int vv = p - ct.nr_normal; r0 = ct.vrow0[vv]; r1 = ct.vrow1[vv]; sgn = (double)ct.vsign[vv]; }
    }
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        double v = 0.0;
        if (kind == 0) { if (i == r0) v = 1.0; }
        else { if (i == r0) v = sgn; else if (i == r1) v = 1.0; }
        aq[i] = v;
    }
    if (kind == 0) {
        __syncthreads();
        const int64_t s0 = csc.col_ptr[p], s1 = csc.col_ptr[p + 1];
        for (int64_t e = s0 + threadIdx.x; e < s1; e += blockDim.x) aq[csc.row_idx[e]] = csc.values[e];
    }
}

// ------------------------------------------------------------------------------------------------
// Sparse LU engine: CSC PRICE and the level-scheduled triangular solves
// ------------------------------------------------------------------------------------------------
// d_j = c_j + sum_i (-pi)_i a_ij over the stored entries of column j (vector/dense.rs:81-92), thread per
// column; the workgroup's best (key, j) goes to the PRICE partials.
__global__ __launch_bounds__(kThreads) void k_price_csc(DeviceCSC csc, ColumnTable ct, const double* __restrict__ vec,
                                                        double* __restrict__ d, int p_lo, int p_hi, int cost_mode,
                                                        SelectPartials sp, const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    const int p = p_lo + blockIdx.x * kThreads + threadIdx.x;
    double key = INFINITY;
    int kj = 0x7fffffff;
    if (p < p_hi) {
        double v = 0.0;
        const int64_t s0 = csc.col_ptr[p], s1 = csc.col_ptr[p + 1];
        for (int64_t e = s0; e < s1; ++e) v = fma(vec[csc.row_idx[e]], csc.values[e], v);
        const int br = ct.bound_row[p];
        if (br >= 0) v += vec[br];
        if (cost_mode == 2) v += ct.cost[p];
        const int j = ct.nr_artificial + p;
        d[j] = v;
        if (sp.k1 && !sp.in_basis[j] && v < -sp.tol_cost) { key = select_key(sp.rule, sp.n, rec, j, v); kj = j; }
    }
    if (sp.k1) block_partial_min(key, kj, sp, sp.offset + blockIdx.x);
}

// ---- level-scheduled triangular solves ---------------------------------------------------------------
// One persistent workgroup of 256 threads.  The work vector lives in LDS; when the factor itself fits
// next to it (the usual case for Netlib-sized bases) its rows and entries are staged into LDS first, so
// that a level costs an LDS round trip and a barrier instead of a chain of dependent global loads.
// 8 to 64 lanes share one row (coalesced entry loads, DPP reduction), and each group fetches its
// first row of the NEXT level - row header and first entries do not depend on x - before it waits at
// the barrier of the current one.
static constexpr int kLuThreads = 256;          // 4 wavefronts: cheap barriers, 32 rows per pass
static constexpr int kLuLdsBytes = 156 * 1024;        // of the CU's 160 KB

__host__ __device__ inline int64_t lu_up16(int64_t b) { return (b + 15) / 16 * 16; }
// bytes needed to hold a schedule (m rows, nnz entries) in LDS
__host__ __device__ inline int64_t schedule_lds_bytes(int m, int64_t nnz, int n_levels) {
    return lu_up16((int64_t)sizeof(LuRow) * m) + lu_up16(8 * nnz) + lu_up16(4 * nnz) + lu_up16(4 * ((int64_t)n_levels + 1));
}

// Sum over the 8 lanes of a group, result valid in the group's lane 0.  DPP row shifts (lane i reads lane
// i + n inside its row of 16) instead of LDS-routed shuffles: the reduction sits on the critical path of
// every level.
template <int kCtrl>
__device__ __forceinline__ double dpp_row_shl(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Sum over the G lanes of a group (G = 8, 16, 32 or 64, uniform over the workgroup); valid in lane 0.
__device__ __forceinline__ double group_sum(double v, int G) {
    if (G >= 64) v += __shfl_down(v, 32, 64);
    if (G >= 32) v += __shfl_down(v, 16, 64);
    if (G >= 16) v += dpp_row_shl<0x108>(v);
    v += dpp_row_shl<0x104>(v);
    v += dpp_row_shl<0x102>(v);
    v += dpp_row_shl<0x101>(v);
    return v;
}
// Lanes per row for a level of `rows` rows: a level with few rows (the dense rows of the bump come one
// per level) gets a whole wavefront per row.
__device__ __forceinline__ int group_lanes(int rows) {
    return rows <= kLuThreads / 64 ? 64 : rows <= kLuThreads / 32 ? 32 : rows <= kLuThreads / 16 ? 16 : 8;
}

// kStage: copy the schedule into LDS at `base` and solve from there; otherwise solve from global memory.
template <bool kStage>
__device__ __forceinline__ void solve_schedule(const DeviceSchedule& s, int m, char* base, double* x) {
    const LuRow* rows = s.rows; const int32_t* idx = s.idx; const double* val = s.val; const int32_t* level_ptr = s.level_ptr;
    if (kStage) {
        LuRow* l_rows = reinterpret_cast<LuRow*>(base); base += lu_up16((int64_t)sizeof(LuRow) * m);
        double* l_val = reinterpret_cast<double*>(base); base += lu_up16(8 * (int64_t)s.nnz);
        int32_t* l_idx = reinterpret_cast<int32_t*>(base); base += lu_up16(4 * (int64_t)s.nnz);
        int32_t* l_lp = reinterpret_cast<int32_t*>(base);
        for (int e = threadIdx.x; e < s.nnz; e += blockDim.x) { l_val[e] = s.val[e]; l_idx[e] = s.idx[e]; }
        for (int k = threadIdx.x; k < m; k += blockDim.x) l_rows[k] = s.rows[k];
        for (int k = threadIdx.x; k <= s.n_levels; k += blockDim.x) l_lp[k] = s.level_ptr[k];
        __syncthreads();
        rows = l_rows; idx = l_idx; val = l_val; level_ptr = l_lp;
    }
    const int n_levels = s.n_levels;
    const int tid = threadIdx.x;
    // prefetched first row of the level about to be solved
    int t0 = level_ptr[0], t1 = level_ptr[1];
    int G = group_lanes(t1 - t0);
    LuRow pr{0, 0, 0, 0, 1.0};
    int pidx = 0; double pval = 0.0;
    bool have = t0 + tid / G < t1;
    if (have) {
        pr = rows[t0 + tid / G];
        if (pr.e0 + tid % G < pr.e1) { pidx = idx[pr.e0 + tid % G]; pval = val[pr.e0 + tid % G]; }
    }
    for (int lev = 0; lev < n_levels; ++lev) {
        const LuRow cr = pr; const int cidx = pidx; const double cval = pval; const bool chave = have;
        const int ct0 = t0, ct1 = t1, cG = G;
        const int g = tid / cG, lane = tid % cG, ngroups = kLuThreads / cG;
        if (lev + 1 < n_levels) {
            t0 = t1; t1 = level_ptr[lev + 2];
            G = group_lanes(t1 - t0);
            have = t0 + tid / G < t1; pval = 0.0; pidx = 0;
            if (have) {
                pr = rows[t0 + tid / G];
                if (pr.e0 + tid % G < pr.e1) { pidx = idx[pr.e0 + tid % G]; pval = val[pr.e0 + tid % G]; }
            }
        }
        if (chave) {
            double sum = (cr.e0 + lane < cr.e1) ? -cval * x[cidx] : 0.0;
            for (int e = cr.e0 + lane + cG; e < cr.e1; e += cG) sum = fma(-val[e], x[idx[e]], sum);
            sum = group_sum(sum, cG);
            if (lane == 0) x[cr.k] = (x[cr.k] + sum) * cr.diag;
        }
        for (int t = ct0 + g + ngroups; t < ct1; t += ngroups) {
            const LuRow r = rows[t];
            double sum = 0.0;
            for (int e = r.e0 + lane; e < r.e1; e += cG) sum = fma(-val[e], x[idx[e]], sum);
            sum = group_sum(sum, cG);
            if (lane == 0) x[r.k] = (x[r.k] + sum) * r.diag;
        }
        __syncthreads();
    }
}

// Dynamic LDS: [x : m doubles (kXLds)] [staged schedule].  The variant is chosen by the host from the
// sizes (plan_lu_lds).
// FTRAN (lower_upper/mod.rs:157-190 without the update loop: the updates live in W): P a -> L -> U -> Q
template <bool kXLds, bool kStage1, bool kStage2>
__global__ __launch_bounds__(kLuThreads) void k_lu_ftran(DeviceLU lu, const double* __restrict__ aq,
                                                           double* __restrict__ v, double* __restrict__ scratch,
                                                           const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    extern __shared__ __align__(16) char lds[];
    double* x = kXLds ? reinterpret_cast<double*>(lds) : scratch;
    char* base = lds + (kXLds ? lu_up16((int64_t)lu.m * 8) : 0);
    for (int k = threadIdx.x; k < lu.m; k += blockDim.x) x[k] = aq[lu.rowperm[k]];
    __syncthreads();
    solve_schedule<kStage1>(lu.Lf, lu.m, base, x);
    solve_schedule<kStage2>(lu.Uf, lu.m, base, x);
    for (int k = threadIdx.x; k < lu.m; k += blockDim.x) v[lu.colperm[k]] = x[k];
}

// BTRAN (lower_upper/mod.rs:204-222): z' B = c'  ->  U' t = Q' c, L' w = t, z = P' w
template <bool kXLds, bool kStage1, bool kStage2>
__global__ __launch_bounds__(kLuThreads) void k_lu_btran(DeviceLU lu, DeferredUpdate du, const double* __restrict__ rhs,
                                                           int row, double* __restrict__ rho, double* __restrict__ scratch,
                                                           const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    extern __shared__ __align__(16) char lds[];
    double* x = kXLds ? reinterpret_cast<double*>(lds) : scratch;
    char* base = lds + (kXLds ? lu_up16((int64_t)lu.m * 8) : 0);
    const int r = rhs ? 0 : (row >= 0 ? row : rec->r);   // rec may be null when rhs or row is given
    for (int k = threadIdx.x; k < lu.m; k += blockDim.x) {
        const int cp = lu.colperm[k];
        double c;
        if (rhs) c = rhs[cp];
        else {
            // c = e_r + sum_j W[r, j] e_S[j]   (the pivot row of (I + W S') B0inv)
            c = (cp == r) ? 1.0 : 0.0;
            if (du.kmax > 0) { const int jt = du.pos_of_row[cp]; if (jt >= 0) c += du.W[(int64_t)jt * du.ld + r]; }
        }
        x[k] = c;
    }
    __syncthreads();
    solve_schedule<kStage1>(lu.Ub, lu.m, base, x);
    solve_schedule<kStage2>(lu.Lb, lu.m, base, x);
    for (int k = threadIdx.x; k < lu.m; k += blockDim.x) rho[lu.rowperm[k]] = x[k];
}


// Single-workgroup selection over the reduced costs.  key = (k1, j) lexicographic minimum:
//   SteepestDescent: k1 = d_j (strict `<` => lowest j wins ties, pivot_rule.rs:118)
//   FirstProfitable[WithMemory]: k1 = position of j in the search order (pivot_rule.rs:88)
__global__ __launch_bounds__(kSingleBlock) void k_select_column(
    const double* __restrict__ d, const uint8_t* __restrict__ in_basis, int n, int rule, double tol_cost,
    double tol_tie, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_k1[kSingleBlock / 64];
    __shared__ int s_j[kSingleBlock / 64];
    const int last = (rule == 1) ? rec->last_selected : -1;
    double k1 = INFINITY;
    int bj = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += kSingleBlock) {
        if (in_basis[j]) continue;
        const double v = d[j];
        if (v < -tol_cost) {
            double key;
            if (rule == 2) key = v;
            else if (last >= 0) key = (double)(j >= last ? j - last : j - last + n);
            else key = (double)j;
            if (key < k1 || (key == k1 && j < bj)) { k1 = key; bj = j; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_down(k1, off, 64);
        const int oj = __shfl_down(bj, off, 64);
        if (ok < k1 || (ok == k1 && oj < bj)) { k1 = ok; bj = oj; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_k1[wave] = k1; s_j[wave] = bj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kSingleBlock / 64; ++w) {
            if (s_k1[w] < k1 || (s_k1[w] == k1 && s_j[w] < bj)) { k1 = s_k1[w]; bj = s_j[w]; }
        }
        s_j[0] = bj;
        s_k1[0] = k1;
    }
    __syncthreads();
    bj = s_j[0];
    k1 = s_k1[0];
    if (bj != 0x7fffffff && rule == 2 && tol_tie > 0.0) {
        // Dantzig ties: lowest index among the columns within the tie band of the minimum
        const double bound = k1 + tol_tie * fmax(1.0, fabs(k1));
        int lowest = 0x7fffffff;
        for (int j = threadIdx.x; j < n; j += kSingleBlock) {
            const double v = d[j];
            if (!in_basis[j] && v < -tol_cost && v <= bound && j < lowest) lowest = j;
        }
        __syncthreads();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lowest = min(lowest, __shfl_down(lowest, off, 64));
        if (lane == 0) s_j[wave] = lowest;
        __syncthreads();
        if (threadIdx.x == 0) {
            int low = 0x7fffffff;
            for (int w = 0; w < kSingleBlock / 64; ++w) low = min(low, s_j[w]);
            bj = low;                                  // the minimum itself is inside the band
        }
    }
    if (threadIdx.x == 0) {
        if (bj == 0x7fffffff) {
            rec->outcome = DEV_NO_CANDIDATE;
            if (rule == 1) rec->last_selected = -1;
        } else {
            rec->q = bj;
            rec->d_q = d[bj];
            rec->key1 = k1;
            if (rule == 1) rec->last_selected = bj;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FTRAN
// ------------------------------------------------------------------------------------------------
// aq := column q of the tableau's original matrix, dense over the m rows (padding stays zero).
__global__ void k_build_column(const double* __restrict__ A, int64_t ld_a, ColumnTable ct, int m,
                               double* __restrict__ aq, const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int q = rec->q;
    double v = 0.0;
    if (q < ct.nr_artificial) {
        v = (i == ct.column_to_row[q]) ? 1.0 : 0.0;
    } else {
        const int p = q - ct.nr_artificial;
        if (p < ct.nr_normal) {
            if (i < ct.nr_constraints) v = A[(int64_t)p * ld_a + i];
            else v = (i == ct.bound_row[p]) ? 1.0 : 0.0;
        } else {
            const int vv = p - ct.nr_normal;
            if (i == ct.vrow0[vv]) v = (double)ct.vsign[vv];
            else if (i == ct.vrow1[vv]) v = 1.0;
        }
    }
    aq[i] = v;
}

__global__ __launch_bounds__(kThreads) void k_ftran(const double* __restrict__ Binv, int64_t ld_b, int m,
                                                    int row_lo, int row_hi, const double* __restrict__ aq,
                                                    double* __restrict__ out, int out_offset,
                                                    const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_partial[4 * kVecPerBlock];
    const int v0 = row_lo + blockIdx.x * kVecPerBlock;
    double dot = 0.0;
    block_multi_dot(Binv, ld_b, m, v0, row_hi, aq, s_partial, dot);
    const int i = v0 + threadIdx.x;
    if (threadIdx.x < kVecPerBlock && i < row_hi) out[i - out_offset] = dot;
}

// ------------------------------------------------------------------------------------------------
// RATIO TEST (single workgroup; two passes: strict minimum, then Bland tie-break on the leaving
// column among rows within the tie band -- identical to tableau/mod.rs:221-247 for zero tolerances)
// ------------------------------------------------------------------------------------------------
// Body of the ratio test for a workgroup of BS threads that keeps up to ITEMS rows per thread in
// registers.  `p` = rec->n_eta read by the caller together with the outcome.  Ends with the block
// bookkeeping of the deferred update.
template <int BS, int ITEMS>
__device__ __forceinline__ void ratio_body(const double* alpha, const double* b, const int32_t* basis_indices, int m,
                                           const Tolerances& tol, const DeferredUpdate& du, int p, PivotRecord* rec) {
    __shared__ double s_min[BS / 64];
    __shared__ int s_leave[BS / 64];
    __shared__ int s_row[BS / 64];
    __shared__ double s_bcast;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // Each thread keeps its rows' ratios and leaving columns in registers (all loads issued at once, one
    // memory round trip); both passes then run out of registers.  m > 16 * 1024 falls back to re-reading.
    constexpr int kItems = ITEMS;
    const bool cached = m <= kItems * BS;
    double ratio_r[kItems];
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
            if (fabs(bi) <= tol.zero) bi = 0.0;
            ratio_r[k] = (in && a > tol.pivot) ? bi / a : INFINITY;
            mn = fmin(mn, ratio_r[k]);
        }
    } else {
#pragma unroll 4
        for (int i = threadIdx.x; i < m; i += BS) {
            const double a = alpha[i];
            double bi = b[i];
            if (fabs(bi) <= tol.zero) bi = 0.0;
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
    int best_leave = 0x7fffffff, best_row = -1;
    if (cached) {
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            // ratio_r is +inf for rows that do not take part, so `<= bound` excludes them
            if (ratio_r[k] <= bound && leave_r[k] < best_leave) { best_leave = leave_r[k]; best_row = threadIdx.x + k * BS; }
        }
    } else {
#pragma unroll 4
        for (int i = threadIdx.x; i < m; i += BS) {
            const double a = alpha[i];
            double bi = b[i];
            const int lv = basis_indices[i];
            if (fabs(bi) <= tol.zero) bi = 0.0;
            if (a > tol.pivot && bi / a <= bound && lv < best_leave) { best_leave = lv; best_row = i; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int ol = __shfl_down(best_leave, off, 64);
        const int orow = __shfl_down(best_row, off, 64);
        if (ol < best_leave) { best_leave = ol; best_row = orow; }
    }
    if (lane == 0) { s_leave[wave] = best_leave; s_row[wave] = best_row; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BS / 64; ++w)
            if (s_leave[w] < best_leave) { best_leave = s_leave[w]; best_row = s_row[w]; }
        rec->r = best_row;
        rec->leaving = best_leave;
        rec->alpha_r = alpha[best_row];
        rec->b_r = b[best_row];
        s_row[0] = best_row;
    }
    if (du.kmax <= 0) return;
    // deferred update bookkeeping (k_eta_prepare): save row r of W, choose the column that receives u
    __syncthreads();
    const int r = s_row[0];
    for (int j = threadIdx.x; j < p; j += BS) du.wr[j] = du.W[(int64_t)j * du.ld + r];
    if (threadIdx.x == 0) {
        int jt = du.pos_of_row[r];
        rec->n_eta_old = p;
        if (jt < 0) { jt = p; du.S[p] = r; du.pos_of_row[r] = p; rec->n_eta = p + 1; }
        rec->eta_target = jt;
    }
}

__global__ __launch_bounds__(kSingleBlock) void k_ratio(const double* __restrict__ alpha,
                                                        const double* __restrict__ b,
                                                        const int32_t* __restrict__ basis_indices, int m,
                                                        Tolerances tol, DeferredUpdate du, PivotRecord* rec) {
    const int outcome = rec->outcome, p = rec->n_eta;          // one round trip for both
    if (outcome != DEV_RUNNING) return;
    ratio_body<kSingleBlock, 16>(alpha, b, basis_indices, m, tol, du, p, rec);
}

// ------------------------------------------------------------------------------------------------
// UPDATE
// ------------------------------------------------------------------------------------------------
// rho = normalised pivot row (row r of the NEW inverse), staged outside B^-1 so that the rank-1
// kernel can overwrite row r without racing its readers.
__global__ void k_compute_rho(const double* __restrict__ Binv, int64_t ld_b, int m, int row_lo, int row_hi,
                              double* __restrict__ rho, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = rec->r;
    const bool own = (r >= row_lo && r < row_hi);
    if (j == 0) rec->owner_has_row = own ? 1 : 0;
    if (j >= (int)ld_b) return;
    double v = 0.0;
    if (own && j < m) {
        v = Binv[(int64_t)r * ld_b + j] / rec->alpha_r;            // element_wise_divide, sparse.rs:291
    }
    rho[j] = v;
}

// b_r /= alpha_r; b_i -= alpha_i b_r; -pi -= d_q rho; -obj -= d_q b_r; basis_indices[r] = q
// (carry/mod.rs:283-333, 549-570; tableau/mod.rs:72-84)
__global__ void k_update_vectors(int m, const double* __restrict__ alpha, const double* __restrict__ rho,
                                 double* __restrict__ b, double* __restrict__ minus_pi,
                                 int32_t* __restrict__ basis_indices, uint8_t* __restrict__ in_basis,
                                 int32_t* __restrict__ trace, int64_t trace_cap, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = rec->r;
    const double d_q = rec->d_q;
    const double br = rec->b_r / rec->alpha_r;
    if (i < m) {
        minus_pi[i] = fma(-d_q, rho[i], minus_pi[i]);
        if (i == r) b[i] = br;
        else {
            const double a = alpha[i];
            if (a != 0.0) b[i] = fma(-a, br, b[i]);
        }
    }
    if (i == 0) {
        const int q = rec->q, leaving = rec->leaving;
        rec->minus_objective = fma(-d_q, br, rec->minus_objective);
        basis_indices[r] = q;
        if (leaving < kWrappedArtificialBase) in_basis[leaving] = 0;   // a wrapped artificial has no flag
        in_basis[q] = 1;
        const long long it = rec->iterations;
        if (trace && it < trace_cap) {
            trace[0 * trace_cap + it] = rec->phase;
            trace[1 * trace_cap + it] = q;
            trace[2 * trace_cap + it] = r;
            trace[3 * trace_cap + it] = leaving;
        }
        rec->iterations = it + 1;
    }
}

// Rank-1 update of the explicit inverse: row_r := rho, row_i := row_i - alpha_i * rho  (i != r).
// Grid = (column strips of 512) x (row chunks); each thread owns one 16-byte column pair of the
// strip, keeps its rho pair in registers and streams the chunk's rows read-modify-write.  Rows
// with alpha_i == 0 are skipped (wave-uniform), like the reference skips absent entries.
static constexpr int kUpdRowsPerBlock = 32;
__global__ __launch_bounds__(kThreads) void k_update_inverse(double* __restrict__ Binv, int64_t ld_b, int m,
                                                             int row_lo, int row_hi,
                                                             const double* __restrict__ alpha,
                                                             const double* __restrict__ rho,
                                                             const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int c = (blockIdx.x * kThreads + threadIdx.x) * 2;
    if (c >= (int)ld_b) return;
    const int r = rec->r;
    const double2 rh = *reinterpret_cast<const double2*>(rho + c);
    const int i0 = row_lo + blockIdx.y * kUpdRowsPerBlock;
    const int i1 = min(i0 + kUpdRowsPerBlock, row_hi);
#pragma unroll 4
    for (int i = i0; i < i1; ++i) {
        double2* p = reinterpret_cast<double2*>(Binv + (int64_t)i * ld_b + c);
        if (i == r) {
            *p = rh;
        } else {
            const double a = alpha[i];
            if (a != 0.0) {
                double2 v = *p;
                v.x = fma(-a, rh.x, v.x);
                v.y = fma(-a, rh.y, v.y);
                *p = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Deferred (blocked) update:  B^-1 = (I + W S') B0inv   -- see DeferredUpdate in relp_kernels.h
// ------------------------------------------------------------------------------------------------
static constexpr int kMaxEta = 128;

// alpha = M v = v + W (S' v).  One thread per row; the p gathered entries v[S[j]] sit in LDS.
__global__ __launch_bounds__(kThreads) void k_apply_w(DeferredUpdate du, int m, const double* __restrict__ v,
                                                      double* __restrict__ alpha, const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_vs[kMaxEta];
    const int p = rec->n_eta;
    if ((int)threadIdx.x < p) s_vs[threadIdx.x] = v[du.S[threadIdx.x]];
    __syncthreads();
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    double a = v[i];
    for (int j = 0; j < p; ++j) a = fma(du.W[(int64_t)j * du.ld + i], s_vs[j], a);
    alpha[i] = a;
}

// One wavefront-sized workgroup: save row r of W, choose the column that will receive u.
__global__ void k_eta_prepare(DeferredUpdate du, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int p = rec->n_eta, r = rec->r;
    __syncthreads();                         // every wavefront has read n_eta before lane 0 bumps it
    for (int j = threadIdx.x; j < p; j += blockDim.x) du.wr[j] = du.W[(int64_t)j * du.ld + r];
    if (threadIdx.x == 0) {
        int jt = du.pos_of_row[r];
        rec->n_eta_old = p;
        if (jt < 0) { jt = p; du.S[p] = r; du.pos_of_row[r] = p; rec->n_eta = p + 1; }
        rec->eta_target = jt;
    }
}

// E_k = I + u e_r' with u_r = 1/alpha_r - 1, u_i = -alpha_i/alpha_r:
//   (I + u e_r')(I + W S') = I + (W + u W[r,:]) S' + u e_r'
__global__ __launch_bounds__(kThreads) void k_update_w(DeferredUpdate du, int m, const double* __restrict__ alpha,
                                                       const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_wr[kMaxEta];
    const int p_old = rec->n_eta_old, jt = rec->eta_target, r = rec->r;
    if ((int)threadIdx.x < p_old) s_wr[threadIdx.x] = du.wr[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    const double ar = rec->alpha_r;
    const double u = (i == r) ? (1.0 / ar - 1.0) : (-alpha[i] / ar);
    if (u != 0.0) {
        for (int j = 0; j < p_old; ++j) {
            const double w = s_wr[j];
            if (w != 0.0) du.W[(int64_t)j * du.ld + i] = fma(u, w, du.W[(int64_t)j * du.ld + i]);
        }
    }
    double* tgt = du.W + (int64_t)jt * du.ld + i;
    if (jt < p_old) *tgt += u; else *tgt = u;
}

// rho = e_r' (I + W S') B0inv restricted to the rows this rank owns (a SUM over ranks completes it).
__global__ __launch_bounds__(kThreads) void k_rho_deferred(DeferredUpdate du, const double* __restrict__ Binv,
                                                           int64_t ld_b, int m, int row_lo, int row_hi,
                                                           double* __restrict__ rho, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_coef[kMaxEta];
    __shared__ int s_row[kMaxEta];
    const int p = rec->n_eta, r = rec->r;
    if ((int)threadIdx.x < p) {
        const int row = du.S[threadIdx.x];
        s_row[threadIdx.x] = row;
        s_coef[threadIdx.x] = (row >= row_lo && row < row_hi) ? du.W[(int64_t)threadIdx.x * du.ld + r] : 0.0;
    }
    __syncthreads();
    const int c = (blockIdx.x * kThreads + threadIdx.x) * 2;
    if (blockIdx.x == 0 && threadIdx.x == 0) rec->owner_has_row = (r >= row_lo && r < row_hi) ? 1 : 0;
    if (c >= (int)ld_b) return;
    double2 acc = make_double2(0.0, 0.0);
    if (r >= row_lo && r < row_hi) acc = *reinterpret_cast<const double2*>(Binv + (int64_t)r * ld_b + c);
    // rows with a zero coefficient are skipped through a clamped row index (no divergent branch, loads
    // stay independent so eight of them are in flight per lane); the sum order is j ascending.
    const int safe = (r >= row_lo && r < row_hi) ? r : row_lo;
    int j = 0;
    for (; j + 8 <= p; j += 8) {
        double2 bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = (s_coef[j + u] != 0.0) ? s_row[j + u] : safe;
            bv[u] = *reinterpret_cast<const double2*>(Binv + (int64_t)row * ld_b + c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc.x = fma(s_coef[j + u], bv[u].x, acc.x);
            acc.y = fma(s_coef[j + u], bv[u].y, acc.y);
        }
    }
    for (; j < p; ++j) {
        const double w = s_coef[j];
        if (w != 0.0) {
            const double2 bv = *reinterpret_cast<const double2*>(Binv + (int64_t)s_row[j] * ld_b + c);
            acc.x = fma(w, bv.x, acc.x);
            acc.y = fma(w, bv.y, acc.y);
        }
    }
    *reinterpret_cast<double2*>(rho + c) = acc;
}

// R[j,:] = B0inv[S[j],:] for the rows this rank owns, zero otherwise (a SUM over ranks completes it).
__global__ void k_flush_snapshot(DeferredUpdate du, const double* __restrict__ Binv, int64_t ld_b, int row_lo,
                                 int row_hi, const PivotRecord* rec) {
    const int p = rec->n_eta;
    const int j = blockIdx.y;
    if (j >= p) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int)ld_b) return;
    const int row = du.S[j];
    du.R[(int64_t)j * du.ld + c] = (row >= row_lo && row < row_hi) ? Binv[(int64_t)row * ld_b + c] : 0.0;
}

// B0inv[i, c] += sum_j W[i, j] R[j, c]   (m x p x m GEMM, p <= kmax).  64 x 64 output tile per
// workgroup, W and R tiles staged through LDS, 4 x 4 outputs per thread.
static constexpr int kFT = 64;   // tile edge
static constexpr int kFK = 16;   // k-chunk
__global__ __launch_bounds__(kThreads) void k_flush_apply(DeferredUpdate du, double* __restrict__ Binv, int64_t ld_b,
                                                          int m, int row_lo, int row_hi, const PivotRecord* rec) {
    const int p = rec->n_eta;
    if (p == 0) return;
    __shared__ double s_w[kFK][kFT + 1];   // W tile, [k][row]
    __shared__ double s_r[kFK][kFT];       // R tile, [k][col]
    const int i0 = row_lo + blockIdx.y * kFT, c0 = blockIdx.x * kFT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, 4 x 4 outputs each
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k0 = 0; k0 < p; k0 += kFK) {
        // 16 x 64 = 1024 entries per tile, 4 per thread
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = threadIdx.x + e * kThreads;
            const int kk = idx >> 6, x = idx & 63;
            const int k = k0 + kk;
            const int i = i0 + x, c = c0 + x;
            s_w[kk][x] = (k < p && i < row_hi) ? du.W[(int64_t)k * du.ld + i] : 0.0;
            s_r[kk][x] = (k < p && c < (int)ld_b) ? du.R[(int64_t)k * du.ld + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kFK; ++kk) {
            double wv[4], rv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) wv[a] = s_w[kk][ty * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) rv[b] = s_r[kk][tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fma(wv[a], rv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ty * 4 + a;
        if (i >= row_hi) continue;
        const int c = c0 + tx * 4;
        if (c + 3 < (int)ld_b) {
            double2* p0 = reinterpret_cast<double2*>(Binv + (int64_t)i * ld_b + c);
            double2 v0 = p0[0], v1 = p0[1];
            v0.x += acc[a][0]; v0.y += acc[a][1]; v1.x += acc[a][2]; v1.y += acc[a][3];
            p0[0] = v0; p0[1] = v1;
        }
    }
}

__global__ void k_flush_reset(DeferredUpdate du, PivotRecord* rec) {
    const int p = rec->n_eta;
    for (int j = threadIdx.x; j < p; j += blockDim.x) du.pos_of_row[du.S[j]] = -1;
    __syncthreads();
    if (threadIdx.x == 0) { rec->n_eta = 0; rec->n_eta_old = 0; rec->eta_target = 0; }
}


// ------------------------------------------------------------------------------------------------
// Dense-tableau engine:  T = (I + W S') T0  (see TableauView / DeferredUpdate in relp_kernels.h)
//   PRICE  = one row of T per pivot:   d <- d - (d_q / alpha_r) T[r,:]      (instead of 8 m n_s bytes)
//   FTRAN  = one column of T per pivot: alpha = T0[:,q] + W R0[:,q]        (instead of 8 m^2 bytes)
//   UPDATE = W <- E W per pivot; T0 += W R0 once per K pivots on the f64 matrix cores
// ------------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_tab_build(TableauView tv, const double* __restrict__ A, int64_t ld_a, ColumnTable ct) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)tv.m * (tv.c_hi - tv.c_lo);
    if (idx >= total) return;
    const int c = tv.c_lo + (int)(idx / tv.m), i = (int)(idx % tv.m);
    double v = 0.0;
    if (c < ct.nr_artificial) {
        v = (i == ct.column_to_row[c]) ? 1.0 : 0.0;
    } else {
        const int p = c - ct.nr_artificial;
        if (p < ct.nr_normal) {
            if (i < ct.nr_constraints) v = A[(int64_t)p * ld_a + i];
            else v = (i == ct.bound_row[p]) ? 1.0 : 0.0;
        } else {
            const int vv = p - ct.nr_normal;
            if (i == ct.vrow0[vv]) v = (double)ct.vsign[vv];
            else if (i == ct.vrow1[vv]) v = 1.0;
        }
    }
    tv.T0[(int64_t)c * tv.ld_t + i] = v;
}

// d[c] = cost[c] - w . T0[:,c]   (the PRICE multi-dot over the stored tableau, phase boundaries only)
__global__ __launch_bounds__(kThreads) void k_tab_price_init(TableauView tv, const double* __restrict__ w,
                                                             const double* __restrict__ cost_store) {
    __shared__ double s_partial[4 * kVecPerBlock];
    const int v0 = tv.c_lo + blockIdx.x * kVecPerBlock;
    double dot = 0.0;
    block_multi_dot(tv.T0, tv.ld_t, tv.m, v0, tv.c_hi, w, s_partial, dot);
    const int c = v0 + threadIdx.x;
    if (threadIdx.x < kVecPerBlock && c < tv.c_hi) tv.d[c] = cost_store[c] - dot;
}

// one slot per 256 storage columns
__global__ __launch_bounds__(kThreads) void k_tab_scan(TableauView tv, SelectPartials sp, const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int c = tv.c_lo + blockIdx.x * kThreads + threadIdx.x;
    const int j = c - tv.col_off;
    double key = INFINITY;
    int kj = 0x7fffffff;
    if (c < tv.c_hi && j >= 0 && j < tv.n) {
        const double v = tv.d[c];
        if (!sp.in_basis[j] && v < -sp.tol_cost) { key = select_key(sp.rule, sp.n, rec, j, v); kj = j; }
    }
    block_partial_min(key, kj, sp, blockIdx.x);
}

__global__ __launch_bounds__(kSingleBlock) void k_tab_select(TableauView tv, SelectPartials sp, int count,
                                                             PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_k1[kSingleBlock / 64];
    __shared__ int s_j[kSingleBlock / 64];
    double k1 = INFINITY;
    int bj = 0x7fffffff;
    for (int t = threadIdx.x; t < count; t += kSingleBlock) {
        const double key = sp.k1[t];
        const int j = sp.j[t];
        if (key < k1 || (key == k1 && j < bj)) { k1 = key; bj = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_down(k1, off, 64);
        const int oj = __shfl_down(bj, off, 64);
        if (ok < k1 || (ok == k1 && oj < bj)) { k1 = ok; bj = oj; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_k1[wave] = k1; s_j[wave] = bj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kSingleBlock / 64; ++w)
            if (s_k1[w] < k1 || (s_k1[w] == k1 && s_j[w] < bj)) { k1 = s_k1[w]; bj = s_j[w]; }
        s_k1[0] = k1;
        s_j[0] = bj;
    }
    __syncthreads();
    k1 = s_k1[0];
    bj = s_j[0];
    __syncthreads();
    if (bj != 0x7fffffff && sp.rule == 2 && sp.tol_tie > 0.0) {
        // Dantzig ties (pivot_rule.rs:118): lowest index inside the band; only slots whose minimum is
        // inside the band can hold such a column
        const double bound = k1 + sp.tol_tie * fmax(1.0, fabs(k1));
        int lowest = 0x7fffffff;
        // four groups of 256 threads walk the slots; a slot inside the band is re-read by its group,
        // one column per thread
        const int grp = threadIdx.x >> 8, u = threadIdx.x & 255;
        for (int t = grp; t < count; t += kSingleBlock / kThreads) {
            if (!(sp.k1[t] <= bound)) continue;
            const int c = tv.c_lo + t * kThreads + u;
            const int j = c - tv.col_off;
            if (c < tv.c_hi && j >= 0 && j < tv.n) {
                const double v = tv.d[c];
                if (!sp.in_basis[j] && v < -sp.tol_cost && v <= bound && j < lowest) lowest = j;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lowest = min(lowest, __shfl_down(lowest, off, 64));
        if (lane == 0) s_j[wave] = lowest;
        __syncthreads();
        if (threadIdx.x == 0) {
            int low = 0x7fffffff;
            for (int w = 0; w < kSingleBlock / 64; ++w) low = min(low, s_j[w]);
            bj = low;
        }
    }
    if (threadIdx.x == 0) {
        if (bj == 0x7fffffff) {
            rec->outcome = DEV_NO_CANDIDATE;
            if (sp.rule == 1) rec->last_selected = -1;
        } else {
            rec->q = bj;
            rec->d_q = tv.d[bj + tv.col_off];
            rec->key1 = k1;
            if (sp.rule == 1) rec->last_selected = bj;
        }
    }
}

// alpha = T[:,q] = T0[:,q] + W (R0[:,q])
__global__ __launch_bounds__(kThreads) void k_tab_column(TableauView tv, DeferredUpdate du, double* __restrict__ alpha,
                                                         const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_vs[kMaxEta];
    const int p = rec->n_eta;
    const int cq = rec->q + tv.col_off;
    if ((int)threadIdx.x < p) s_vs[threadIdx.x] = tv.R0[(int64_t)threadIdx.x * tv.ld_r + cq];
    __syncthreads();
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= tv.m) return;
    double a = tv.T0[(int64_t)cq * tv.ld_t + i];
    for (int j = 0; j < p; ++j) a = fma(du.W[(int64_t)j * du.ld + i], s_vs[j], a);
    alpha[i] = a;
}

// Row r of T before the pivot, the reduced-cost update and the next PRICE's partial argmin in one
// pass over the stored columns.  When row r is new in the block its T0 row is appended to R0 here.
// `block` = index of this workgroup among the row-update workgroups.
// `R` is the workgroup's snapshot of the PivotRecord (one cache line, fetched once at kernel start: reading
// it field by field between stores costs a dependent memory round trip each time).
__device__ __forceinline__ void tab_row_update_body(const TableauView& tv, const DeferredUpdate& du,
                                                    const SelectPartials& sp, const PivotRecord& R, int block) {
    __shared__ double s_wr[kMaxEta];
    const PivotRecord* rec = &R;
    const int p_old = R.n_eta_old, jt = R.eta_target, r = R.r, q = R.q, leaving = R.leaving;
    // fetched without waiting for p_old (entries beyond it are never used)
    if ((int)threadIdx.x < du.kmax) s_wr[threadIdx.x] = du.wr[threadIdx.x];
    const int c = tv.c_lo + block * kThreads + threadIdx.x;
    const double d_old = c < tv.c_hi ? tv.d[c] : 0.0;
    __syncthreads();
    double key = INFINITY;
    int kj = 0x7fffffff;
    if (c < tv.c_hi) {
        double base;
        if (jt < p_old) base = tv.R0[(int64_t)jt * tv.ld_r + c];
        else { base = tv.T0[(int64_t)c * tv.ld_t + r]; tv.R0[(int64_t)jt * tv.ld_r + c] = base; }
        double row = base;
        for (int j = 0; j < p_old; ++j) row = fma(s_wr[j], tv.R0[(int64_t)j * tv.ld_r + c], row);
        const double theta = R.d_q / R.alpha_r;
        const int j = c - tv.col_off;
        double dn = fma(-theta, row, d_old);
        if (j == q) dn = 0.0;
        tv.d[c] = dn;
        if (j >= 0 && j < tv.n) {
            // correct for old AND new flags (the flags may be flipped concurrently by the W/vector part)
            const bool basic = (j == q) || (sp.in_basis[j] && j != leaving);
            if (!basic && dn < -sp.tol_cost) { key = select_key(sp.rule, sp.n, rec, j, dn); kj = j; }
        }
    }
    block_partial_min(key, kj, sp, block);
}

__global__ __launch_bounds__(kThreads) void k_tab_row_update(TableauView tv, DeferredUpdate du, SelectPartials sp,
                                                             PivotRecord* rec) {
    const PivotRecord R = *rec;
    if (R.outcome != DEV_RUNNING) return;
    tab_row_update_body(tv, du, sp, R, blockIdx.x);
}

// W <- E W  and  b, -obj, basis, flags, trace (both walk the m rows)
__device__ __forceinline__ void tab_update_w_vectors_body(const DeferredUpdate& du, int m, const double* __restrict__ alpha,
                                                          double* __restrict__ b, int32_t* __restrict__ basis_indices,
                                                          uint8_t* __restrict__ in_basis, int32_t* __restrict__ trace,
                                                          int64_t trace_cap, const PivotRecord& R, PivotRecord* rec,
                                                          int block) {
    __shared__ double s_wr2[kMaxEta];
    const int p_old = R.n_eta_old, jt = R.eta_target, r = R.r;
    if ((int)threadIdx.x < du.kmax) s_wr2[threadIdx.x] = du.wr[threadIdx.x];
    const int i = block * kThreads + threadIdx.x;
    const double a_i = i < m ? alpha[i] : 0.0;
    const double b_i = i < m ? b[i] : 0.0;
    __syncthreads();
    const double ar = R.alpha_r;
    const double br = R.b_r / ar;
    if (i < m) {
        const double a = a_i;
        const double u = (i == r) ? (1.0 / ar - 1.0) : (-a / ar);
        if (u != 0.0) {
            for (int j = 0; j < p_old; ++j) {
                const double w = s_wr2[j];
                if (w != 0.0) du.W[(int64_t)j * du.ld + i] = fma(u, w, du.W[(int64_t)j * du.ld + i]);
            }
        }
        double* tgt = du.W + (int64_t)jt * du.ld + i;
        if (jt < p_old) *tgt += u; else *tgt = u;
        if (i == r) b[i] = br;
        else if (a != 0.0) b[i] = fma(-a, br, b_i);
    }
    if (i == 0) {
        const int q = R.q, leaving = R.leaving;
        rec->minus_objective = fma(-R.d_q, br, R.minus_objective);
        basis_indices[r] = q;
        if (leaving < kWrappedArtificialBase) in_basis[leaving] = 0;   // a wrapped artificial has no flag
        in_basis[q] = 1;
        const long long it = R.iterations;
        if (trace && it < trace_cap) {
            trace[0 * trace_cap + it] = R.phase;
            trace[1 * trace_cap + it] = q;
            trace[2 * trace_cap + it] = r;
            trace[3 * trace_cap + it] = leaving;
        }
        rec->iterations = it + 1;
    }
}

__global__ __launch_bounds__(kThreads) void k_tab_update_w_vectors(DeferredUpdate du, int m,
                                                                   const double* __restrict__ alpha,
                                                                   double* __restrict__ b,
                                                                   int32_t* __restrict__ basis_indices,
                                                                   uint8_t* __restrict__ in_basis,
                                                                   int32_t* __restrict__ trace, int64_t trace_cap,
                                                                   PivotRecord* rec) {
    const PivotRecord R = *rec;
    if (R.outcome != DEV_RUNNING) return;
    tab_update_w_vectors_body(du, m, alpha, b, basis_indices, in_basis, trace, trace_cap, R, rec, blockIdx.x);
}

// Both halves of the update in ONE launch: workgroups [0, nb_row) update the tableau row / reduced
// costs / PRICE partials of their columns, workgroups [nb_row, ..) update W, b and the bookkeeping.
// The halves touch disjoint data; the basis flags flipped by the second half are read by the first
// through an expression that is the same for the old and the new flags.
__global__ __launch_bounds__(kThreads) void k_tab_update_all(TableauView tv, DeferredUpdate du, SelectPartials sp,
                                                             int nb_row, int m, const double* __restrict__ alpha,
                                                             double* __restrict__ b, int32_t* __restrict__ basis_indices,
                                                             uint8_t* __restrict__ in_basis, int32_t* __restrict__ trace,
                                                             int64_t trace_cap, PivotRecord* rec) {
    const PivotRecord R = *rec;
    if (R.outcome != DEV_RUNNING) return;
    if ((int)blockIdx.x < nb_row) tab_row_update_body(tv, du, sp, R, blockIdx.x);
    else tab_update_w_vectors_body(du, m, alpha, b, basis_indices, in_basis, trace, trace_cap, R, rec, blockIdx.x - nb_row);
}

// PRICE's final reduction and the tableau column in one launch: every workgroup reduces the (few)
// partials to the same entering column q, then forms alpha = T0[:,q] + W R0[:,q] for its rows.
// `msg` (sharded engines): the candidate message [key, j, d_j, alpha(m)] of this rank is written instead
// of the record; a rank without a candidate sends key = +inf and stays RUNNING (another rank may have one).
__global__ __launch_bounds__(kThreads) void k_tab_select_column(TableauView tv, DeferredUpdate du, SelectPartials sp,
                                                                int count, double* __restrict__ alpha, double* msg,
                                                                PivotRecord* rec) {
    const int outcome = rec->outcome, p = rec->n_eta;          // one round trip for both
    if (outcome != DEV_RUNNING) return;
    __shared__ double s_k1[kThreads / 64];
    __shared__ int s_j[kThreads / 64];
    __shared__ double s_vs[kMaxEta];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double k1 = INFINITY;
    int bj = 0x7fffffff;
    for (int t = threadIdx.x; t < count; t += kThreads) {
        const double key = sp.k1[t];
        const int j = sp.j[t];
        if (key < k1 || (key == k1 && j < bj)) { k1 = key; bj = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_down(k1, off, 64);
        const int oj = __shfl_down(bj, off, 64);
        if (ok < k1 || (ok == k1 && oj < bj)) { k1 = ok; bj = oj; }
    }
    if (lane == 0) { s_k1[wave] = k1; s_j[wave] = bj; }
    __syncthreads();
    k1 = s_k1[0]; bj = s_j[0];
    for (int w = 1; w < kThreads / 64; ++w)
        if (s_k1[w] < k1 || (s_k1[w] == k1 && s_j[w] < bj)) { k1 = s_k1[w]; bj = s_j[w]; }
    __syncthreads();
    if (bj != 0x7fffffff && sp.rule == 2 && sp.tol_tie > 0.0) {
        // Dantzig tie band (see k_tab_select): slots inside the band are re-read, one column per thread
        const double bound = k1 + sp.tol_tie * fmax(1.0, fabs(k1));
        int lowest = 0x7fffffff;
        for (int t = 0; t < count; ++t) {
            if (!(sp.k1[t] <= bound)) continue;
            const int c = tv.c_lo + t * kThreads + threadIdx.x;
            const int j = c - tv.col_off;
            if (c < tv.c_hi && j >= 0 && j < tv.n) {
                const double v = tv.d[c];
                if (!sp.in_basis[j] && v < -sp.tol_cost && v <= bound && j < lowest) lowest = j;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lowest = min(lowest, __shfl_down(lowest, off, 64));
        if (lane == 0) s_j[wave] = lowest;
        __syncthreads();
        bj = s_j[0];
        for (int w = 1; w < kThreads / 64; ++w) bj = min(bj, s_j[w]);
    }
    if (bj == 0x7fffffff) {
        if (msg) {
            const int i = blockIdx.x * kThreads + threadIdx.x;
            if (i < tv.m) alpha[i] = 0.0;
            if (i == 0) { msg[0] = INFINITY; msg[1] = 0.0; msg[2] = 0.0; }
        } else if (blockIdx.x == 0 && threadIdx.x == 0) {
            rec->outcome = DEV_NO_CANDIDATE;
            if (sp.rule == 1) rec->last_selected = -1;
        }
        return;
    }
    const int cq = bj + tv.col_off;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (msg) { msg[0] = k1; msg[1] = (double)bj; msg[2] = tv.d[cq]; }
        else {
            rec->q = bj;
            rec->d_q = tv.d[cq];
            rec->key1 = k1;
            if (sp.rule == 1) rec->last_selected = bj;
        }
    }
    if ((int)threadIdx.x < p) s_vs[threadIdx.x] = tv.R0[(int64_t)threadIdx.x * tv.ld_r + cq];
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const double t0 = i < tv.m ? tv.T0[(int64_t)cq * tv.ld_t + i] : 0.0;     // in flight together with the R0 column
    __syncthreads();
    if (i >= tv.m) return;
    double a = t0;
    for (int j = 0; j < p; ++j) a = fma(du.W[(int64_t)j * du.ld + i], s_vs[j], a);
    alpha[i] = a;
}

__global__ void k_tab_update_vectors(int m, const double* __restrict__ alpha, double* __restrict__ b,
                                     int32_t* __restrict__ basis_indices, uint8_t* __restrict__ in_basis,
                                     int32_t* __restrict__ trace, int64_t trace_cap, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = rec->r;
    const double br = rec->b_r / rec->alpha_r;
    if (i < m) {
        if (i == r) b[i] = br;
        else {
            const double a = alpha[i];
            if (a != 0.0) b[i] = fma(-a, br, b[i]);
        }
    }
    if (i == 0) {
        const int q = rec->q, leaving = rec->leaving;
        rec->minus_objective = fma(-rec->d_q, br, rec->minus_objective);
        basis_indices[r] = q;
        if (leaving < kWrappedArtificialBase) in_basis[leaving] = 0;   // a wrapped artificial has no flag
        in_basis[q] = 1;
        const long long it = rec->iterations;
        if (trace && it < trace_cap) {
            trace[0 * trace_cap + it] = rec->phase;
            trace[1 * trace_cap + it] = q;
            trace[2 * trace_cap + it] = r;
            trace[3 * trace_cap + it] = leaving;
        }
        rec->iterations = it + 1;
    }
}

// Flush: T0 += W R0 on the f64 matrix cores (v_mfma_f64_16x16x4_f64).  The MFMA computes the
// transposed tile (R0' W')  so that the fast lane index of the accumulator runs along the rows of
// T0, which are contiguous (column-major): stores are 128-byte segments.
//   A operand (16 x 4): A[M][k] = R0[k][c0 + M]      lane l: M = l & 15, k = l >> 4
//   B operand (4 x 16): B[k][N] = W[i0 + N][k]       lane l: N = l & 15, k = l >> 4
//   D (16 x 16):        D[M][N] -> T0[i0 + N, c0 + M], lane l holds N = l & 15, M = (l >> 4) + 4 g, g = 0..3
// Wavefront tile 64 columns x 64 rows (4 x 4 MFMA tiles, 8 operand loads per 16 MFMAs), workgroup
// 128 x 128.
template <int MT, int NT>
__global__ __launch_bounds__(kThreads) void k_tab_flush(TableauView tv, DeferredUpdate du, const PivotRecord* rec) {
    constexpr int kFlushMT = MT, kFlushNT = NT;
    const int p = rec->n_eta;
    if (p == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c_wave = tv.c_lo + blockIdx.x * (2 * 16 * MT) + (wave & 1) * (16 * MT);   // first T0 column of this wavefront
    const int i_wave = blockIdx.y * (2 * 16 * NT) + (wave >> 1) * (16 * NT);            // first T0 row
    if (c_wave >= tv.c_hi || i_wave >= tv.m) return;
    const int lm = lane & 15, lk = lane >> 4;
    // the accumulators start as the T0 tile itself: all of its loads are in flight before the first MFMA
    double4_t acc[kFlushMT][kFlushNT];
#pragma unroll
    for (int a = 0; a < kFlushMT; ++a)
#pragma unroll
        for (int b = 0; b < kFlushNT; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c_wave + a * 16 + lk + 4 * g;
                const int i = i_wave + b * 16 + lm;
                acc[a][b][g] = (c < tv.c_hi && i < tv.m) ? tv.T0[(int64_t)c * tv.ld_t + i] : 0.0;
            }
    // operand fragments of step k0 + 4 are requested before the MFMAs of step k0 are issued, so their
    // L2 latency (~1-2 us) overlaps the 8 x 64-cycle MFMAs instead of serialising with them
    double af[kFlushMT], bf[kFlushNT], afn[kFlushMT], bfn[kFlushNT];
    auto load_frags = [&](int k0, double* fa, double* fb) {
        const int k = k0 + lk;
        const bool kv = k < p;
#pragma unroll
        for (int a = 0; a < kFlushMT; ++a) {
            const int c = c_wave + a * 16 + lm;
            fa[a] = (kv && c < tv.c_hi) ? tv.R0[(int64_t)k * tv.ld_r + c] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < kFlushNT; ++b) {
            const int i = i_wave + b * 16 + lm;
            fb[b] = (kv && i < tv.m) ? du.W[(int64_t)k * du.ld + i] : 0.0;
        }
    };
    load_frags(0, af, bf);
    for (int k0 = 0; k0 < p; k0 += 4) {
        load_frags(k0 + 4, afn, bfn);                 // k >= p yields zeros, no branch
#pragma unroll
        for (int a = 0; a < kFlushMT; ++a)
#pragma unroll
            for (int b = 0; b < kFlushNT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < kFlushMT; ++a) af[a] = afn[a];
#pragma unroll
        for (int b = 0; b < kFlushNT; ++b) bf[b] = bfn[b];
    }
#pragma unroll
    for (int a = 0; a < kFlushMT; ++a)
#pragma unroll
        for (int b = 0; b < kFlushNT; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c_wave + a * 16 + lk + 4 * g;
                const int i = i_wave + b * 16 + lm;
                if (c < tv.c_hi && i < tv.m) tv.T0[(int64_t)c * tv.ld_t + i] = acc[a][b][g];
            }
}

// The same product with the operands staged through LDS.  A workgroup of WC x WR wavefronts owns a
// (WC * 64 columns) x (WR * 32 rows) tile of T0; per chunk of KC pivots of the block it copies the
// R0 rows (KC x TC) and W columns (KC x TR) it needs into LDS once (double-buffered, 16-byte global
// loads issued one chunk ahead) and every wavefront takes its MFMA fragments from there.  Without this
// each wavefront fetches its own operands from L2 / Infinity Cache: 48 KB per 32 KB of T0 traffic
// (4.7 GB per flush at 10k x 20k against 3.2 GB of HBM traffic); with a 256 x 128 tile it is 12 KB.
template <int WC, int WR, int KC>
__global__ __launch_bounds__(WC * WR * 64) void k_tab_flush_lds(TableauView tv, DeferredUpdate du, const PivotRecord* rec) {
    constexpr int MT = 4, NT = 2;                     // wavefront tile: 64 columns x 32 rows
    constexpr int TC = WC * 16 * MT, TR = WR * 16 * NT, NTHR = WC * WR * 64;
    constexpr int RA = KC * TC / 2 / NTHR, RB = KC * TR / 2 / NTHR;
    static_assert(RA * NTHR * 2 == KC * TC && RB * NTHR * 2 == KC * TR, "staging must divide evenly");
    const int p = rec->n_eta;
    if (p == 0) return;
    __shared__ __align__(16) double As[2][KC][TC];
    __shared__ __align__(16) double Bs[2][KC][TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave % WC, wr = wave / WC;
    // consecutive workgroups walk down the rows of the same 256 columns: T0 is column-major, so the
    // workgroups in flight stream whole columns (sequential DRAM pages) and share one R0 chunk in L2
    const int c_blk = tv.c_lo + blockIdx.y * TC, i_blk = blockIdx.x * TR;
    const int c_wave = c_blk + wc * 16 * MT, i_wave = i_blk + wr * 16 * NT;
    const bool active = c_wave < tv.c_hi && i_wave < tv.m;
    const int lm = lane & 15, lk = lane >> 4;
    const int c_end = tv.c_lo + (int)tv.ld_r;         // R0 rows are readable up to their (even) pitch
    double2 ra[RA], rb[RB];
    auto gload = [&](int kc) {
#pragma unroll
        for (int u = 0; u < RA; ++u) {
            const int idx = tid + NTHR * u, k = idx / (TC / 2), c = c_blk + 2 * (idx % (TC / 2));
            ra[u] = (kc + k < p && c < c_end) ? *reinterpret_cast<const double2*>(tv.R0 + (int64_t)(kc + k) * tv.ld_r + c)
                                              : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int idx = tid + NTHR * u, k = idx / (TR / 2), i = i_blk + 2 * (idx % (TR / 2));
            rb[u] = (kc + k < p && i < (int)du.ld) ? *reinterpret_cast<const double2*>(du.W + (int64_t)(kc + k) * du.ld + i)
                                                   : make_double2(0.0, 0.0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < RA; ++u) {
            const int idx = tid + NTHR * u;
            *reinterpret_cast<double2*>(&As[buf][idx / (TC / 2)][2 * (idx % (TC / 2))]) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int idx = tid + NTHR * u;
            *reinterpret_cast<double2*>(&Bs[buf][idx / (TR / 2)][2 * (idx % (TR / 2))]) = rb[u];
        }
    };
    gload(0);
    // the accumulators start as the T0 tile itself
    double4_t acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c_wave + a * 16 + lk + 4 * g;
                const int i = i_wave + b * 16 + lm;
                acc[a][b][g] = (active && c < tv.c_hi && i < tv.m) ? tv.T0[(int64_t)c * tv.ld_t + i] : 0.0;
            }
    lstore(0);
    __syncthreads();
    const int nchunks = (p + KC - 1) / KC;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) gload((ch + 1) * KC);
        if (active) {
            const int kmax = min(KC, p - ch * KC);
            for (int k0 = 0; k0 < kmax; k0 += 4) {
                double af[MT], bf[NT];
#pragma unroll
                for (int a = 0; a < MT; ++a) af[a] = As[buf][k0 + lk][wc * 16 * MT + a * 16 + lm];
#pragma unroll
                for (int b = 0; b < NT; ++b) bf[b] = Bs[buf][k0 + lk][wr * 16 * NT + b * 16 + lm];
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) lstore(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c_wave + a * 16 + lk + 4 * g;
                const int i = i_wave + b * 16 + lm;
                if (c < tv.c_hi && i < tv.m) tv.T0[(int64_t)c * tv.ld_t + i] = acc[a][b][g];
            }
}

__global__ void k_tab_gather_columns(TableauView tv, const int32_t* __restrict__ cols, double* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)tv.m * tv.m;
    if (idx >= total) return;
    const int k = (int)(idx / tv.m), i = (int)(idx % tv.m);        // consecutive threads walk down a column of T0
    out[(int64_t)i * tv.m + k] = tv.T0[(int64_t)cols[k] * tv.ld_t + i];
}

__global__ __launch_bounds__(kThreads) void k_tab_row(TableauView tv, DeferredUpdate du, int row, double* __restrict__ out,
                                                      const PivotRecord* rec) {
    __shared__ double s_w[kMaxEta];
    const int p = rec->n_eta;
    if ((int)threadIdx.x < p) s_w[threadIdx.x] = du.W[(int64_t)threadIdx.x * du.ld + row];
    __syncthreads();
    const int c = tv.c_lo + blockIdx.x * kThreads + threadIdx.x;
    if (c >= tv.c_hi) return;
    double v = tv.T0[(int64_t)c * tv.ld_t + row];
    for (int j = 0; j < p; ++j) v = fma(s_w[j], tv.R0[(int64_t)j * tv.ld_r + c], v);
    out[c - tv.c_lo] = v;
}

// ------------------------------------------------------------------------------------------------
// Phase switch, identity, synthetic fill
// ------------------------------------------------------------------------------------------------
__global__ void k_weighted_column_sums(const double* __restrict__ Binv, int64_t ld_b, int m,
                                       const double* __restrict__ w, double* __restrict__ minus_pi) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int i = 0; i < m; ++i) {
        const double wi = w[i];
        if (wi != 0.0) s = fma(Binv[(int64_t)i * ld_b + j], wi, s);
    }
    minus_pi[j] = -s;
}

// rows [row_lo, row_hi) of the identity, stored locally starting at row 0
__global__ void k_set_identity(double* __restrict__ Binv, int64_t ld_b, int row_lo, int row_hi) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)(row_hi - row_lo) * ld_b;
    if (idx >= total) return;
    const int64_t i = idx / ld_b, j = idx % ld_b;
    Binv[idx] = (row_lo + i == j) ? 1.0 : 0.0;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t seed, uint64_t stream, uint64_t idx) {
    uint64_t z = seed + stream * 0xD1B54A32D192ED03ull + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// A[i, j] = (1 + x(0, (first_column + j) * m + i) % 999) / 1000   (rust-lp_amd/synthetic.py)
__global__ void k_fill_dense(double* __restrict__ A, int64_t ld, int m, int n, uint64_t seed,
                             int64_t first_column) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)m * n;
    if (idx >= total) return;
    const int64_t j = idx / m, i = idx % m;
    const uint64_t x = splitmix64(seed, 0, (uint64_t)((first_column + j) * m + i));
    A[j * ld + i] = (double)(1 + x % 999) / 1000.0;
}

// ------------------------------------------------------------------------------------------------
// Sharded pricing / FTRAN helpers
// ------------------------------------------------------------------------------------------------
// msg = [key1, j, d_j, a_j[0..m)]; key1 = +inf when this rank has no candidate.  Resets the local
// "no candidate" so that only the GLOBAL decision freezes the loop.
__global__ void k_pack_candidate(const double* __restrict__ aq, int m, double* __restrict__ msg,
                                 PivotRecord* rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int outcome = rec->outcome;
    if (outcome == DEV_NO_ROW) return;
    if (i < m) msg[3 + i] = (outcome == DEV_RUNNING) ? aq[i] : 0.0;
    if (i == 0) {
        if (outcome == DEV_RUNNING) {
            msg[0] = rec->key1; msg[1] = (double)rec->q; msg[2] = rec->d_q;
        } else {
            msg[0] = INFINITY; msg[1] = 0.0; msg[2] = 0.0;
        }
    }
}

// The local "no candidate" must not freeze this rank: only the global decision does.  Runs after
// k_pack_candidate (separate launch, so every thread of the pack saw the old outcome).
__global__ void k_clear_no_candidate(PivotRecord* rec) {
    if (rec->outcome == DEV_NO_CANDIDATE) rec->outcome = DEV_RUNNING;
}

__global__ __launch_bounds__(kSingleBlock) void k_select_candidate(const double* __restrict__ msgs, int count,
                                                                   int64_t msg_len, int m,
                                                                   double* __restrict__ aq, int rule,
                                                                   double tol_tie, PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ int s_win;
    if (threadIdx.x == 0) {
        int win = -1; double k1 = INFINITY; double kj = 0.0;
        for (int g = 0; g < count; ++g) {
            const double a = msgs[g * msg_len + 0], j = msgs[g * msg_len + 1];
            if (a < k1 || (a == k1 && win >= 0 && j < kj)) { k1 = a; kj = j; win = g; }
        }
        if (win >= 0 && rule == 2 && tol_tie > 0.0) {
            // Dantzig ties across ranks: every rank sent (its minimum, its lowest index within the band
            // of that minimum); the lowest index among the ranks inside the global band wins
            const double bound = k1 + tol_tie * fmax(1.0, fabs(k1));
            for (int g = 0; g < count; ++g) {
                const double a = msgs[g * msg_len + 0], j = msgs[g * msg_len + 1];
                if (a <= bound && j < kj) { kj = j; win = g; }
            }
        }
        s_win = win;
        if (win < 0) {
            rec->outcome = DEV_NO_CANDIDATE;
            if (rule == 1) rec->last_selected = -1;
        } else {
            rec->q = (int)msgs[win * msg_len + 1];
            rec->d_q = msgs[win * msg_len + 2];
            if (rule == 1) rec->last_selected = rec->q;
        }
    }
    __syncthreads();
    const int win = s_win;
    if (win < 0) return;
    for (int i = threadIdx.x; i < m; i += kSingleBlock) aq[i] = msgs[win * msg_len + 3 + i];
}

// The winner among the gathered candidates, its tableau column, and the ratio test on it in one
// single-workgroup launch (tableau engine: the candidate's payload IS alpha).
__global__ __launch_bounds__(kSingleBlock) void k_select_candidate_ratio(const double* __restrict__ msgs, int count,
                                                                         int64_t msg_len, int m, double* alpha,
                                                                         const double* b, const int32_t* basis_indices,
                                                                         int rule, Tolerances tol, DeferredUpdate du,
                                                                         PivotRecord* rec) {
    const int outcome = rec->outcome, p = rec->n_eta;
    if (outcome != DEV_RUNNING) return;
    __shared__ int s_win;
    if (threadIdx.x == 0) {
        int win = -1; double k1 = INFINITY; double kj = 0.0;
        for (int g = 0; g < count; ++g) {
            const double a = msgs[g * msg_len + 0], j = msgs[g * msg_len + 1];
            if (a < k1 || (a == k1 && win >= 0 && j < kj)) { k1 = a; kj = j; win = g; }
        }
        if (win >= 0 && rule == 2 && tol.tie > 0.0) {
            const double bound = k1 + tol.tie * fmax(1.0, fabs(k1));
            for (int g = 0; g < count; ++g) {
                const double a = msgs[g * msg_len + 0], j = msgs[g * msg_len + 1];
                if (a <= bound && j < kj) { kj = j; win = g; }
            }
        }
        s_win = win;
        if (win < 0) {
            rec->outcome = DEV_NO_CANDIDATE;
            if (rule == 1) rec->last_selected = -1;
        } else {
            rec->q = (int)msgs[win * msg_len + 1];
            rec->d_q = msgs[win * msg_len + 2];
            if (rule == 1) rec->last_selected = rec->q;
        }
    }
    __syncthreads();
    const int win = s_win;
    if (win < 0) return;
    for (int i = threadIdx.x; i < m; i += kSingleBlock) alpha[i] = msgs[win * msg_len + 3 + i];
    __syncthreads();                                   // alpha was written by this workgroup: visible to it
    ratio_body<kSingleBlock, 16>(alpha, b, basis_indices, m, tol, du, p, rec);
}

__global__ void k_gather_alpha(const double* __restrict__ slices, int count, int stride, int m,
                               double* __restrict__ alpha, const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int g = i / stride;
    alpha[i] = (g < count) ? slices[(int64_t)g * stride + (i - g * stride)] : 0.0;
}

__global__ void k_pad_slice(double* __restrict__ slice, int valid, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= valid && i < stride) slice[i] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

void launch_price_structural(const double* A, int64_t ld_a, const ColumnTable& ct, const double* minus_pi,
                             double* d, int32_t p_lo, int32_t p_hi, int32_t cost_mode, const PivotRecord* rec,
                             hipStream_t s) {
    if (p_hi <= p_lo) return;
    const int blocks = cdiv(p_hi - p_lo, kVecPerBlock);
    hipLaunchKernelGGL(k_price_structural, dim3(blocks), dim3(kThreads), 0, s, A, ld_a, ct, minus_pi, d, p_lo,
                       p_hi, cost_mode, SelectPartials{}, rec);
}

void launch_price_mask_unowned(const ColumnTable& ct, double* d, int32_t p_lo, int32_t p_hi,
                               const PivotRecord* rec, hipStream_t s) {
    if (ct.nr_normal <= 0) return;
    hipLaunchKernelGGL(k_price_mask_unowned, dim3(cdiv(ct.nr_normal, 256)), dim3(256), 0, s, ct, d, p_lo, p_hi,
                       rec);
}

void launch_price_virtual(const ColumnTable& ct, const double* minus_pi, double* d, int32_t cost_mode,
                          const PivotRecord* rec, hipStream_t s) {
    const int n = ct.nr_artificial + ct.nr_virtual;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_price_virtual, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, s, ct, minus_pi, d, cost_mode,
                       SelectPartials{}, rec);
}

int32_t price_structural_blocks(int32_t p_lo, int32_t p_hi) { return p_hi > p_lo ? cdiv(p_hi - p_lo, kVecPerBlock) : 0; }
int32_t price_virtual_blocks(const ColumnTable& ct) {
    const int n = ct.nr_artificial + ct.nr_virtual;
    return n > 0 ? cdiv(n, kThreads) : 0;
}

void launch_price_structural_sel(const double* A, int64_t ld_a, const ColumnTable& ct, const double* minus_pi,
                                 double* d, int32_t p_lo, int32_t p_hi, int32_t cost_mode, SelectPartials sp,
                                 const PivotRecord* rec, hipStream_t s) {
    const int blocks = price_structural_blocks(p_lo, p_hi);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_price_structural, dim3(blocks), dim3(kThreads), 0, s, A, ld_a, ct, minus_pi, d, p_lo,
                       p_hi, cost_mode, sp, rec);
}

void launch_price_virtual_sel(const ColumnTable& ct, const double* minus_pi, double* d, int32_t cost_mode,
                              SelectPartials sp, const PivotRecord* rec, hipStream_t s) {
    const int blocks = price_virtual_blocks(ct);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_price_virtual, dim3(blocks), dim3(kThreads), 0, s, ct, minus_pi, d, cost_mode, sp, rec);
}

void launch_select_partials(SelectPartials sp, int32_t count, const double* d, const double* A, int64_t ld_a,
                            const ColumnTable& ct, int32_t m, double* aq, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_select_partials, dim3(1), dim3(kSingleBlock), 0, s, sp, count, d, A, ld_a, DeviceCSC{}, ct, m, aq,
                       rec);
}

void launch_ratio_eta(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m, Tolerances tol,
                      const DeferredUpdate& du, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_ratio, dim3(1), dim3(kSingleBlock), 0, s, alpha, b, basis_indices, m, tol, du, rec);
}

void launch_select_column(const double* d, const uint8_t* in_basis, int32_t n, int32_t rule, double tol_cost,
                          double tol_tie, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_select_column, dim3(1), dim3(kSingleBlock), 0, s, d, in_basis, n, rule, tol_cost, tol_tie,
                       rec);
}

void launch_build_column(const double* A, int64_t ld_a, const ColumnTable& ct, int32_t m, double* aq,
                         const PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_build_column, dim3(cdiv(m, 256)), dim3(256), 0, s, A, ld_a, ct, m, aq, rec);
}

void launch_ftran(const double* Binv, int64_t ld_b, int32_t m, int32_t row_lo, int32_t row_hi,
                  const double* aq, double* out, int32_t out_offset, const PivotRecord* rec, hipStream_t s) {
    if (row_hi <= row_lo) return;
    const int blocks = cdiv(row_hi - row_lo, kVecPerBlock);
    hipLaunchKernelGGL(k_ftran, dim3(blocks), dim3(kThreads), 0, s, Binv, ld_b, m, row_lo, row_hi, aq, out,
                       out_offset, rec);
}

void launch_ratio(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m,
                  Tolerances tol, PivotRecord* rec, hipStream_t s) {
    DeferredUpdate none{};
    hipLaunchKernelGGL(k_ratio, dim3(1), dim3(kSingleBlock), 0, s, alpha, b, basis_indices, m, tol, none, rec);
}

void launch_compute_rho(const double* Binv, int64_t ld_b, int32_t m, int32_t row_lo, int32_t row_hi,
                        double* rho, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_compute_rho, dim3(cdiv(ld_b, 256)), dim3(256), 0, s, Binv, ld_b, m, row_lo, row_hi, rho,
                       rec);
}

void launch_update_vectors(int32_t m, const double* alpha, const double* rho, double* b, double* minus_pi,
                           int32_t* basis_indices, uint8_t* in_basis, int32_t* trace, int64_t trace_cap,
                           PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_update_vectors, dim3(cdiv(m, 256)), dim3(256), 0, s, m, alpha, rho, b, minus_pi,
                       basis_indices, in_basis, trace, trace_cap, rec);
}

void launch_update_inverse(double* Binv, int64_t ld_b, int32_t m, int32_t row_lo, int32_t row_hi,
                           const double* alpha, const double* rho, const PivotRecord* rec, hipStream_t s) {
    if (row_hi <= row_lo) return;
    dim3 grid(cdiv(ld_b, 2 * kThreads), cdiv(row_hi - row_lo, kUpdRowsPerBlock));
    hipLaunchKernelGGL(k_update_inverse, grid, dim3(kThreads), 0, s, Binv, ld_b, m, row_lo, row_hi, alpha, rho,
                       rec);
}

void launch_weighted_column_sums(const double* Binv, int64_t ld_b, int32_t m, const double* w,
                                 double* minus_pi, hipStream_t s) {
    hipLaunchKernelGGL(k_weighted_column_sums, dim3(cdiv(m, 128)), dim3(128), 0, s, Binv, ld_b, m, w, minus_pi);
}

void launch_set_identity(double* Binv, int64_t ld_b, int32_t row_lo, int32_t row_hi, hipStream_t s) {
    const int64_t total = (int64_t)(row_hi - row_lo) * ld_b;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_set_identity, dim3(cdiv(total, 256)), dim3(256), 0, s, Binv, ld_b, row_lo, row_hi);
}

void launch_fill_dense(double* A, int64_t ld, int32_t m, int32_t n, uint64_t seed, int64_t first_column,
                       hipStream_t s) {
    const int64_t total = (int64_t)m * n;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_fill_dense, dim3(cdiv(total, 256)), dim3(256), 0, s, A, ld, m, n, seed, first_column);
}

void launch_apply_w(const DeferredUpdate& du, int32_t m, const double* v, double* alpha, const PivotRecord* rec,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_apply_w, dim3(cdiv(m, kThreads)), dim3(kThreads), 0, s, du, m, v, alpha, rec);
}

void launch_eta_prepare(const DeferredUpdate& du, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_eta_prepare, dim3(1), dim3(128), 0, s, du, rec);
}

void launch_update_w(const DeferredUpdate& du, int32_t m, const double* alpha, const PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_update_w, dim3(cdiv(m, kThreads)), dim3(kThreads), 0, s, du, m, alpha, rec);
}

void launch_rho_deferred(const DeferredUpdate& du, const double* Binv, int64_t ld_b, int32_t m, int32_t row_lo,
                         int32_t row_hi, double* rho, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_rho_deferred, dim3(cdiv(ld_b, 2 * kThreads)), dim3(kThreads), 0, s, du, Binv, ld_b, m, row_lo,
                       row_hi, rho, rec);
}

void launch_flush_snapshot(const DeferredUpdate& du, const double* Binv, int64_t ld_b, int32_t row_lo, int32_t row_hi,
                           const PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_flush_snapshot, dim3(cdiv(ld_b, 256), du.kmax), dim3(256), 0, s, du, Binv, ld_b, row_lo, row_hi,
                       rec);
}

void launch_flush_apply(const DeferredUpdate& du, double* Binv, int64_t ld_b, int32_t m, int32_t row_lo,
                        int32_t row_hi, const PivotRecord* rec, hipStream_t s) {
    if (row_hi <= row_lo) return;
    dim3 grid(cdiv(ld_b, kFT), cdiv(row_hi - row_lo, kFT));
    hipLaunchKernelGGL(k_flush_apply, grid, dim3(kThreads), 0, s, du, Binv, ld_b, m, row_lo, row_hi, rec);
}

void launch_flush_reset(const DeferredUpdate& du, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_flush_reset, dim3(1), dim3(128), 0, s, du, rec);
}


int32_t tab_scan_blocks(int32_t n_owned_columns) { return cdiv(n_owned_columns, kThreads); }

void launch_tab_build(const TableauView& tv, const double* A, int64_t ld_a, const ColumnTable& ct, hipStream_t s) {
    const int64_t total = (int64_t)tv.m * (tv.c_hi - tv.c_lo);
    if (total <= 0) return;
    hipLaunchKernelGGL(k_tab_build, dim3(cdiv(total, 256)), dim3(256), 0, s, tv, A, ld_a, ct);
}

void launch_tab_price_init(const TableauView& tv, const double* w, const double* cost_store, hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    hipLaunchKernelGGL(k_tab_price_init, dim3(cdiv(tv.c_hi - tv.c_lo, kVecPerBlock)), dim3(kThreads), 0, s, tv, w,
                       cost_store);
}

void launch_tab_scan(const TableauView& tv, SelectPartials sp, const PivotRecord* rec, hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    hipLaunchKernelGGL(k_tab_scan, dim3(tab_scan_blocks(tv.c_hi - tv.c_lo)), dim3(kThreads), 0, s, tv, sp, rec);
}

void launch_tab_select(const TableauView& tv, SelectPartials sp, int32_t count, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_select, dim3(1), dim3(kSingleBlock), 0, s, tv, sp, count, rec);
}

void launch_tab_column(const TableauView& tv, const DeferredUpdate& du, double* alpha, const PivotRecord* rec,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_tab_column, dim3(cdiv(tv.m, kThreads)), dim3(kThreads), 0, s, tv, du, alpha, rec);
}

void launch_tab_row_update(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, PivotRecord* rec,
                           hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    hipLaunchKernelGGL(k_tab_row_update, dim3(tab_scan_blocks(tv.c_hi - tv.c_lo)), dim3(kThreads), 0, s, tv, du, sp,
                       rec);
}

void launch_tab_update_vectors(int32_t m, const double* alpha, double* b, int32_t* basis_indices, uint8_t* in_basis,
                               int32_t* trace, int64_t trace_cap, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_update_vectors, dim3(cdiv(m, 256)), dim3(256), 0, s, m, alpha, b, basis_indices, in_basis,
                       trace, trace_cap, rec);
}

void launch_tab_update_w_vectors(const DeferredUpdate& du, int32_t m, const double* alpha, double* b,
                                 int32_t* basis_indices, uint8_t* in_basis, int32_t* trace, int64_t trace_cap,
                                 PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_update_w_vectors, dim3(cdiv(m, kThreads)), dim3(kThreads), 0, s, du, m, alpha, b,
                       basis_indices, in_basis, trace, trace_cap, rec);
}

void launch_tab_select_column(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t count,
                              double* alpha, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_select_column, dim3(cdiv(tv.m, kThreads)), dim3(kThreads), 0, s, tv, du, sp, count, alpha,
                       (double*)nullptr, rec);
}

void launch_tab_select_column_msg(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t count,
                                  double* msg, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_select_column, dim3(cdiv(tv.m, kThreads)), dim3(kThreads), 0, s, tv, du, sp, count, msg + 3,
                       msg, rec);
}

void launch_tab_update_all(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t m,
                           const double* alpha, double* b, int32_t* basis_indices, uint8_t* in_basis, int32_t* trace,
                           int64_t trace_cap, PivotRecord* rec, hipStream_t s) {
    const int nb_row = tv.c_hi > tv.c_lo ? tab_scan_blocks(tv.c_hi - tv.c_lo) : 0;
    const int nb_w = cdiv(m, kThreads);
    hipLaunchKernelGGL(k_tab_update_all, dim3(nb_row + nb_w), dim3(kThreads), 0, s, tv, du, sp, nb_row, m, alpha, b,
                       basis_indices, in_basis, trace, trace_cap, rec);
}

void launch_tab_flush(const TableauView& tv, const DeferredUpdate& du, const PivotRecord* rec, hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    const int ncols = tv.c_hi - tv.c_lo;
    if ((int64_t)ncols * tv.m >= (1 << 16)) {
        // LDS-staged operands: 8 wavefronts, 128 columns x 128 rows per workgroup, chunks of 16 pivots
        // (64 KB of LDS, <= 128 VGPRs: two workgroups per CU, so one streams its T0 tile while the
        // other one is in its MFMA loop)
        constexpr int WC = 2, WR = 4, KC = 16;
        dim3 grid(cdiv(tv.m, WR * 32), cdiv(ncols, WC * 64));
        hipLaunchKernelGGL((k_tab_flush_lds<WC, WR, KC>), grid, dim3(WC * WR * 64), 0, s, tv, du, rec);
        return;
    }
    constexpr int MT = 4, NT = 2;      // wavefront tile 64 columns x 32 rows, workgroup 128 x 64
    dim3 grid(cdiv(ncols, 2 * 16 * MT), cdiv(tv.m, 2 * 16 * NT));
    hipLaunchKernelGGL((k_tab_flush<MT, NT>), grid, dim3(kThreads), 0, s, tv, du, rec);
}

void launch_tab_gather_columns(const TableauView& tv, const int32_t* cols, double* out, hipStream_t s) {
    const int64_t total = (int64_t)tv.m * tv.m;
    hipLaunchKernelGGL(k_tab_gather_columns, dim3(cdiv(total, 256)), dim3(256), 0, s, tv, cols, out);
}

void launch_tab_row(const TableauView& tv, const DeferredUpdate& du, int32_t row, double* out, const PivotRecord* rec,
                    hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    hipLaunchKernelGGL(k_tab_row, dim3(tab_scan_blocks(tv.c_hi - tv.c_lo)), dim3(kThreads), 0, s, tv, du, row, out, rec);
}

int32_t price_csc_blocks(int32_t p_lo, int32_t p_hi) { return p_hi > p_lo ? cdiv(p_hi - p_lo, kThreads) : 0; }

void launch_price_csc(const DeviceCSC& csc, const ColumnTable& ct, const double* vec, double* d, int32_t p_lo,
                      int32_t p_hi, int32_t cost_mode, SelectPartials sp, const PivotRecord* rec, hipStream_t s) {
    const int blocks = price_csc_blocks(p_lo, p_hi);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_price_csc, dim3(blocks), dim3(kThreads), 0, s, csc, ct, vec, d, p_lo, p_hi, cost_mode, sp, rec);
}

void launch_select_partials_csc(SelectPartials sp, int32_t count, const double* d, const DeviceCSC& csc,
                                const ColumnTable& ct, int32_t m, double* aq, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_select_partials, dim3(1), dim3(kSingleBlock), 0, s, sp, count, d, (const double*)nullptr,
                       (int64_t)0, csc, ct, m, aq, rec);
}

void launch_build_column_csc(const DeviceCSC& csc, const ColumnTable& ct, int32_t m, double* aq, const PivotRecord* rec,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_build_column_csc, dim3(1), dim3(kSingleBlock), 0, s, csc, ct, m, aq, rec);
}

// LDS plan of one solve kernel: x first, then whichever schedule fits behind it
struct LuLdsPlan { int x_in_lds, stage_first, stage_second; size_t bytes; };
static LuLdsPlan plan_lu_lds(int m, const DeviceSchedule& a, const DeviceSchedule& b) {
    LuLdsPlan p{0, 0, 0, 0};
    int64_t used = 0;
    if (lu_up16((int64_t)m * 8) <= kLuLdsBytes / 2) { p.x_in_lds = 1; used = lu_up16((int64_t)m * 8); }
    const int64_t na = schedule_lds_bytes(m, a.nnz, a.n_levels), nb = schedule_lds_bytes(m, b.nnz, b.n_levels);
    int64_t extra = 0;
    if (p.x_in_lds && used + na <= kLuLdsBytes) { p.stage_first = 1; extra = na; }
    if (p.x_in_lds && used + nb <= kLuLdsBytes) { p.stage_second = 1; extra = std::max(extra, nb); }
    p.bytes = (size_t)(used + extra);
    return p;
}

template <class K>
static K pick_lu_variant(const LuLdsPlan& p, K v000, K v100, K v110, K v101, K v111) {
    if (!p.x_in_lds) return v000;
    if (p.stage_first && p.stage_second) return v111;
    if (p.stage_first) return v110;
    if (p.stage_second) return v101;
    return v100;
}

static void allow_big_lds(const void* fn) {
    static std::vector<const void*> done;
    for (auto f : done) if (f == fn) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLuLdsBytes);
    done.push_back(fn);
}

void launch_lu_ftran(const DeviceLU& lu, const double* aq, double* v, double* scratch, const PivotRecord* rec,
                     hipStream_t s) {
    const LuLdsPlan p = plan_lu_lds(lu.m, lu.Lf, lu.Uf);
    auto fn = pick_lu_variant(p, k_lu_ftran<false, false, false>, k_lu_ftran<true, false, false>,
                              k_lu_ftran<true, true, false>, k_lu_ftran<true, false, true>, k_lu_ftran<true, true, true>);
    allow_big_lds(reinterpret_cast<const void*>(fn));
    hipLaunchKernelGGL(fn, dim3(1), dim3(kLuThreads), p.bytes, s, lu, aq, v, scratch, rec);
}

void launch_lu_btran(const DeviceLU& lu, const DeferredUpdate& du, const double* rhs, int32_t row, double* rho,
                     double* scratch, const PivotRecord* rec, hipStream_t s) {
    const LuLdsPlan p = plan_lu_lds(lu.m, lu.Ub, lu.Lb);
    auto fn = pick_lu_variant(p, k_lu_btran<false, false, false>, k_lu_btran<true, false, false>,
                              k_lu_btran<true, true, false>, k_lu_btran<true, false, true>, k_lu_btran<true, true, true>);
    allow_big_lds(reinterpret_cast<const void*>(fn));
    hipLaunchKernelGGL(fn, dim3(1), dim3(kLuThreads), p.bytes, s, lu, du, rhs, row, rho, scratch, rec);
}

void launch_pack_candidate(const double* aq, int32_t m, double* msg, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_candidate, dim3(cdiv(m, 256)), dim3(256), 0, s, aq, m, msg, rec);
    hipLaunchKernelGGL(k_clear_no_candidate, dim3(1), dim3(1), 0, s, rec);
}

void launch_select_candidate_ratio(const double* msgs, int32_t count, int64_t msg_len, int32_t m, double* alpha,
                                   const double* b, const int32_t* basis_indices, int32_t rule, Tolerances tol,
                                   const DeferredUpdate& du, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_select_candidate_ratio, dim3(1), dim3(kSingleBlock), 0, s, msgs, count, msg_len, m, alpha, b,
                       basis_indices, rule, tol, du, rec);
}

void launch_select_candidate(const double* msgs, int32_t count, int64_t msg_len, int32_t m, double* aq,
                             int32_t rule, double tol_tie, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_select_candidate, dim3(1), dim3(kSingleBlock), 0, s, msgs, count, msg_len, m, aq, rule,
                       tol_tie, rec);
}

void launch_gather_alpha(const double* slices, int32_t count, int32_t stride, int32_t m, double* alpha,
                         const PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_alpha, dim3(cdiv(m, 256)), dim3(256), 0, s, slices, count, stride, m, alpha, rec);
}

void launch_pad_slice(double* slice, int32_t valid, int32_t stride, hipStream_t s) {
    if (stride <= valid) return;
    hipLaunchKernelGGL(k_pad_slice, dim3(cdiv(stride, 256)), dim3(256), 0, s, slice, valid, stride);
}

}  // namespace relp
