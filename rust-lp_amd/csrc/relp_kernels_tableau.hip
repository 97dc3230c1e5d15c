// relp_kernels_tableau.hip -- kernels of the dense-tableau engine (RELP_ENGINE_TABLEAU), including the
// f64-MFMA flush T0 += W R0.
#include "relp_device_common.h"

namespace relp {

// ------------------------------------------------------------------------------------------------
// Dense-tableau engine:  T = (I + W S') T0  (see TableauView / DeferredUpdate in relp_kernels.h)
//   PRICE  = one row of T per pivot:   d <- d - (d_q / alpha_r) T[r,:]      (instead of 8 m n_s bytes)
//   FTRAN  = one column of T per pivot: alpha = T0[:,q] + W R0[:,q]        (instead of 8 m^2 bytes)
//   UPDATE = W <- E W per pivot; T0 += W R0 once per K pivots on the f64 matrix cores
// ------------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_tab_build(TableauView tv, const double* __restrict__ A, int64_t ld_a, ColumnTable ct) {
    const int64_t total = (int64_t)tv.m * (tv.c_hi - tv.c_lo);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
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

// w[i] = cost of the variable that is basic in row i (current phase), for re-pricing d from T0
__global__ void k_tab_basis_costs(const int32_t* __restrict__ basis_indices, const double* __restrict__ cost_store,
                                  int col_off, int n_store, int m, double* __restrict__ w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int j = basis_indices[i];
    const int c = j + col_off;
    w[i] = (j < kWrappedArtificialBase && c >= 0 && c < n_store) ? cost_store[c] : 0.0;
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
// `wr` (shared memory) = row r of W before this pivot; the pivot itself as plain arguments (the fused launch computes them
// per workgroup, the others take them from the record).  `R` is only read for the selection key's rule memory.
__device__ __forceinline__ void tab_row_update_core(const TableauView& tv, const DeferredUpdate& du, const SelectPartials& sp,
                                                    const PivotRecord& R, int block, const double* s_wr, int p_old, int jt, int r,
                                                    int q, int leaving, double d_q, double alpha_r) {
    const PivotRecord* rec = &R;
    const int c = tv.c_lo + block * kThreads + threadIdx.x;
    const double d_old = c < tv.c_hi ? tv.d[c] : 0.0;
    __syncthreads();
    double key = INFINITY;
    int kj = 0x7fffffff;
    if (c < tv.c_hi) {
        double base;
        if (jt < p_old) base = tv.R0[(int64_t)jt * tv.ld_r + c];
        else {
            // row r is new in this block: its row of the tableau the block started from
            base = tv.T0[(int64_t)c * tv.ld_t + r];
            tv.R0[(int64_t)jt * tv.ld_r + c] = base;
        }
        double row = base;
        for (int j = 0; j < p_old; ++j) row = fma(s_wr[j], tv.R0[(int64_t)j * tv.ld_r + c], row);
        const double theta = d_q / alpha_r;
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

__device__ __forceinline__ void tab_row_update_body(const TableauView& tv, const DeferredUpdate& du,
                                                    const SelectPartials& sp, const PivotRecord& R, int block) {
    __shared__ double s_wr[kMaxEta];
    // fetched without waiting for p_old (entries beyond it are never used)
    if ((int)threadIdx.x < du.kmax) s_wr[threadIdx.x] = du.wr[threadIdx.x];
    tab_row_update_core(tv, du, sp, R, block, s_wr, R.n_eta_old, R.eta_target, R.r, R.q, R.leaving, R.d_q, R.alpha_r);
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
        if (br == 0.0) rec->degenerate += 1;            // ratio 0: the basis changes, the vertex does not
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

// Ratio test + both halves of the update in ONE launch (relp_kernels.h: launch_tab_ratio_update_all).  Workgroups
// [0, nb_row): tableau row / reduced costs / PRICE partials; [nb_row, ..): W, b, basis, bookkeeping.  Every workgroup repeats
// the ratio test (ratio_blocks_pick: same inputs, same code, same row) and fetches row r of W, alpha_r, b_r and the slot of
// row r itself.  Nothing a workgroup reads is rewritten by another one in this launch: b and the basis array go in -> out,
// the new row r of W goes to `shadow`, n_eta is read as p_now; the basis flags are read through the expression that is
// the same for old and new flags; pos_of_row[r] reads as -1 or as the value it is about to get.  Same arithmetic as
// k_ratio_blocks + k_tab_update_all, bit for bit.
__global__ __launch_bounds__(kThreads) void k_tab_ratio_update_all(TableauView tv, DeferredUpdate du, SelectPartials sp, int nb_row,
                                                                   int m, const double* alpha,
                                                                   const double* __restrict__ b_in, double* __restrict__ b_out,
                                                                   const int32_t* __restrict__ basis_in,
                                                                   int32_t* __restrict__ basis_out, uint8_t* in_basis,
                                                                   int32_t* __restrict__ trace, int64_t trace_cap, Tolerances tol,
                                                                   const double* rmin, int nblk,
                                                                   double* __restrict__ shadow, int32_t* __restrict__ shadow_meta,
                                                                   PivotRecord* rec, const double* __restrict__ msgs, int count,
                                                                   int64_t msg_len, int rule) {
    const double first = (!msgs && (int)threadIdx.x < nblk) ? rmin[threadIdx.x] : INFINITY;      // in flight with the record
    PivotRecord R = *rec;
    const bool w_half = (int)blockIdx.x >= nb_row;
    const int wblock = (int)blockIdx.x - nb_row;
    const int i = wblock * kThreads + threadIdx.x;
    // the loop has ended (or ends here): the double buffers still advance, because the host keeps swapping them
    const bool mine = w_half && i < m;
    if (R.outcome != DEV_RUNNING) {
        if (mine) { b_out[i] = b_in[i]; basis_out[i] = basis_in[i]; }
        return;
    }
    if (msgs) {
        // sharded loop: the entering column is the winner among the gathered candidates [key, j, d_j, alpha (m), block
        // minima of the ratios]; every workgroup picks it with the rules of k_tab_select_candidate_ratio
        constexpr int kMaxRanks = 64;
        __shared__ double s_key[kMaxRanks], s_idx[kMaxRanks], s_dq[kMaxRanks];
        __shared__ int s_win;
        for (int g = threadIdx.x; g < count && g < kMaxRanks; g += kThreads) {
            s_key[g] = msgs[g * msg_len + 0]; s_idx[g] = msgs[g * msg_len + 1]; s_dq[g] = msgs[g * msg_len + 2];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int win = -1; double k1 = INFINITY; double kj = 0.0;
            for (int g = 0; g < count; ++g) {
                const double a = s_key[g], j = s_idx[g];
                if (a < k1 || (a == k1 && win >= 0 && j < kj)) { k1 = a; kj = j; win = g; }
            }
            if (win >= 0 && rule == 2 && tol.tie > 0.0) {
                const double bound = k1 + tol.tie * fmax(1.0, fabs(k1));
                for (int g = 0; g < count; ++g)
                    if (s_key[g] <= bound && s_idx[g] < kj) { kj = s_idx[g]; win = g; }
            }
            s_win = win;
        }
        __syncthreads();
        const int win = s_win;
        if (win < 0) {
            if (mine) { b_out[i] = b_in[i]; basis_out[i] = basis_in[i]; }
            if (wblock == 0 && threadIdx.x == 0) {
                rec->outcome = DEV_NO_CANDIDATE;
                if (rule == 1) rec->last_selected = -1;
            }
            return;
        }
        R.q = (int)s_idx[win];
        R.d_q = s_dq[win];
        if (rule == 1) R.last_selected = R.q;                      // (what the row update's selection keys start from)
        alpha = msgs + win * msg_len + 3;
        rmin = alpha + m;
    }
    int r, leaving;
    ratio_blocks_pick<kThreads>(alpha, b_in, basis_in, m, tol, rmin, nblk, &r, &leaving, first, !msgs);
    if (r < 0) {
        if (mine) { b_out[i] = b_in[i]; basis_out[i] = basis_in[i]; }
        if (wblock == 0 && threadIdx.x == 0) rec->outcome = DEV_NO_ROW;
        return;
    }
    __shared__ double s_wr[kMaxEta];
    __shared__ double s_ab[2];
    __shared__ int s_jt;
    const int p_old = R.p_now;
    if ((int)threadIdx.x < p_old) s_wr[threadIdx.x] = du.W[(int64_t)threadIdx.x * du.ld + r];
    if (threadIdx.x == 0) {
        s_ab[0] = alpha[r]; s_ab[1] = b_in[r];
        const int slot = du.pos_of_row[r];
        s_jt = slot < 0 ? p_old : slot;
    }
    __syncthreads();
    const double alpha_r = s_ab[0], b_r = s_ab[1];
    const int jt = s_jt, q = R.q;
    if (!w_half) {
        tab_row_update_core(tv, du, sp, R, blockIdx.x, s_wr, p_old, jt, r, q, leaving, R.d_q, alpha_r);
        return;
    }
    const double br = b_r / alpha_r;
    if (i < m) {
        const double a = alpha[i], b_i = b_in[i];
        const double u = (i == r) ? (1.0 / alpha_r - 1.0) : (-a / alpha_r);
        if (i != r) {
            if (u != 0.0) {
                for (int j = 0; j < p_old; ++j) {
                    const double w = s_wr[j];
                    if (w != 0.0) du.W[(int64_t)j * du.ld + i] = fma(u, w, du.W[(int64_t)j * du.ld + i]);
                }
            }
            double* tgt = du.W + (int64_t)jt * du.ld + i;
            if (jt < p_old) *tgt += u; else *tgt = u;
        } else {
            // row r itself: other workgroups are reading its old values right now, the new ones go to the shadow row
            for (int j = 0; j < p_old; ++j) {
                const double w = s_wr[j];
                shadow[j] = (u != 0.0 && w != 0.0) ? fma(u, w, w) : w;
            }
            if (jt < p_old) shadow[jt] += u; else shadow[jt] = u;
            shadow_meta[0] = r;
            shadow_meta[1] = jt < p_old ? p_old : p_old + 1;
        }
        b_out[i] = (i == r) ? br : (a != 0.0 ? fma(-a, br, b_i) : b_i);
        basis_out[i] = (i == r) ? q : basis_in[i];
    }
    if (i == 0) {
        if (msgs) { rec->q = q; rec->d_q = R.d_q; if (rule == 1) rec->last_selected = q; }
        rec->r = r; rec->leaving = leaving; rec->alpha_r = alpha_r; rec->b_r = b_r;
        rec->n_eta_old = p_old; rec->eta_target = jt;
        if (jt >= p_old) { du.S[p_old] = r; du.pos_of_row[r] = p_old; rec->n_eta = p_old + 1; }
        rec->minus_objective = fma(-R.d_q, br, R.minus_objective);
        if (leaving < kWrappedArtificialBase) in_basis[leaving] = 0;   // a wrapped artificial has no flag
        in_basis[q] = 1;
        const long long it = R.iterations;
        if (trace && it < trace_cap) {
            trace[0 * trace_cap + it] = R.phase;
            trace[1 * trace_cap + it] = q;
            trace[2 * trace_cap + it] = r;
            trace[3 * trace_cap + it] = leaving;
        }
        if (br == 0.0) rec->degenerate += 1;
        rec->iterations = it + 1;
    }
}

// Folds a pending shadow row (k_tab_ratio_update_all) into W; {row, length} = {-1, 0} afterwards.
__global__ void k_tab_apply_shadow(DeferredUpdate du, double* __restrict__ shadow, int32_t* __restrict__ shadow_meta) {
    const int row = shadow_meta[0], len = shadow_meta[1];
    if (row < 0) return;
    for (int j = threadIdx.x; j < len; j += blockDim.x) du.W[(int64_t)j * du.ld + row] = shadow[j];
    __syncthreads();
    if (threadIdx.x == 0) shadow_meta[0] = -1;
}

// PRICE's final reduction and the tableau column in one launch: every workgroup reduces the (few)
// partials to the same entering column q, then forms alpha = T0[:,q] + W R0[:,q] for its rows.
// `msg` (sharded engines): the candidate message [key, j, d_j, alpha(m)] of this rank is written instead
// of the record; a rank without a candidate sends key = +inf and stays RUNNING (another rank may have one).
// `rmin` (single-GPU loop): the minimum ratio b_i / alpha_i over this workgroup's 256 rows, for
// k_ratio_blocks.
__global__ __launch_bounds__(kThreads) void k_tab_select_column(TableauView tv, DeferredUpdate du, SelectPartials sp,
                                                                int count, double* alpha, double* msg,
                                                                const double* b, Tolerances tol,
                                                                double* rmin, PivotRecord* rec,
                                                                const double* shadow = nullptr, int32_t* shadow_meta = nullptr) {
    const int outcome = rec->outcome, p = rec->n_eta;          // one round trip for both ...
    // ... and for the shadow row of the previous pivot's fused update: the workgroup that owns the row folds it into W before
    // it reads the row (no other workgroup reads it)
    if (shadow_meta) {
        const int srow = shadow_meta[0], slen = shadow_meta[1];
        if (srow >= 0 && srow / kThreads == (int)blockIdx.x) {
            if ((int)threadIdx.x < slen) du.W[(int64_t)threadIdx.x * du.ld + srow] = shadow[threadIdx.x];
            __syncthreads();
            if (threadIdx.x == 0) shadow_meta[0] = -1;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) rec->p_now = p;
    }
    // ... and for the first PRICE partial and b of this thread's row, which do not depend on the record
    double k1 = INFINITY;
    int bj = 0x7fffffff;
    if ((int)threadIdx.x < count) { k1 = sp.k1[threadIdx.x]; bj = sp.j[threadIdx.x]; }
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const double b_i = (rmin && i < tv.m) ? b[i] : 0.0;
    if (outcome != DEV_RUNNING) return;
    __shared__ double s_k1[kThreads / 64];
    __shared__ int s_j[kThreads / 64];
    __shared__ double s_vs[kMaxEta];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x + kThreads; t < count; t += kThreads) {
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
        // the slots whose own minimum is inside the band (normally one or two) are listed first, so the
        // scan does not walk all `count` slots one dependent load after the other
        constexpr int kListMax = 32;
        __shared__ int s_list[kListMax];
        __shared__ int s_cnt;
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < count; t += kThreads) {
            if (!(sp.k1[t] <= bound)) continue;
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < kListMax) s_list[pos] = t;
        }
        __syncthreads();
        const int listed = s_cnt;
        auto scan_slot = [&](int t) {
            const int c = tv.c_lo + t * kThreads + threadIdx.x;
            const int j = c - tv.col_off;
            if (c < tv.c_hi && j >= 0 && j < tv.n) {
                const double v = tv.d[c];
                if (!sp.in_basis[j] && v < -sp.tol_cost && v <= bound && j < lowest) lowest = j;
            }
        };
        if (listed <= kListMax) {
            for (int i = 0; i < listed; ++i) scan_slot(s_list[i]);
        } else {
            for (int t = 0; t < count; ++t)
                if (sp.k1[t] <= bound) scan_slot(t);
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
    const double t0 = i < tv.m ? tv.T0[(int64_t)cq * tv.ld_t + i] : 0.0;     // in flight together with the R0 column
    __syncthreads();
    double ratio = INFINITY;
    if (i < tv.m) {
        double a = t0;
        for (int j = 0; j < p; ++j) a = fma(du.W[(int64_t)j * du.ld + i], s_vs[j], a);
        alpha[i] = a;
        // the same expression as the ratio test's first pass (ratio_body), so min over the block minima is
        // bit for bit the minimum over all rows
        const double bz = b_i <= tol.zero ? 0.0 : b_i;
        if (a > tol.pivot) ratio = bz / a;
    }
    if (!rmin) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ratio = fmin(ratio, __shfl_down(ratio, off, 64));
    if (lane == 0) s_k1[wave] = ratio;
    __syncthreads();
    if (threadIdx.x == 0) rmin[blockIdx.x] = fmin(fmin(s_k1[0], s_k1[1]), fmin(s_k1[2], s_k1[3]));
}

// Ratio test from the per-block minima of k_tab_select_column (ratio_blocks_body): one workgroup.
__global__ __launch_bounds__(kSingleBlock) void k_ratio_blocks(const double* __restrict__ alpha, const double* __restrict__ b,
                                                               const int32_t* __restrict__ basis_indices, int m,
                                                               Tolerances tol, DeferredUpdate du,
                                                               const double* __restrict__ rmin, int nblk, PivotRecord* rec) {
    const int outcome = rec->outcome, p = rec->n_eta;
    const double first = (int)threadIdx.x < nblk ? rmin[threadIdx.x] : INFINITY;      // same round trip as the record
    if (outcome != DEV_RUNNING) return;
    ratio_blocks_body<kSingleBlock>(alpha, b, basis_indices, m, tol, du, rmin, nblk, p, rec, first, true);
}

// Sharded engines: the winner among the gathered candidates [key, j, d_j, alpha (m), block minima of the
// ratios], its tableau column copied for the update launch, and the ratio test on it from the block minima the
// sender computed -- one single-workgroup launch.  Same choice rules as k_select_candidate.
__global__ __launch_bounds__(kSingleBlock) void k_tab_select_candidate_ratio(const double* __restrict__ msgs, int count,
                                                                             int64_t msg_len, int m, double* __restrict__ alpha,
                                                                             const double* __restrict__ b,
                                                                             const int32_t* __restrict__ basis_indices,
                                                                             int rule, Tolerances tol, DeferredUpdate du,
                                                                             int forced_row, PivotRecord* rec) {
    const int outcome = rec->outcome, p = rec->n_eta;
    if (outcome != DEV_RUNNING) return;
    constexpr int kMaxRanks = 64;
    __shared__ double s_key[kMaxRanks], s_idx[kMaxRanks], s_dq[kMaxRanks];
    __shared__ int s_win;
    for (int g = threadIdx.x; g < count && g < kMaxRanks; g += kSingleBlock) {       // all heads in one round trip
        s_key[g] = msgs[g * msg_len + 0];
        s_idx[g] = msgs[g * msg_len + 1];
        s_dq[g] = msgs[g * msg_len + 2];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int win = -1; double k1 = INFINITY; double kj = 0.0;
        for (int g = 0; g < count; ++g) {
            const double a = s_key[g], j = s_idx[g];
            if (a < k1 || (a == k1 && win >= 0 && j < kj)) { k1 = a; kj = j; win = g; }
        }
        if (win >= 0 && rule == 2 && tol.tie > 0.0) {
            const double bound = k1 + tol.tie * fmax(1.0, fabs(k1));
            for (int g = 0; g < count; ++g)
                if (s_key[g] <= bound && s_idx[g] < kj) { kj = s_idx[g]; win = g; }
        }
        s_win = win;
        if (win < 0) {
            rec->outcome = DEV_NO_CANDIDATE;
            if (rule == 1) rec->last_selected = -1;
        } else {
            rec->q = (int)s_idx[win];
            rec->d_q = s_dq[win];
            if (rule == 1) rec->last_selected = (int)s_idx[win];
        }
    }
    __syncthreads();
    const int win = s_win;
    if (win < 0) return;
    const double* __restrict__ col = msgs + win * msg_len + 3;
    // the winner's block minima and its column (copied for the update launch) leave in the same round trip; the
    // copy is stored after the ratio test, which reads the column from the message itself
    const int nblk = (m + kThreads - 1) / kThreads;
    if (forced_row >= 0) {
        // zero-level pivot in a given row (phase_one.rs:246-250): no ratio test, the variable basic there leaves
        for (int i = threadIdx.x; i < m; i += kSingleBlock) alpha[i] = col[i];
        ratio_commit<kSingleBlock>(basis_indices[forced_row], forced_row, col, b, du, p, rec);
        return;
    }
    const double first = (int)threadIdx.x < nblk ? col[m + threadIdx.x] : INFINITY;
    constexpr int kCopy = 16;
    double cp[kCopy];
#pragma unroll
    for (int u = 0; u < kCopy; ++u) {
        const int i = threadIdx.x + u * kSingleBlock;
        cp[u] = i < m ? col[i] : 0.0;
    }
    ratio_blocks_body<kSingleBlock>(col, b, basis_indices, m, tol, du, col + m, nblk, p, rec, first, true);
#pragma unroll
    for (int u = 0; u < kCopy; ++u) {
        const int i = threadIdx.x + u * kSingleBlock;
        if (i < m) alpha[i] = cp[u];
    }
    for (int i = threadIdx.x + kCopy * kSingleBlock; i < m; i += kSingleBlock) alpha[i] = col[i];
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
        if (br == 0.0) rec->degenerate += 1;            // ratio 0: the basis changes, the vertex does not
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
__global__ __launch_bounds__(kThreads) void k_tab_flush(TableauView tv, DeferredUpdate du, const int32_t* p_dev) {
    constexpr int kFlushMT = MT, kFlushNT = NT;
    const int p = *p_dev;
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
__global__ __launch_bounds__(WC * WR * 64) void k_tab_flush_lds(TableauView tv, DeferredUpdate du, const int32_t* p_dev) {
    constexpr int MT = 4, NT = 2;                     // wavefront tile: 64 columns x 32 rows
    constexpr int TC = WC * 16 * MT, TR = WR * 16 * NT, NTHR = WC * WR * 64;
    constexpr int RA = KC * TC / 2 / NTHR, RB = KC * TR / 2 / NTHR;
    static_assert(RA * NTHR * 2 == KC * TC && RB * NTHR * 2 == KC * TR, "staging must divide evenly");
    const int p = *p_dev;
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
    const int64_t total = (int64_t)tv.m * tv.m;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx / tv.m), i = (int)(idx % tv.m);        // consecutive threads walk down a column of T0
        out[(int64_t)i * tv.m + k] = tv.T0[(int64_t)cols[k] * tv.ld_t + i];
    }
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

// phase_one.rs:236-244 for the sharded engine: among the owned columns, the candidates to replace a basic
// artificial variable at zero level in tableau row `row` -- non-basic, not artificial, reduced cost 0, tableau
// entry != 0 -- as PRICE partials with key = column index (the first one wins, like FirstProfitable).
__global__ __launch_bounds__(kThreads) void k_tab_zero_level_scan(TableauView tv, DeferredUpdate du, SelectPartials sp, int row,
                                                                  int nr_artificial, Tolerances tol, const PivotRecord* rec) {
    const int outcome = rec->outcome, p = rec->n_eta;
    if (outcome != DEV_RUNNING) return;
    __shared__ double s_w[kMaxEta];
    if ((int)threadIdx.x < p) s_w[threadIdx.x] = du.W[(int64_t)threadIdx.x * du.ld + row];
    __syncthreads();
    const int c = tv.c_lo + blockIdx.x * kThreads + threadIdx.x;
    const int j = c - tv.col_off;
    double key = INFINITY;
    int kj = 0x7fffffff;
    if (c < tv.c_hi && j >= nr_artificial && j < tv.n && !sp.in_basis[j] && fabs(tv.d[c]) <= tol.cost) {
        double v = tv.T0[(int64_t)c * tv.ld_t + row];
        for (int k = 0; k < p; ++k) v = fma(s_w[k], tv.R0[(int64_t)k * tv.ld_r + c], v);
        if (fabs(v) > tol.pivot) { key = (double)j; kj = j; }
    }
    block_partial_min(key, kj, sp, blockIdx.x);
}

int32_t tab_scan_blocks(int32_t n_owned_columns) { return cdiv(n_owned_columns, kThreads); }

void launch_tab_build(const TableauView& tv, const double* A, int64_t ld_a, const ColumnTable& ct, hipStream_t s) {
    const int64_t total = (int64_t)tv.m * (tv.c_hi - tv.c_lo);
    if (total <= 0) return;
    hipLaunchKernelGGL(k_tab_build, dim3(element_blocks(total)), dim3(256), 0, s, tv, A, ld_a, ct);
}

void launch_tab_price_init(const TableauView& tv, const double* w, const double* cost_store, hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    hipLaunchKernelGGL(k_tab_price_init, dim3(cdiv(tv.c_hi - tv.c_lo, kVecPerBlock)), dim3(kThreads), 0, s, tv, w,
                       cost_store);
}

void launch_tab_basis_costs(const TableauView& tv, const int32_t* basis_indices, const double* cost_store, double* w,
                            hipStream_t s) {
    hipLaunchKernelGGL(k_tab_basis_costs, dim3(cdiv(tv.m, 256)), dim3(256), 0, s, basis_indices, cost_store, tv.col_off,
                       tv.n_store, tv.m, w);
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
                       (double*)nullptr, (const double*)nullptr, Tolerances{}, (double*)nullptr, rec, (const double*)nullptr,
                       (int32_t*)nullptr);
}

void launch_tab_select_column_rmin(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t count,
                                   double* alpha, const double* b, Tolerances tol, double* rmin, PivotRecord* rec,
                                   hipStream_t s, const double* shadow, int32_t* shadow_meta) {
    hipLaunchKernelGGL(k_tab_select_column, dim3(cdiv(tv.m, kThreads)), dim3(kThreads), 0, s, tv, du, sp, count, alpha,
                       (double*)nullptr, b, tol, rmin, rec, shadow, shadow_meta);
}

void launch_tab_ratio_update_all(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t m,
                                 const double* alpha, const double* b_in, double* b_out, const int32_t* basis_in,
                                 int32_t* basis_out, uint8_t* in_basis, int32_t* trace, int64_t trace_cap, Tolerances tol,
                                 const double* rmin, double* shadow, int32_t* shadow_meta, PivotRecord* rec, hipStream_t s,
                                 const double* msgs, int32_t count, int64_t msg_len, int32_t rule) {
    const int nb_row = tv.c_hi > tv.c_lo ? tab_scan_blocks(tv.c_hi - tv.c_lo) : 0;
    const int nb_w = cdiv(m, kThreads);
    hipLaunchKernelGGL(k_tab_ratio_update_all, dim3(nb_row + nb_w), dim3(kThreads), 0, s, tv, du, sp, nb_row, m, alpha, b_in,
                       b_out, basis_in, basis_out, in_basis, trace, trace_cap, tol, rmin, nb_w, shadow, shadow_meta, rec, msgs,
                       (int)count, msg_len, (int)rule);
}

void launch_tab_apply_shadow(const DeferredUpdate& du, double* shadow, int32_t* shadow_meta, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_apply_shadow, dim3(1), dim3(128), 0, s, du, shadow, shadow_meta);
}

void launch_ratio_blocks(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m, Tolerances tol,
                         const DeferredUpdate& du, const double* rmin, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_ratio_blocks, dim3(1), dim3(kSingleBlock), 0, s, alpha, b, basis_indices, m, tol, du, rmin,
                       cdiv(m, kThreads), rec);
}

void launch_tab_select_column_msg(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t count,
                                  double* msg, const double* b, Tolerances tol, PivotRecord* rec, hipStream_t s,
                                  const double* shadow, int32_t* shadow_meta) {
    hipLaunchKernelGGL(k_tab_select_column, dim3(cdiv(tv.m, kThreads)), dim3(kThreads), 0, s, tv, du, sp, count, msg + 3,
                       msg, b, tol, msg + 3 + tv.m, rec, shadow, shadow_meta);
}

void launch_tab_select_candidate_ratio(const double* msgs, int32_t count, int64_t msg_len, int32_t m, double* alpha,
                                       const double* b, const int32_t* basis_indices, int32_t rule, Tolerances tol,
                                       const DeferredUpdate& du, int32_t forced_row, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_select_candidate_ratio, dim3(1), dim3(kSingleBlock), 0, s, msgs, count, msg_len, m, alpha, b,
                       basis_indices, rule, tol, du, forced_row, rec);
}

void launch_tab_zero_level_scan(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t row,
                                int32_t nr_artificial, Tolerances tol, const PivotRecord* rec, hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    hipLaunchKernelGGL(k_tab_zero_level_scan, dim3(tab_scan_blocks(tv.c_hi - tv.c_lo)), dim3(kThreads), 0, s, tv, du, sp, row,
                       nr_artificial, tol, rec);
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
    const int32_t* p_dev = &rec->n_eta;
    if ((int64_t)ncols * tv.m >= (1 << 16)) {
        // LDS-staged operands: 8 wavefronts, 128 columns x 128 rows per workgroup, chunks of 16 pivots
        // (64 KB of LDS, <= 128 VGPRs: two workgroups per CU, so one streams its T0 tile while the
        // other one is in its MFMA loop)
        constexpr int WC = 2, WR = 4, KC = 16;
        dim3 grid(cdiv(tv.m, WR * 32), cdiv(ncols, WC * 64));
        hipLaunchKernelGGL((k_tab_flush_lds<WC, WR, KC>), grid, dim3(WC * WR * 64), 0, s, tv, du, p_dev);
        return;
    }
    constexpr int MT = 4, NT = 2;      // wavefront tile 64 columns x 32 rows, workgroup 128 x 64
    dim3 grid(cdiv(ncols, 2 * 16 * MT), cdiv(tv.m, 2 * 16 * NT));
    hipLaunchKernelGGL((k_tab_flush<MT, NT>), grid, dim3(kThreads), 0, s, tv, du, p_dev);
}

void launch_tab_gather_columns(const TableauView& tv, const int32_t* cols, double* out, hipStream_t s) {
    const int64_t total = (int64_t)tv.m * tv.m;
    hipLaunchKernelGGL(k_tab_gather_columns, dim3(element_blocks(total)), dim3(256), 0, s, tv, cols, out);
}

void launch_tab_row(const TableauView& tv, const DeferredUpdate& du, int32_t row, double* out, const PivotRecord* rec,
                    hipStream_t s) {
    if (tv.c_hi <= tv.c_lo) return;
    hipLaunchKernelGGL(k_tab_row, dim3(tab_scan_blocks(tv.c_hi - tv.c_lo)), dim3(kThreads), 0, s, tv, du, row, out, rec);
}


}  // namespace relp
