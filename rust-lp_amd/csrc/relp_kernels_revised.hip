// relp_kernels_revised.hip -- hand-written gfx950 kernels of the explicit-inverse revised-simplex engine
// (and the pieces every engine shares: PRICE of the virtual columns, the ratio test, b / -pi updates, the
// deferred-update bookkeeping, the candidate exchange of the sharded engines).
//
// Every hot kernel is an HBM stream (<= 0.25 flop/byte), so the design rules are: 16-byte
// coalesced loads per lane along the contiguous dimension (columns of A, rows of B^-1), several
// independent vectors in flight per thread, 64-wide wavefront shuffle reductions, no host sync:
// per-pivot scalars travel through the device-resident PivotRecord.
//
// Reference rows implemented (SURVEY.md section 8a / 8a'):
//   k_price_structural + k_price_virtual  a2+a3  tableau/mod.rs:102-108, carry/mod.rs:572-577
//   k_select_column, k_select_partials     a2     strategy/pivot_rule.rs:38-126
//   k_build_column + k_ftran               a4     carry/basis_inverse_rows.rs:144-173
//   k_ratio                                a5     tableau/mod.rs:221-247
//   k_compute_rho + k_update_vectors       a6     carry/mod.rs:283-333, 549-570
//   k_update_inverse                       a7     carry/basis_inverse_rows.rs:42-83,131-142
//   k_weighted_column_sums                 a10    carry/mod.rs:214-248
#include "relp_device_common.h"

namespace relp {

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void price_structural_body(
    const double* __restrict__ A, int64_t ld_a, const ColumnTable& ct, const double* __restrict__ minus_pi,
    double* __restrict__ d, int p_lo, int p_hi, int cost_mode, const SelectPartials& sp, const PivotRecord* rec, int block) {
    __shared__ double s_partial[4 * kVecPerBlock];
    const int v0 = p_lo + block * kVecPerBlock;
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
        if (threadIdx.x == 0) { sp.k1[sp.offset + block] = key; sp.j[sp.offset + block] = kj; }
    }
}

__global__ __launch_bounds__(kThreads) void k_price_structural(
    const double* __restrict__ A, int64_t ld_a, ColumnTable ct, const double* __restrict__ minus_pi,
    double* __restrict__ d, int p_lo, int p_hi, int cost_mode, SelectPartials sp, const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    price_structural_body(A, ld_a, ct, minus_pi, d, p_lo, p_hi, cost_mode, sp, rec, blockIdx.x);
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
    price_virtual_body(ct, minus_pi, d, cost_mode, sp, rec, blockIdx.x);
}

// PRICE of the structural and of the virtual (artificial, slack, bound) columns in ONE launch: workgroups
// [0, nb_struct) take 8 structural columns each, the others 256 virtual columns each; their partial argmins
// land in consecutive slots.
__global__ __launch_bounds__(kThreads) void k_price_all(const double* __restrict__ A, int64_t ld_a, ColumnTable ct,
                                                        const double* __restrict__ minus_pi, double* __restrict__ d, int p_lo,
                                                        int p_hi, int cost_mode, SelectPartials sp, int nb_struct,
                                                        const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    if ((int)blockIdx.x < nb_struct) {
        price_structural_body(A, ld_a, ct, minus_pi, d, p_lo, p_hi, cost_mode, sp, rec, blockIdx.x);
    } else {
        SelectPartials spv = sp;
        spv.offset = sp.offset + nb_struct;
        price_virtual_body(ct, minus_pi, d, cost_mode, spv, rec, blockIdx.x - nb_struct);
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
        auto block_low = [&](int v) {                      // minimum over the workgroup, in every thread
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, 64));
            __syncthreads();
            if (lane == 0) s_j[wave] = v;
            __syncthreads();
            int low = s_j[0];
            for (int w = 1; w < kSingleBlock / 64; ++w) low = min(low, s_j[w]);
            return low;
        };
        if (listed <= kListMax) {
            for (int i = 0; i < listed; ++i) scan_slot(s_list[i], threadIdx.x, kSingleBlock);
            lowest = block_low(lowest);
        } else {
            // Many slots inside the band (exact ties: integer costs at the first pivots of a phase).  A slot inside the band
            // holds a column inside the band -- its own minimum -- and the slots are in column order, so the lowest index is
            // in the FIRST such slot: artificial columns (the low end of the virtual slots) before the structural slots before
            // the other virtual columns.  (One thread per slot walking its 256 columns was 58 us per pivot at 141,000 columns.)
            const int n_virt = count - sp.nb_struct;
            const int n_art_slots = min(n_virt, (ct.nr_artificial + kThreads - 1) / kThreads);
            for (int v = threadIdx.x / kThreads; v < n_art_slots; v += kSingleBlock / kThreads) {
                if (!(sp.k1[sp.nb_struct + v] <= bound)) continue;
                const int j = v * kThreads + threadIdx.x % kThreads;
                if (j < ct.nr_artificial) {
                    const double dv = d[j];
                    if (!sp.in_basis[j] && dv < -sp.tol_cost && dv <= bound && j < lowest) lowest = j;
                }
            }
            lowest = n_art_slots > 0 ? block_low(lowest) : 0x7fffffff;
            if (lowest == 0x7fffffff) {
                int first = 0x7fffffff;
                for (int t = threadIdx.x; t < sp.nb_struct; t += kSingleBlock) if (sp.k1[t] <= bound) { first = t; break; }
                first = block_low(first);
                if (first != 0x7fffffff) {
                    scan_slot(first, threadIdx.x, kSingleBlock);
                    lowest = block_low(lowest);
                }
            }
            int after = sp.nb_struct - 1;                  // the virtual slots, ascending, until one yields a column
            while (lowest == 0x7fffffff) {
                int first = 0x7fffffff;
                for (int t = after + 1 + threadIdx.x; t < count; t += kSingleBlock) if (sp.k1[t] <= bound) { first = t; break; }
                first = block_low(first);
                if (first == 0x7fffffff) break;
                scan_slot(first, threadIdx.x, kSingleBlock);
                lowest = block_low(lowest);
                after = first;
            }
        }
        if (threadIdx.x == 0) bj = lowest;
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
    if (q < 0 || !aq) return;                          // (aq == null: the caller scatters the column itself -- the persistent LU kernel)
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

// `rmin` (may be null; unsharded only): the minimum ratio b_i / alpha_i of this workgroup's kVecPerBlock rows
__global__ __launch_bounds__(kThreads) void k_ftran(const double* __restrict__ Binv, int64_t ld_b, int m,
                                                    int row_lo, int row_hi, const double* __restrict__ aq,
                                                    double* __restrict__ out, int out_offset,
                                                    const double* __restrict__ b, Tolerances tol, double* __restrict__ rmin,
                                                    const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_partial[4 * kVecPerBlock];
    const int v0 = row_lo + blockIdx.x * kVecPerBlock;
    const int i = v0 + threadIdx.x;
    const double b_i = (rmin && threadIdx.x < kVecPerBlock && i < row_hi) ? b[i] : 0.0;
    double dot = 0.0;
    block_multi_dot(Binv, ld_b, m, v0, row_hi, aq, s_partial, dot);
    if (threadIdx.x < kVecPerBlock && i < row_hi) out[i - out_offset] = dot;
    if (rmin && threadIdx.x < 64) {                               // lanes 0..7 of wavefront 0 hold the rows
        double ratio = (threadIdx.x < kVecPerBlock && i < row_hi) ? row_ratio(dot, b_i, tol) : INFINITY;
#pragma unroll
        for (int off = kVecPerBlock / 2; off > 0; off >>= 1) ratio = fmin(ratio, __shfl_down(ratio, off, 64));
        if (threadIdx.x == 0) rmin[blockIdx.x] = ratio;
    }
}

// Ratio test from the block minima k_ftran / k_apply_w leave (ratio_blocks_body): one workgroup; `rpb` rows per
// block.  Same result as k_ratio.
__global__ __launch_bounds__(kSingleBlock) void k_ratio_rows(const double* __restrict__ alpha, const double* __restrict__ b,
                                                             const int32_t* __restrict__ basis_indices, int m, Tolerances tol,
                                                             DeferredUpdate du, const double* __restrict__ rmin, int nblk,
                                                             int rpb, PivotRecord* rec) {
    const int outcome = rec->outcome, p = rec->n_eta;
    const double first = (int)threadIdx.x < nblk ? rmin[threadIdx.x] : INFINITY;      // same round trip as the record
    if (outcome != DEV_RUNNING) return;
    ratio_blocks_body<kSingleBlock>(alpha, b, basis_indices, m, tol, du, rmin, nblk, p, rec, first, true, rpb);
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
        if (br == 0.0) rec->degenerate += 1;            // ratio 0: the basis changes, the vertex does not
        rec->iterations = it + 1;
    }
}

// Rank-1 update of the explicit inverse: row_r := rho, row_i := row_i - alpha_i * rho  (i != r).
// Grid = (column strips of 512) x (row chunks); each thread owns one 16-byte column pair of the
// strip, keeps its rho pair in registers and streams the chunk's rows read-modify-write.  Rows
// with alpha_i == 0 are skipped (wave-uniform), like the reference skips absent entries.
static constexpr int kUpdRowsPerBlock = 32;
__device__ __forceinline__ void update_inverse_body(double* __restrict__ Binv, int64_t ld_b, int m, int row_lo, int row_hi,
                                                    const double* __restrict__ alpha, const double* __restrict__ rho,
                                                    const PivotRecord* rec) {
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

__global__ __launch_bounds__(kThreads) void k_update_inverse(double* __restrict__ Binv, int64_t ld_b, int m,
                                                             int row_lo, int row_hi,
                                                             const double* __restrict__ alpha,
                                                             const double* __restrict__ rho,
                                                             const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    update_inverse_body(Binv, ld_b, m, row_lo, row_hi, alpha, rho, rec);
}

// The rank-1 update of B^-1 and the update of b, -pi, -obj, basis, flags, trace in ONE launch (unsharded explicit
// path): the workgroups with blockIdx.y < ny update B^-1, the extra slab blockIdx.y == ny walks the m-vectors.  The
// two parts write disjoint data and both only read alpha, rho and the pivot's record fields.
__global__ __launch_bounds__(kThreads) void k_update_inverse_vectors(double* __restrict__ Binv, int64_t ld_b, int m,
                                                                     const double* __restrict__ alpha,
                                                                     const double* __restrict__ rho, double* __restrict__ b,
                                                                     double* __restrict__ minus_pi,
                                                                     int32_t* __restrict__ basis_indices,
                                                                     uint8_t* __restrict__ in_basis, int32_t* __restrict__ trace,
                                                                     int64_t trace_cap, int ny, PivotRecord* rec) {
    const PivotRecord R = *rec;
    if (R.outcome != DEV_RUNNING) return;
    if ((int)blockIdx.y < ny) { update_inverse_body(Binv, ld_b, m, 0, m, alpha, rho, &R); return; }
    const double d_q = R.d_q, br = R.b_r / R.alpha_r;
    const int r = R.r;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < m; i += gridDim.x * kThreads) {
        minus_pi[i] = fma(-d_q, rho[i], minus_pi[i]);
        if (i == r) b[i] = br;
        else {
            const double a = alpha[i];
            if (a != 0.0) b[i] = fma(-a, br, b[i]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int q = R.q, leaving = R.leaving;
        rec->minus_objective = fma(-d_q, br, R.minus_objective);
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

// ------------------------------------------------------------------------------------------------
// Deferred (blocked) update:  B^-1 = (I + W S') B0inv   -- see DeferredUpdate in relp_kernels.h
// ------------------------------------------------------------------------------------------------

// alpha = M v = v + W (S' v).  One thread per row; the p gathered entries v[S[j]] sit in LDS.
// `rmin` (may be null): the minimum ratio b_i / alpha_i of this workgroup's 256 rows, for k_ratio_rows
__global__ __launch_bounds__(kThreads) void k_apply_w(DeferredUpdate du, int m, const double* __restrict__ v,
                                                      double* __restrict__ alpha, const double* __restrict__ b, Tolerances tol,
                                                      double* __restrict__ rmin, const PivotRecord* rec) {
    if (rec->outcome != DEV_RUNNING) return;
    __shared__ double s_vs[kMaxEta];
    __shared__ double s_min[kThreads / 64];
    const int p = rec->n_eta;
    if ((int)threadIdx.x < p) s_vs[threadIdx.x] = v[du.S[threadIdx.x]];
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const double b_i = (rmin && i < m) ? b[i] : 0.0;
    __syncthreads();
    double ratio = INFINITY;
    if (i < m) {
        double a = v[i];
        for (int j = 0; j < p; ++j) a = fma(du.W[(int64_t)j * du.ld + i], s_vs[j], a);
        alpha[i] = a;
        ratio = row_ratio(a, b_i, tol);
    }
    if (!rmin) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ratio = fmin(ratio, __shfl_down(ratio, off, 64));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = ratio;
    __syncthreads();
    if (threadIdx.x == 0) rmin[blockIdx.x] = fmin(fmin(s_min[0], s_min[1]), fmin(s_min[2], s_min[3]));
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
    const int64_t total = (int64_t)(row_hi - row_lo) * ld_b;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / ld_b, j = idx % ld_b;
        Binv[idx] = (row_lo + i == j) ? 1.0 : 0.0;
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t seed, uint64_t stream, uint64_t idx) {
    uint64_t z = seed + stream * 0xD1B54A32D192ED03ull + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// dense column-major copy of n_cols CSC columns (entries [col_ptr[j] - col_ptr[0], ..) of row_idx / values) into a zeroed A:
// the dense engines' own copy of a sparse input, built where it lives (a 64,000 x 192,000 LP is 98 GB dense: staging it on the
// host took 30 s of the engine's creation).  A thread per column, entries in order: a repeated row index keeps its last value.
__global__ void k_csc_to_dense(const int64_t* __restrict__ col_ptr, const int32_t* __restrict__ row_idx,
                               const double* __restrict__ values, int n_cols, double* __restrict__ A, int64_t ld) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cols) return;
    const int64_t base = col_ptr[0];
    for (int64_t e = col_ptr[j] - base; e < col_ptr[j + 1] - base; ++e) A[(int64_t)j * ld + row_idx[e]] = values[e];
}

// A[i, j] = (1 + x(0, (first_column + j) * m + i) % 999) / 1000   (rust-lp_amd/synthetic.py)
__global__ void k_fill_dense(double* __restrict__ A, int64_t ld, int m, int n, uint64_t seed,
                             int64_t first_column) {
    const int64_t total = (int64_t)m * n;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = idx / m, i = idx % m;
        const uint64_t x = splitmix64(seed, 0, (uint64_t)((first_column + j) * m + i));
        A[j * ld + i] = (double)(1 + x % 999) / 1000.0;
    }
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

void launch_price_all_sel(const double* A, int64_t ld_a, const ColumnTable& ct, const double* minus_pi, double* d,
                          int32_t p_lo, int32_t p_hi, int32_t cost_mode, SelectPartials sp, const PivotRecord* rec,
                          hipStream_t s) {
    const int nb_struct = price_structural_blocks(p_lo, p_hi), nb_virt = price_virtual_blocks(ct);
    if (nb_struct + nb_virt == 0) return;
    hipLaunchKernelGGL(k_price_all, dim3(nb_struct + nb_virt), dim3(kThreads), 0, s, A, ld_a, ct, minus_pi, d, p_lo, p_hi,
                       cost_mode, sp, nb_struct, rec);
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
                       out_offset, (const double*)nullptr, Tolerances{}, (double*)nullptr, rec);
}

int32_t ftran_rows_per_block() { return kVecPerBlock; }

void launch_ftran_rmin(const double* Binv, int64_t ld_b, int32_t m, const double* aq, double* out, const double* b,
                       Tolerances tol, double* rmin, const PivotRecord* rec, hipStream_t s) {
    if (m <= 0) return;
    hipLaunchKernelGGL(k_ftran, dim3(cdiv(m, kVecPerBlock)), dim3(kThreads), 0, s, Binv, ld_b, m, 0, m, aq, out, 0, b, tol,
                       rmin, rec);
}

void launch_ratio_rows(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m, Tolerances tol,
                       const DeferredUpdate& du, const double* rmin, int32_t rows_per_block, PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_ratio_rows, dim3(1), dim3(kSingleBlock), 0, s, alpha, b, basis_indices, m, tol, du, rmin,
                       cdiv(m, rows_per_block), rows_per_block, rec);
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

void launch_update_inverse_vectors(double* Binv, int64_t ld_b, int32_t m, const double* alpha, const double* rho, double* b,
                                   double* minus_pi, int32_t* basis_indices, uint8_t* in_basis, int32_t* trace,
                                   int64_t trace_cap, PivotRecord* rec, hipStream_t s) {
    if (m <= 0) return;
    const int ny = cdiv(m, kUpdRowsPerBlock);
    dim3 grid(cdiv(ld_b, 2 * kThreads), ny + 1);
    hipLaunchKernelGGL(k_update_inverse_vectors, grid, dim3(kThreads), 0, s, Binv, ld_b, m, alpha, rho, b, minus_pi,
                       basis_indices, in_basis, trace, trace_cap, ny, rec);
}

void launch_weighted_column_sums(const double* Binv, int64_t ld_b, int32_t m, const double* w,
                                 double* minus_pi, hipStream_t s) {
    hipLaunchKernelGGL(k_weighted_column_sums, dim3(cdiv(m, 128)), dim3(128), 0, s, Binv, ld_b, m, w, minus_pi);
}

void launch_set_identity(double* Binv, int64_t ld_b, int32_t row_lo, int32_t row_hi, hipStream_t s) {
    const int64_t total = (int64_t)(row_hi - row_lo) * ld_b;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_set_identity, dim3(element_blocks(total)), dim3(256), 0, s, Binv, ld_b, row_lo, row_hi);
}

void launch_csc_to_dense(const int64_t* col_ptr, const int32_t* row_idx, const double* values, int32_t n_cols, double* A,
                         int64_t ld, hipStream_t s) {
    if (n_cols <= 0) return;
    hipLaunchKernelGGL(k_csc_to_dense, dim3(cdiv(n_cols, 256)), dim3(256), 0, s, col_ptr, row_idx, values, n_cols, A, ld);
}

void launch_fill_dense(double* A, int64_t ld, int32_t m, int32_t n, uint64_t seed, int64_t first_column,
                       hipStream_t s) {
    const int64_t total = (int64_t)m * n;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_fill_dense, dim3(element_blocks(total)), dim3(256), 0, s, A, ld, m, n, seed, first_column);
}

void launch_apply_w(const DeferredUpdate& du, int32_t m, const double* v, double* alpha, const PivotRecord* rec,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_apply_w, dim3(cdiv(m, kThreads)), dim3(kThreads), 0, s, du, m, v, alpha, (const double*)nullptr,
                       Tolerances{}, (double*)nullptr, rec);
}

void launch_apply_w_rmin(const DeferredUpdate& du, int32_t m, const double* v, double* alpha, const double* b, Tolerances tol,
                         double* rmin, const PivotRecord* rec, hipStream_t s) {
    hipLaunchKernelGGL(k_apply_w, dim3(cdiv(m, kThreads)), dim3(kThreads), 0, s, du, m, v, alpha, b, tol, rmin, rec);
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
