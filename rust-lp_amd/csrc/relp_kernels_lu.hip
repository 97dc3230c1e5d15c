// relp_kernels_lu.hip -- kernels of the sparse LU engine (RELP_ENGINE_LU): CSC PRICE and the
// level-scheduled, LDS-resident triangular solves (FTRAN / BTRAN).  Reference rows a4 (LU), a8:
// carry/lower_upper/mod.rs:157-222.
#include "relp_lu_device.h"

#include <algorithm>
#include <vector>

namespace relp {

// ------------------------------------------------------------------------------------------------
// Sparse LU engine: CSC PRICE and the level-scheduled triangular solves
// ------------------------------------------------------------------------------------------------
// d_j = c_j + sum_i (-pi)_i a_ij over the stored entries of column j (vector/dense.rs:81-92), thread per
// column; the workgroup's best (key, j) goes to the PRICE partials.
// nb_struct > 0: the workgroups from nb_struct on price the virtual columns in the same launch (price_virtual_body)
__global__ __launch_bounds__(kThreads) void k_price_csc(DeviceCSC csc, ColumnTable ct, const double* __restrict__ vec,
                                                        double* __restrict__ d, int p_lo, int p_hi, int cost_mode,
                                                        SelectPartials sp, int nb_struct, const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    if (nb_struct > 0 && (int)blockIdx.x >= nb_struct) {
        SelectPartials spv = sp;
        spv.offset = sp.offset + nb_struct;
        price_virtual_body(ct, vec, d, cost_mode, spv, rec, blockIdx.x - nb_struct);
        return;
    }
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
__device__ __forceinline__ void lu_btran_body(const DeviceLU& lu, const DeferredUpdate& du, const double* __restrict__ rhs,
                                              int row, double* __restrict__ rho, double* __restrict__ scratch,
                                              const PivotRecord* rec) {
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

template <bool kXLds, bool kStage1, bool kStage2>
__global__ __launch_bounds__(kLuThreads) void k_lu_btran(DeviceLU lu, DeferredUpdate du, const double* __restrict__ rhs,
                                                           int row, double* __restrict__ rho, double* __restrict__ scratch,
                                                           const PivotRecord* rec) {
    if (rec && rec->outcome != DEV_RUNNING) return;
    lu_btran_body<kXLds, kStage1, kStage2>(lu, du, rhs, row, rho, scratch, rec);
}

// All m rows of B^-1 at once (re-inversion, warm start): workgroup i solves e_i' B^-1 and writes row i of `out`.
// x lives in each workgroup's LDS (the launcher falls back to one launch per row otherwise).
template <bool kStage1, bool kStage2>
__global__ __launch_bounds__(kLuThreads) void k_lu_btran_rows(DeviceLU lu, DeferredUpdate none, double* __restrict__ out,
                                                                int64_t ld) {
    lu_btran_body<true, kStage1, kStage2>(lu, none, nullptr, blockIdx.x, out + (int64_t)blockIdx.x * ld, nullptr, nullptr);
}


// Single-workgroup selection over the reduced costs.  key = (k1, j) lexicographic minimum:
//   SteepestDescent: k1 = d_j (strict `<` => lowest j wins ties, pivot_rule.rs:118)
//   FirstProfitable[WithMemory]: k1 = position of j in the search order (pivot_rule.rs:88)
int32_t price_csc_blocks(int32_t p_lo, int32_t p_hi) { return p_hi > p_lo ? cdiv(p_hi - p_lo, kThreads) : 0; }

void launch_price_csc(const DeviceCSC& csc, const ColumnTable& ct, const double* vec, double* d, int32_t p_lo,
                      int32_t p_hi, int32_t cost_mode, SelectPartials sp, const PivotRecord* rec, hipStream_t s) {
    const int blocks = price_csc_blocks(p_lo, p_hi);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_price_csc, dim3(blocks), dim3(kThreads), 0, s, csc, ct, vec, d, p_lo, p_hi, cost_mode, sp, 0, rec);
}

// structural (CSC) and virtual columns in one launch; `nb_virtual` = price_virtual_blocks(ct)
void launch_price_csc_all(const DeviceCSC& csc, const ColumnTable& ct, const double* vec, double* d, int32_t nr_normal,
                          int32_t cost_mode, SelectPartials sp, int32_t nb_virtual, const PivotRecord* rec, hipStream_t s) {
    const int nb_struct = price_csc_blocks(0, nr_normal);
    if (nb_struct == 0 || nb_virtual == 0) {            // degenerate shapes: the two-launch path handles them
        launch_price_csc(csc, ct, vec, d, 0, nr_normal, cost_mode, sp, rec, s);
        return;
    }
    hipLaunchKernelGGL(k_price_csc, dim3(nb_struct + nb_virtual), dim3(kThreads), 0, s, csc, ct, vec, d, 0, nr_normal,
                       cost_mode, sp, nb_struct, rec);
}

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

// Re-tabulation of the dense tableau: workgroup c solves B^-1 a_c for stored column c (artificial unit column,
// structural column of A + its bound row, virtual unit column -- the same definition as k_tab_build) and writes
// column c of T0.  x lives in LDS.
template <bool kStage1, bool kStage2>
__global__ __launch_bounds__(kLuThreads) void k_lu_ftran_cols(DeviceLU lu, TableauView tv, const double* __restrict__ A,
                                                                int64_t ld_a, ColumnTable ct, int c_first) {
    extern __shared__ __align__(16) char lds[];
    double* x = reinterpret_cast<double*>(lds);
    char* base = lds + lu_up16((int64_t)lu.m * 8);
    const int c = c_first + blockIdx.x;
    for (int k = threadIdx.x; k < lu.m; k += blockDim.x) {
        const int i = lu.rowperm[k];
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
        x[k] = v;
    }
    __syncthreads();
    solve_schedule<kStage1>(lu.Lf, lu.m, base, x);
    solve_schedule<kStage2>(lu.Uf, lu.m, base, x);
    double* out = tv.T0 + (int64_t)c * tv.ld_t;
    for (int k = threadIdx.x; k < lu.m; k += blockDim.x) out[lu.colperm[k]] = x[k];
}

// returns false when x does not fit into LDS (the caller then leaves the tableau as it is)
bool launch_lu_ftran_cols(const DeviceLU& lu, const TableauView& tv, const double* A, int64_t ld_a, const ColumnTable& ct,
                          int32_t c_first, int32_t c_count, hipStream_t s) {
    const LuLdsPlan p = plan_lu_lds(lu.m, lu.Lf, lu.Uf);
    if (!p.x_in_lds) return false;
    if (c_count <= 0) return true;
    auto fn = p.stage_first ? (p.stage_second ? k_lu_ftran_cols<true, true> : k_lu_ftran_cols<true, false>)
                            : (p.stage_second ? k_lu_ftran_cols<false, true> : k_lu_ftran_cols<false, false>);
    allow_big_lds(reinterpret_cast<const void*>(fn));
    hipLaunchKernelGGL(fn, dim3(c_count), dim3(kLuThreads), p.bytes, s, lu, tv, A, ld_a, ct, c_first);
    return true;
}

void launch_lu_btran_rows(const DeviceLU& lu, const DeferredUpdate& none, double* out, int64_t ld, double* scratch,
                          hipStream_t s) {
    const LuLdsPlan p = plan_lu_lds(lu.m, lu.Ub, lu.Lb);
    if (!p.x_in_lds) {                                   // x in global scratch: one solve at a time
        for (int32_t i = 0; i < lu.m; ++i) launch_lu_btran(lu, none, nullptr, i, out + (int64_t)i * ld, scratch, nullptr, s);
        return;
    }
    auto fn = p.stage_first ? (p.stage_second ? k_lu_btran_rows<true, true> : k_lu_btran_rows<true, false>)
                            : (p.stage_second ? k_lu_btran_rows<false, true> : k_lu_btran_rows<false, false>);
    allow_big_lds(reinterpret_cast<const void*>(fn));
    hipLaunchKernelGGL(fn, dim3(lu.m), dim3(kLuThreads), p.bytes, s, lu, none, out, ld);
}

void launch_lu_btran(const DeviceLU& lu, const DeferredUpdate& du, const double* rhs, int32_t row, double* rho,
                     double* scratch, const PivotRecord* rec, hipStream_t s) {
    const LuLdsPlan p = plan_lu_lds(lu.m, lu.Ub, lu.Lb);
    auto fn = pick_lu_variant(p, k_lu_btran<false, false, false>, k_lu_btran<true, false, false>,
                              k_lu_btran<true, true, false>, k_lu_btran<true, false, true>, k_lu_btran<true, true, true>);
    allow_big_lds(reinterpret_cast<const void*>(fn));
    hipLaunchKernelGGL(fn, dim3(1), dim3(kLuThreads), p.bytes, s, lu, du, rhs, row, rho, scratch, rec);
}


}  // namespace relp
