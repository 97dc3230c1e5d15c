// relp_lu_device.h -- device code shared by the kernels of the sparse LU engine (relp_kernels_lu.hip: one solve per
// launch; relp_kernels_ft.hip: the persistent pivot kernel with the Forrest-Tomlin update): the level-scheduled,
// LDS-resident triangular solve and its lane-group reductions.
#pragma once
#include "relp_device_common.h"

namespace relp {

// ---- level-scheduled triangular solves ---------------------------------------------------------------
// One persistent workgroup of 256 threads.  The work vector lives in LDS; when the factor itself fits
// next to it (the usual case for Netlib-sized bases) its rows and entries are staged into LDS first, so
// that a level costs an LDS round trip and a barrier instead of a chain of dependent global loads.
// 8 to 64 lanes share one row (coalesced entry loads, DPP reduction), and each group fetches its
// first row of the NEXT level - row header and first entries do not depend on x - before it waits at
// the barrier of the current one.
static constexpr int kLuThreads = 256;          // one solve per launch: 4 wavefronts (cheap barriers, 32 rows per pass)
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
// The two steps that cross rows of 16 lanes use the gfx950 lane swaps (v_permlane32_swap / v_permlane16_swap,
// register to register) instead of LDS-routed shuffles: lane i receives lane i + 32 / i + 16 for the lanes that
// matter (i < 32 / the first row of each half), the same summation order as with __shfl_down.
__device__ __forceinline__ double lane_plus_32(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double lane_plus_16(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double group_sum(double v, int G) {
    if (G >= 64) v += lane_plus_32(v);
    if (G >= 32) v += lane_plus_16(v);
    if (G >= 16) v += dpp_row_shl<0x108>(v);
    v += dpp_row_shl<0x104>(v);
    v += dpp_row_shl<0x102>(v);
    v += dpp_row_shl<0x101>(v);
    return v;
}
// Lanes per row for a level of `rows` rows: a level with few rows (the dense rows of the bump come one
// per level) gets a whole wavefront per row.
template <int NT>
__device__ __forceinline__ int group_lanes(int rows) {
    return rows <= NT / 64 ? 64 : rows <= NT / 32 ? 32 : rows <= NT / 16 ? 16 : 8;
}

// kStage: copy the schedule into LDS at `base` and solve from there; otherwise solve from global memory.
// NT = threads of the workgroup.
template <bool kStage, int NT = kLuThreads>
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
    int G = group_lanes<NT>(t1 - t0);
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
        const int g = tid / cG, lane = tid % cG, ngroups = NT / cG;
        if (lev + 1 < n_levels) {
            t0 = t1; t1 = level_ptr[lev + 2];
            G = group_lanes<NT>(t1 - t0);
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

}  // namespace relp
