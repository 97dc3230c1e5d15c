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
__host__ __device__ inline int64_t schedule_lds_bytes(int m, int64_t nnz, int n_levels, int n_seg = 0) {
    return lu_up16((int64_t)sizeof(LuRow) * m) + lu_up16(8 * nnz) + lu_up16(4 * nnz) + lu_up16(4 * ((int64_t)n_levels + 1)) +
           lu_up16(12 * (int64_t)n_seg);
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
// the same as a shift count: the group width is a power of two, and an integer division by a run-time value costs
// ~40 instructions on this hardware (five of them per level were most of a level's 1,200 clocks)
template <int NT>
__device__ __forceinline__ int group_shift(int rows) {
    return rows <= NT / 64 ? 6 : rows <= NT / 32 ? 5 : rows <= NT / 16 ? 4 : 3;
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
    int lg = group_shift<NT>(t1 - t0);
    LuRow pr{0, 0, 0, 0, 1.0};
    int pidx = 0; double pval = 0.0;
    bool have = t0 + (tid >> lg) < t1;
    if (have) {
        pr = rows[t0 + (tid >> lg)];
        const int l0 = tid & ((1 << lg) - 1);
        if (pr.e0 + l0 < pr.e1) { pidx = idx[pr.e0 + l0]; pval = val[pr.e0 + l0]; }
    }
    for (int lev = 0; lev < n_levels; ++lev) {
        const LuRow cr = pr; const int cidx = pidx; const double cval = pval; const bool chave = have;
        const int ct0 = t0, ct1 = t1, cG = 1 << lg;
        const int g = tid >> lg, lane = tid & (cG - 1), ngroups = NT >> lg;
        if (lev + 1 < n_levels) {
            t0 = t1; t1 = level_ptr[lev + 2];
            lg = group_shift<NT>(t1 - t0);
            have = t0 + (tid >> lg) < t1; pval = 0.0; pidx = 0;
            if (have) {
                pr = rows[t0 + (tid >> lg)];
                const int l0 = tid & ((1 << lg) - 1);
                if (pr.e0 + l0 < pr.e1) { pidx = idx[pr.e0 + l0]; pval = val[pr.e0 + l0]; }
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

// ---- the solve as a software pipeline over segments of levels (the persistent pivot kernel) -----------------------------
// Every load whose address does not depend on x is issued one (entries) or two (row headers, level offsets) levels ahead,
// so that inside a level a wavefront waits for one LDS round trip - the gather x[idx] of its own row - then reduces
// over the lane group and writes x[k].  Levels [lo, hi) on the first W threads of the workgroup:
//   kBarrier = true   every level ends with a workgroup barrier (wide levels: rows spread over W / 64 wavefronts);
//   kBarrier = false  W = 64: ONE wavefront walks a run of narrow levels (<= 8 rows each) with no barrier at all - LDS
//                     operations of a wavefront execute in order, so the write of x[k] in one level is seen by the
//                     gather of the next.  A barrier costs more than the arithmetic of a level with three rows, and the
//                     tail of a basis factor is dozens of such levels.
template <int W, bool kBarrier>
__device__ __forceinline__ void levels_pipelined(const LuRow* rows, const int32_t* idx, const double* val, const int32_t* level_ptr,
                                                 int nl, int lo, int hi, double* x) {
    const int tid = threadIdx.x;
    auto lp = [&](int i) { return level_ptr[i < nl ? i : nl]; };
    // c = the level being solved, n = the next one (header loaded), f = the one after (offsets known)
    int c_t0 = lp(lo), c_t1 = lp(lo + 1), n_t1 = lp(lo + 2), f_t1 = lp(lo + 3);
    int c_lg = group_shift<W>(c_t1 - c_t0), n_lg = group_shift<W>(n_t1 - c_t1), f_lg = group_shift<W>(f_t1 - n_t1);
    const LuRow none{0, 0, 0, 0, 0.0};
    LuRow c_h = none, n_h = none;
    bool c_on = c_t0 + (tid >> c_lg) < c_t1, n_on = c_t1 + (tid >> n_lg) < n_t1;
    if (c_on) c_h = rows[c_t0 + (tid >> c_lg)];
    if (n_on) n_h = rows[c_t1 + (tid >> n_lg)];
    int c_idx = 0; double c_val = 0.0;
    {
        const int l0 = tid & ((1 << c_lg) - 1);
        if (c_on && c_h.e0 + l0 < c_h.e1) { c_idx = idx[c_h.e0 + l0]; c_val = val[c_h.e0 + l0]; }
    }
    for (int lev = lo; lev < hi; ++lev) {
        const int c_G = 1 << c_lg, lane = tid & (c_G - 1), g = tid >> c_lg, ngroups = W >> c_lg;
        // the loads the critical path waits for: the operand of this lane's first entry and the row's own unknown
        const double xv = (c_on && c_h.e0 + lane < c_h.e1) ? x[c_idx] : 0.0;
        const double xk = (c_on && lane == 0) ? x[c_h.k] : 0.0;
        // first entry of the next level's row, header of the level after that, one more level offset
        int n_idx = 0; double n_val = 0.0;
        {
            const int ln = tid & ((1 << n_lg) - 1);
            if (n_on && n_h.e0 + ln < n_h.e1) { n_idx = idx[n_h.e0 + ln]; n_val = val[n_h.e0 + ln]; }
        }
        const int f_t0 = n_t1;
        const bool f_on = f_t0 + (tid >> f_lg) < f_t1;
        LuRow f_h = none;
        if (f_on) f_h = rows[f_t0 + (tid >> f_lg)];
        const int ff_t1 = lp(lev + 4);
        if (c_on) {
            double sum = (c_h.e0 + lane < c_h.e1) ? -c_val * xv : 0.0;
            for (int e = c_h.e0 + lane + c_G; e < c_h.e1; e += c_G) sum = fma(-val[e], x[idx[e]], sum);
            sum = group_sum(sum, c_G);
            if (lane == 0) x[c_h.k] = (xk + sum) * c_h.diag;
        }
        for (int t = c_t0 + g + ngroups; t < c_t1; t += ngroups) {        // levels wider than one pass
            const LuRow r = rows[t];
            double sum = 0.0;
            for (int e = r.e0 + lane; e < r.e1; e += c_G) sum = fma(-val[e], x[idx[e]], sum);
            sum = group_sum(sum, c_G);
            if (lane == 0) x[r.k] = (x[r.k] + sum) * r.diag;
        }
        if (kBarrier) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        c_t0 = c_t1; c_t1 = n_t1; n_t1 = f_t1; f_t1 = ff_t1;
        c_lg = n_lg; n_lg = f_lg; f_lg = group_shift<W>(f_t1 - n_t1);
        c_h = n_h; c_on = n_on; c_idx = n_idx; c_val = n_val;
        n_h = f_h; n_on = f_on;
    }
}

// NT threads stage the schedule (one contiguous copy) and solve level 0; the segments of levels run on NTW threads with
// barriers (wide) or on one wavefront without (solo), as the host marked them (Engine::lu_upload_factors).  `first_level`
// >= 1: levels below it are known to hold zeros only.  Same arithmetic and order as solve_schedule.
struct NoLap { __device__ void operator()() const {} };
template <bool kStage, int NT, int NTW = NT, class Lap = NoLap>
__device__ __forceinline__ void solve_schedule_pipelined(const DeviceSchedule& s, int m, char* base, double* x,
                                                         int first_level = 1, Lap lap = Lap()) {
    const LuRow* rows = s.rows; const int32_t* idx = s.idx; const double* val = s.val; const int32_t* level_ptr = s.level_ptr;
    const int32_t* seg = s.seg;
    if (kStage) {
        // The engine packs a schedule's arrays back to back, each padded to 16 bytes (rows, idx, val, level_ptr,
        // segments), and the LDS image keeps that layout.  16-byte loads, eight in flight per thread.
        const int64_t b_rows = lu_up16((int64_t)sizeof(LuRow) * m), b_idx = lu_up16(4 * (int64_t)s.nnz),
                      b_val = lu_up16(8 * (int64_t)s.nnz), b_lp = lu_up16(4 * ((int64_t)s.n_levels + 1)),
                      b_seg = lu_up16(12 * (int64_t)s.n_seg);
        const int n16 = (int)((b_rows + b_idx + b_val + b_lp + b_seg) / 16);
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i* src = reinterpret_cast<const v4i*>(s.rows);
        v4i* dst = reinterpret_cast<v4i*>(base);
        for (int i0 = threadIdx.x; i0 < n16; i0 += 8 * NT) {           // (clamped, unconditional: see ell_stage)
            v4i buf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) buf[u] = src[min(i0 + u * NT, n16 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[min(i0 + u * NT, n16 - 1)] = buf[u];
        }
        __syncthreads();
        rows = reinterpret_cast<const LuRow*>(base);
        idx = reinterpret_cast<const int32_t*>(base + b_rows);
        val = reinterpret_cast<const double*>(base + b_rows + b_idx);
        level_ptr = reinterpret_cast<const int32_t*>(base + b_rows + b_idx + b_val);
        seg = reinterpret_cast<const int32_t*>(base + b_rows + b_idx + b_val + b_lp);
    }
    const int nl = s.n_levels;
    const int tid = threadIdx.x;
    // level 0 = the rows without entries (often most of the factor: slack and singleton columns): one thread per row
    if (first_level <= 1) {
        const int z0 = level_ptr[0], z1 = level_ptr[nl < 1 ? nl : 1];
        for (int t = z0 + tid; t < z1; t += NT) {
            const LuRow r = rows[t];
            if (r.diag != 1.0) x[r.k] *= r.diag;
        }
        __syncthreads();
    }
    lap();                                             // (phase clock of the caller: staging + level 0 end here)
    const int l0v = first_level < 1 ? 1 : first_level;
    for (int sg = 0; sg < s.n_seg; ++sg) {
        const int lo = max(seg[3 * sg], l0v), hi = seg[3 * sg + 1], solo = seg[3 * sg + 2];
        if (lo >= hi) continue;
        if (solo) {
            if (tid < 64) levels_pipelined<64, false>(rows, idx, val, level_ptr, nl, lo, hi, x);
            __syncthreads();
        } else if (tid < NTW) {
            levels_pipelined<NTW, true>(rows, idx, val, level_ptr, nl, lo, hi, x);
        } else {
            for (int lev = lo; lev < hi; ++lev) __syncthreads();      // wavefronts that only keep the barrier count
        }
    }
}

// ---- the solve on the "ELL by pass" image (relp_lu.hpp: ell_pack) -----------------------------------------------------
// levels_pipelined above spends ~120 instructions per level on row headers, entry ranges and lane-group arithmetic, and a
// wavefront of this kernel retires roughly one instruction per 9 clocks (dependent address arithmetic), so a level costs
// ~1,100 clocks whatever its three rows hold.  Here thread t finds its (index, value) slot and its row at lane0 + t; the
// slots of the pass after next, the row descriptors of the next pass and the operands of this pass are fetched in the
// same iteration, none of these loads depending on another.
// sum over the 2^lg lanes of every row of a pass whose widest row has 2^MAXLG lanes (rows are aligned to their width; a lane
// whose row is narrower than a step's reach adds 0).  Straight-line: on this machine a branch costs ~20 clocks, an
// instruction ~5 (scripts/microbench/issue_rate.hip), so the variant is chosen once per pass, not once per step.
template <int MAXLG>
__device__ __forceinline__ double ell_reduce(double sum, int lg) {
    if (MAXLG >= 6) sum = fma(lane_plus_32(sum), lg >= 6 ? 1.0 : 0.0, sum);
    if (MAXLG >= 5) sum = fma(lane_plus_16(sum), lg >= 5 ? 1.0 : 0.0, sum);
    if (MAXLG >= 4) sum = fma(dpp_row_shl<0x108>(sum), lg >= 4 ? 1.0 : 0.0, sum);
    if (MAXLG >= 3) sum = fma(dpp_row_shl<0x104>(sum), lg >= 3 ? 1.0 : 0.0, sum);
    if (MAXLG >= 2) sum = fma(dpp_row_shl<0x102>(sum), lg >= 2 ? 1.0 : 0.0, sum);
    if (MAXLG >= 1) sum = fma(dpp_row_shl<0x101>(sum), lg >= 1 ? 1.0 : 0.0, sum);
    return sum;
}

constexpr int kEllIdxMask = (1 << kEllLg) - 1;
// The slot index array comes in two widths: 16 bits (index | lg << 13: m <= 4,095 with right-hand-side copies) and, for the
// larger bases whose work vectors no longer fit the LDS next to everything else (FtState::big), 32 bits (index | lg << 24).
template <bool kWide> struct EllIdx;
template <> struct EllIdx<false> { typedef uint16_t type; static constexpr int shift = kEllLg, mask = kEllIdxMask; };
template <> struct EllIdx<true> { typedef uint32_t type; static constexpr int shift = kEllLgWide, mask = (1 << kEllLgWide) - 1; };

// Image -> LDS (one contiguous copy), right-hand side -> its copy behind x (fused schedules); ends with a barrier when
// anything was written.  Returns the pointers of the image the solve should read.
template <bool kWide = false>
struct EllImage {
    typedef typename EllIdx<kWide>::type idx_t;
    const EllPass* passes; const int32_t* lvl_pass; const double* rdiag; const double* sval; const double* oval;
    const int32_t* rovf; const idx_t* sidx; const idx_t* oidx;
};
// kL2: x lives in global memory (layout 2 of the persistent kernel): the copies are two dependent round trips each, requested
// four at a time
// (kL2 also: x is a sparse vector, `bx` its "may be non-zero" bitmap in LDS with a bit per 2^gs entries -- relp_kernels_ft.hip,
// hs_*: only the copies of non-zero right-hand sides are made, the stale ones of an earlier sweep zeroed first)
#ifdef PASS_DIAG
// cycle stamps of the pass loop of ell_solve_pp (lane 0 of wavefronts 0 and 4, one per set), diagnostic builds only:
// [set]: clocks in own passes, [2 + set]: own passes, [4 + set]: clocks at level-end barriers, [6 + set]: barriers,
// [8 + set]: clocks fetching the next own pass, [10 + set]: clocks in the other set's passes
__device__ unsigned long long pass_diag[12];
#endif
template <int NT, class F>
__device__ __forceinline__ void ell_for_each_bit(const uint32_t* bits, int gs, int first, int n, F f) {
    const int w0 = (first >> gs) >> 5, w1 = (((n - 1) >> gs) >> 5) + 1;
    for (int w = w0 + (int)threadIdx.x; w < w1; w += NT) {
        uint32_t word = bits[w];
        while (word) {
            const int b = __ffs((int)word) - 1;
            word &= word - 1;
            const int k0 = ((w << 5) + b) << gs;
            for (int k = max(k0, first); k < min(k0 + (1 << gs), n); ++k) f(k);
        }
    }
}
template <bool kStage, int NT, bool kWide = false, bool kL2 = false>
__device__ __forceinline__ EllImage<kWide> ell_stage(const EllSchedule& s, char* base, double* x, uint32_t* bx = nullptr, int gs = 0,
                                                     int rhs_cap = 0) {
    typedef typename EllIdx<kWide>::type idx_t;
    EllImage<kWide> im{s.passes, s.lvl_pass, s.rdiag, s.sval, s.oval, s.rovf, reinterpret_cast<const idx_t*>(s.sidx),
                       reinterpret_cast<const idx_t*>(s.oidx)};
    const int tid = threadIdx.x;
    // (the wide images belong to the big layout: right-hand-side copies compacted; the all-in-LDS layout keeps a copy per
    // pivot, one LDS loop.  Everything the big layout adds is compiled out of the other instantiation: the persistent
    // kernel sits at 256 VGPRs with spills, and a few more live values cost 5 % of a 25FV47 pivot)
    if (s.rhs_base) {
        if constexpr (kL2) {
            ell_for_each_bit<NT>(bx, gs, s.m + 1, s.m + 1 + rhs_cap, [&](int k) { x[k] = 0.0; });
            __syncthreads();
            ell_for_each_bit<NT>(bx, gs, 0, s.m, [&](int k) {
                const int i = s.rhs_pos[k];
                if (i >= 0) {
                    const double v = x[k];
                    if (v != 0.0) {
                        x[s.rhs_base + i] = v;
                        atomicOr(&bx[((s.rhs_base + i) >> gs) >> 5], 1u << (((s.rhs_base + i) >> gs) & 31));
                    }
                }
            });
        } else if constexpr (kWide) { for (int i = tid; i < s.n_rhs; i += NT) x[s.rhs_base + i] = x[s.rhs_src[i]]; }
        else { for (int i = tid; i < s.m; i += NT) x[s.rhs_base + i] = x[i]; }
    }
    if (kStage) {
        const int n16 = s.bytes / 16;
        typedef int v4i __attribute__((ext_vector_type(4)));            // (an array of HIP's int4 struct ends up in scratch)
        const v4i* src = reinterpret_cast<const v4i*>(s.passes);
        v4i* dst = reinterpret_cast<v4i*>(base);
        // (loads AND stores at clamped indices, no conditions: with `if (i < n16) dst[i] = ..` the compiler sank each load
        // into its conditional block, i.e. eight dependent round trips to L2 per staging, 6,000 clocks whatever the size;
        // the threads beyond the end rewrite the last 16 bytes with the same value)
        for (int i0 = tid; i0 < n16; i0 += 8 * NT) {
            v4i buf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = min(i0 + u * NT, n16 - 1); buf[u] = src[i]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = min(i0 + u * NT, n16 - 1); dst[i] = buf[u]; }
        }
        char* q = base;
        im.passes = reinterpret_cast<const EllPass*>(q); q += lu_up16(16LL * (s.n_passes + kEllPadHeaders));
        im.lvl_pass = reinterpret_cast<const int32_t*>(q); q += lu_up16(4LL * (s.n_levels + 1));
        im.rdiag = reinterpret_cast<const double*>(q); q += lu_up16(8LL * (s.m + 1));
        im.sval = reinterpret_cast<const double*>(q); q += lu_up16(8LL * s.n_lanes);
        im.oval = reinterpret_cast<const double*>(q); q += lu_up16(8LL * s.n_ovf);
        im.rovf = reinterpret_cast<const int32_t*>(q); q += lu_up16(s.n_ovf > 0 ? 8LL * s.m : 0);
        im.sidx = reinterpret_cast<const idx_t*>(q); q += lu_up16((int64_t)sizeof(idx_t) * s.n_lanes);
        im.oidx = reinterpret_cast<const idx_t*>(q);
    }
    if (kStage || s.rhs_base) __syncthreads();
    return im;
}

// NT threads stage the image and keep the barrier count; the first NTW threads walk the passes from level `first_level`
// on (levels below it are known to hold zeros only; 0 = everything, incl. the rows without entries).  x[dummy] is a
// scratch word behind the vector: lanes that are not the first of their row store there, so the loop body has no
// divergent branch (its loads are unconditional too: lanes beyond a pass re-read its last slot and contribute 0).
// Per pass a lane issues: x[idx] (this pass), rdiag (next pass), its slot of the pass after next, one pass header.
template <bool kStage, int NT, int NTW, class Lap = NoLap>
__device__ __forceinline__ void ell_solve(const EllSchedule& s, char* base, double* x, int dummy, int first_level = 0, Lap lap = Lap()) {
    const EllImage<false> im = ell_stage<kStage, NT>(s, base, x);
    const EllPass* passes = im.passes; const double* rdiag = im.rdiag;
    const double* sval = im.sval; const double* oval = im.oval; const int32_t* rovf = im.rovf;
    const uint16_t* sidx = im.sidx; const uint16_t* oidx = im.oidx;
    const int tid = threadIdx.x;
    lap();
    const int fl = first_level < 0 ? 0 : (first_level > s.n_levels ? s.n_levels : first_level);
    const int p0 = __builtin_amdgcn_readfirstlane(im.lvl_pass[fl]), p1 = s.n_passes;
    if (p0 >= p1) return;                              // (uniform)
    if (tid >= NTW) {                                  // wavefronts that only keep the barrier count
        for (int p = p0; p < p1; ++p)
            if ((reinterpret_cast<const int4*>(passes + p)->z >> 8) & 1) __syncthreads();
        return;
    }
    // c = the pass being solved (slot and 1/diag loaded), n = the next one (slot loaded), f = the one after.  Pass headers
    // stay in vector registers (lane0, lanes, info): converting them to scalars every pass (v_readfirstlane after a wait
    // for the load) cost 210 of a pass's 770 clocks (scripts/microbench/ell_pass.hip); only `info` is made scalar, when it
    // is used, long after it arrived.  The image ends with empty headers (kEllPadHeaders), so p + 3 needs no bounds check.
    const int4* hdr = reinterpret_cast<const int4*>(passes);
    int4 hc = hdr[p0], hn = hdr[p0 + 1], hf = hdr[p0 + 2];
    auto slot_of = [&](const int4& h) { const int top = h.y > 0 ? h.y - 1 : 0; return h.x + (tid < top ? tid : top); };
    const int mm = s.m;
    int c_iv, n_iv;
    double c_val, c_diag, n_val;
    {
        const int sc = slot_of(hc), sn = slot_of(hn);
        c_iv = sidx[sc]; c_val = sval[sc];
        n_iv = sidx[sn]; n_val = sval[sn];
        c_diag = rdiag[min(c_iv & kEllIdxMask, mm)];
    }
    for (int p = p0; p < p1; ++p) {
        const bool act = tid < hc.y;
        const int c_idx = c_iv & kEllIdxMask, lg = c_iv >> kEllLg;
        const double xv = x[c_idx];                    // the one load the critical path waits for
        const double n_diag = rdiag[min(n_iv & kEllIdxMask, mm)];
        const int sf = slot_of(hf);
        const int f_iv = sidx[sf];
        const double f_val = sval[sf];
        const int4 hff = hdr[p + 3];
        const int info = __builtin_amdgcn_readfirstlane(hc.z);
        double sum = act ? -c_val * xv : 0.0;
        if ((info & 0x2ff) <= 3) sum = ell_reduce<3>(sum, lg);
        else {
            if ((info & 0x200) && act && lg == 6) {    // a row with more than 63 entries: it owns a whole wavefront
                const int k = __builtin_amdgcn_readfirstlane(c_idx);
                for (int e = rovf[2 * k] + (tid & 63); e < rovf[2 * k + 1]; e += 64) sum = fma(-oval[e], x[oidx[e]], sum);
            }
            sum = ell_reduce<6>(sum, lg);
        }
        const bool lead = act && (tid & ((1 << lg) - 1)) == 0;
        x[lead ? c_idx : dummy] = sum * c_diag;
        if (info & 0x100) __syncthreads();
        hc = hn; hn = hf; hf = hff;
        c_iv = n_iv; c_val = n_val; c_diag = n_diag;
        n_iv = f_iv; n_val = f_val;
    }
}

// The same solve with the passes dealt alternately to two sets of four wavefronts (threads 0-255: passes p0, p0 + 2, ..;
// threads 256-511: p0 + 1, p0 + 3, ..).  Wavefront w and w + 4 share a SIMD, so while one set is on the critical path of
// its pass (gather x -> multiply -> lane sums -> store -> barrier) the other fetches the slot, the 1 / diagonal and the header
// of its next pass: the bookkeeping that was half of a pass's clocks in ell_solve above overlaps with the other set's pass.
// Every wavefront joins the barrier that ends a level, whoever owned its last pass; two passes of one level may run at the
// same time (they are independent).
template <bool kStage, int NT, bool kWide = false, class Lap = NoLap, bool kL2 = false>
__device__ __forceinline__ int ell_solve_pp(const EllSchedule& s, char* base, double* x, int dummy, int first_level = 0, Lap lap = Lap(),
                                            uint32_t* bx = nullptr, int gs = 0, int rhs_cap = 0, int4* hdr_lds = nullptr) {
    static_assert(NT == 512, "two sets of 256 lanes");
    typedef typename EllIdx<kWide>::type idx_t;
    constexpr int kEllLg = EllIdx<kWide>::shift, kEllIdxMask = EllIdx<kWide>::mask;       // (shadow the 16-bit constants)
    const EllImage<kWide> im = ell_stage<kStage, NT, kWide, kL2>(s, base, x, bx, gs, rhs_cap);
    const double* rdiag = im.rdiag; const double* sval = im.sval; const double* oval = im.oval; const int32_t* rovf = im.rovf;
    const idx_t* sidx = im.sidx; const idx_t* oidx = im.oidx;
    if constexpr (kL2) {                               // the rows without entries (U, U'), those of them x is not zero in
        if (s.n_triv > 0) {
            ell_for_each_bit<NT>(bx, gs, 0, s.m, [&](int k) {
                if ((s.triv_bits[k >> 5] >> (k & 31)) & 1u) { const double v = x[k]; if (v != 0.0) x[k] = v * rdiag[k]; }
            });
            __syncthreads();
        }
    } else
    if constexpr (kWide) if (s.n_triv > 0) {           // the rows without entries (U, U'): nothing to wait for, no passes
        for (int i0 = threadIdx.x; i0 < s.n_triv; i0 += 4 * NT) {
            int k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = s.triv[min(i0 + u * NT, s.n_triv - 1)];
            // (only where x is not zero: the right-hand sides are sparse, and 1 / diagonal of an image that is not staged is a
            // dependent round trip to L2)
            if constexpr (kL2) {                      // (x in L2: operand and 1 / diagonal of all four requested together)
                asm volatile("" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]));
                double v[4], rd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v[u] = x[k[u]]; rd[u] = rdiag[k[u]]; }
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(rd[0]), "+v"(rd[1]), "+v"(rd[2]), "+v"(rd[3]));
#pragma unroll
                for (int u = 0; u < 4; ++u) if (i0 + u * NT < s.n_triv && v[u] != 0.0) x[k[u]] = v[u] * rd[u];
            } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double v = x[k[u]];
                if (i0 + u * NT < s.n_triv && v != 0.0) x[k[u]] = v * rdiag[k[u]];
            }
            }
        }
        __syncthreads();
    }
    lap();
    const int fl = first_level < 0 ? 0 : (first_level > s.n_levels ? s.n_levels : first_level);
    const int p0 = __builtin_amdgcn_readfirstlane(im.lvl_pass[fl]), p1 = s.n_passes;
    if (p0 >= p1) return 0;                            // (uniform)
    const int set = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), lt = threadIdx.x & 255;
    const int4* hdr = reinterpret_cast<const int4*>(im.passes);
    const int mm = s.m;
    auto slot_of = [&](const int4& h) { const int top = h.y > 0 ? h.y - 1 : 0; return h.x + (lt < top ? lt : top); };
    if constexpr (kL2) if (kStage || hdr_lds != nullptr) {
        // x is a sparse vector (bitmap bx): a wavefront none of whose operands may be non-zero would compute zeros over zeros, and
        // with a dozen non-zeros in 10^5 entries that is nearly every wavefront of nearly every pass (measured on a 64,000-row
        // multi-commodity LP: 1.3 of 1,050 wavefront-passes per pivot have an operand that may be non-zero).  The loop is
        // arranged around that case.  The pass headers sit in LDS (copied there unless the whole image is) and are read, like
        // the bitmap, through LDS-typed pointers (a pointer that is global on one path and LDS on another is a flat access,
        // which waits like a global one -- behind every slot word requested ahead).  Blocks of 2 kAhead passes, kAhead of them
        // this wavefront's: at the start of a block everything the NEXT block needs is requested (its headers, then the slot
        // words: an L2 round trip when the image is not staged) into the other of two register sets -- no register is copied
        // while its load is in flight (a copy waits for it) -- and the level-end flags of this block are read in one go; what
        // is left on the chain of a pass is one LDS read (the bit) and a ballot.  Value, 1 / diagonal and operand are fetched --
        // together -- only by a wavefront that has something to compute.  (A row with overflow entries owns its wavefront and
        // is always computed: those operands are not in the slots.)  What a pass of nothing still costs is its ~50
        // instructions at this kernel's ~5 clocks each.
        constexpr int kAhead = 4;
        typedef int hs_v4i __attribute__((ext_vector_type(4)));
        typedef const hs_v4i __attribute__((address_space(3)))* lds_hdr_t;
        typedef const uint32_t __attribute__((address_space(3)))* lds_bits_t;
        if constexpr (!kStage) {
            const int n16 = s.n_passes + kEllPadHeaders;
            for (int i = threadIdx.x; i < n16; i += NT) hdr_lds[i] = hdr[i];
            __syncthreads();
        }
        const lds_hdr_t hdrL = kStage ? (lds_hdr_t)reinterpret_cast<const hs_v4i*>(im.passes) : (lds_hdr_t)reinterpret_cast<const hs_v4i*>(hdr_lds);
        const lds_bits_t bxL = (lds_bits_t)bx;
        auto hdr_at = [&](int p) -> hs_v4i { return hdrL[p]; };
        auto slot_at = [&](const hs_v4i& h) { const int top = h[1] > 0 ? h[1] - 1 : 0; return h[0] + (lt < top ? lt : top); };
        bool stored = false;                               // this wavefront has written x since the last barrier
        auto prefetch = [&](int pb, int (&iqn)[kAhead], int (&yqn)[kAhead]) {
            hs_v4i hh[kAhead];
#pragma unroll
            for (int d = 0; d < kAhead; ++d) hh[d] = hdr_at(min(pb + 2 * d + set, p1 - 1));
#pragma unroll
            for (int d = 0; d < kAhead; ++d) { yqn[d] = hh[d][1]; iqn[d] = sidx[slot_at(hh[d])]; }
        };
        auto block = [&](int pb, int (&iq)[kAhead], int (&yq)[kAhead], int (&iqn)[kAhead], int (&yqn)[kAhead]) {
            int zq[2 * kAhead];
#pragma unroll
            for (int u = 0; u < 2 * kAhead; ++u) zq[u] = hdrL[min(pb + u, p1 - 1)][2];
            prefetch(pb + 2 * kAhead, iqn, yqn);
#pragma unroll
            for (int u = 0; u < 2 * kAhead; ++u) {
                const int p = pb + u;
                if (p < p1) {                              // (uniform)
                    const int info = __builtin_amdgcn_readfirstlane(zq[u]);
                    if ((u & 1) == set) {
                        const int m_iv = iq[u >> 1];
                        const int c_idx = m_iv & kEllIdxMask, lg = m_iv >> kEllLg;
                        const bool act = lt < yq[u >> 1];
                        const bool may = act && ((bxL[(c_idx >> gs) >> 5] >> ((c_idx >> gs) & 31)) & 1u);
                        if ((info & 0x200) || __ballot(may) != 0ull) {
                            const double xv = x[c_idx];
                            const double m_val = sval[slot_at(hdr_at(p))];
                            const double m_diag = rdiag[min(c_idx, mm)];
                            double sum = act ? -m_val * xv : 0.0;
                            if ((info & 0x2ff) <= 3) sum = ell_reduce<3>(sum, lg);
                            else {
                                if ((info & 0x200) && act && lg == 6) {
                                    const int k = __builtin_amdgcn_readfirstlane(c_idx);
                                    for (int e = rovf[2 * k] + (lt & 63); e < rovf[2 * k + 1]; e += 64) sum = fma(-oval[e], x[oidx[e]], sum);
                                }
                                sum = ell_reduce<6>(sum, lg);
                            }
                            const double res = sum * m_diag;
                            if (act && (lt & ((1 << lg) - 1)) == 0) {
                                x[c_idx] = res;
                                if (res != 0.0) atomicOr(&bx[(c_idx >> gs) >> 5], 1u << ((c_idx >> gs) & 31));
                            }
                            stored = true;
                        }
                    }
                    if (info & 0x100) {
                        // End of a level.  __syncthreads() would also wait for the slot words requested ahead (a release fence
                        // drains every outstanding vector memory operation).  Only a wavefront that stored waits for its stores
                        // (x) and LDS atomics (bitmap); all meet at the bare barrier.  (The wavefronts of a workgroup share the
                        // CU's vector L1: nothing to invalidate.)
                        if (stored) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stored = false; }
                        asm volatile("s_barrier" ::: "memory");
                    }
                }
            }
        };
        int iqa[kAhead], yqa[kAhead], iqb[kAhead], yqb[kAhead];
        prefetch(p0, iqa, yqa);
        for (int pb = p0; pb < p1; pb += 4 * kAhead) {
            block(pb, iqa, yqa, iqb, yqb);
            if (pb + 2 * kAhead < p1) block(pb + 2 * kAhead, iqb, yqb, iqa, yqa);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        return p1 - p0;
    }
    // my next pass (header, slot, 1 / diagonal in registers) and the header of the one after it; `z` = info word of the
    // pass the loop is at, `zn` of the next one (both sets need every pass's level-end flag)
    int4 hm = hdr[p0 + set], hm2 = hdr[p0 + set + 2];
    int m_iv = sidx[slot_of(hm)];
    double m_val = sval[slot_of(hm)];
    double m_diag = rdiag[min(m_iv & kEllIdxMask, mm)];
    int z = hdr[p0].z, zn = hdr[p0 + 1].z;
    for (int p = p0; p < p1; ++p) {
#ifdef PASS_DIAG
        const long long pd0 = clock64();
#endif
        const int znn = hdr[p + 2].z;
        const int info = __builtin_amdgcn_readfirstlane(z);
        const bool mine = ((p - p0) & 1) == set;       // (uniform per wavefront)
        bool go = mine;
        if constexpr (kL2) {
            // x is a sparse vector: a wavefront none of whose operands may be non-zero would compute zeros over zeros.  (A row
            // with overflow entries owns its wavefront and is always computed: those operands are not in the slots.)
            if (mine) {
                const int ci = m_iv & kEllIdxMask;
                const bool may = lt < hm.y && ((bx[(ci >> gs) >> 5] >> ((ci >> gs) & 31)) & 1u);
                go = (info & 0x200) || __ballot(may) != 0ull;
            }
        }
        if (go) {
            const bool act = lt < hm.y;
            const int c_idx = m_iv & kEllIdxMask, lg = m_iv >> kEllLg;
            const double xv = x[c_idx];
            double sum = act ? -m_val * xv : 0.0;
            if ((info & 0x2ff) <= 3) sum = ell_reduce<3>(sum, lg);
            else {
                if ((info & 0x200) && act && lg == 6) {
                    const int k = __builtin_amdgcn_readfirstlane(c_idx);
                    for (int e = rovf[2 * k] + (lt & 63); e < rovf[2 * k + 1]; e += 64) sum = fma(-oval[e], x[oidx[e]], sum);
                }
                sum = ell_reduce<6>(sum, lg);
            }
            const bool lead = act && (lt & ((1 << lg) - 1)) == 0;
#ifdef ELL_PRED_STORE
            if (lead) x[c_idx] = sum * m_diag;
#else
            if constexpr (kL2) {
                const double res = sum * m_diag;
                if (lead) { x[c_idx] = res; if (res != 0.0) atomicOr(&bx[(c_idx >> gs) >> 5], 1u << ((c_idx >> gs) & 31)); }
            } else
            x[lead ? c_idx : dummy] = sum * m_diag;
#endif
        }
#ifdef PASS_DIAG
        const long long pd1 = clock64();
        if (info & 0x100) __syncthreads();
        const long long pd2 = clock64();
#else
        if (info & 0x100) __syncthreads();
#endif
        if (mine) {                                    // off the critical path: the other set is solving pass p + 1
            hm = hm2;
            hm2 = hdr[p + 4];
            const int sm = slot_of(hm);
            m_iv = sidx[sm]; m_val = sval[sm];
            m_diag = rdiag[min(m_iv & kEllIdxMask, mm)];
        }
#ifdef PASS_DIAG
        if (lt == 0) {
            const long long pd3 = clock64();
            atomicAdd(&pass_diag[(mine ? 0 : 10) + set], (unsigned long long)(pd1 - pd0));
            if (mine) atomicAdd(&pass_diag[2 + set], 1ull);
            if (info & 0x100) { atomicAdd(&pass_diag[4 + set], (unsigned long long)(pd2 - pd1)); atomicAdd(&pass_diag[6 + set], 1ull); }
            if (mine) atomicAdd(&pass_diag[8 + set], (unsigned long long)(pd3 - pd2));
        }
#endif
        z = zn; zn = znn;
    }
    return p1 - p0;
}

}  // namespace relp
