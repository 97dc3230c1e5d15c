// relp_kernels.h -- device-side state and kernel launchers of the explicit-inverse pivot engine.
//
// Layout in HBM (all f64 unless noted):
//   A        dense column-major  nr_constraints x nr_normal (ld_a even, column j contiguous)
//   Binv     dense row-major     m x m (ld_b multiple of 16, row i contiguous) = `BasisInverseRows`
//   minus_pi, b, alpha, aq, rho  dense m-vectors (`Carry`, carry/mod.rs:45-65)
//   d        reduced costs, one per tableau column
//   basis_indices (i32, m), in_basis (u8, n_tableau), virtual-column descriptors (i32)
//   PivotRecord: the per-pivot scalars every kernel reads instead of the host (no sync in the loop)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace relp {

enum DeviceOutcome : int32_t { DEV_RUNNING = 0, DEV_NO_CANDIDATE = 1, DEV_NO_ROW = 2 };

struct PivotRecord {
    int32_t outcome;        // DeviceOutcome; once != RUNNING every loop kernel is a no-op
    int32_t q;              // entering column (tableau index)
    double  d_q;            // its reduced cost
    int32_t r;              // pivot row
    int32_t leaving;        // basis_indices[r] before the update
    double  alpha_r;        // pivot element
    double  b_r;            // b[r] before the update
    double  minus_objective;
    long long iterations;   // basis changes so far (all phases)
    int32_t last_selected;  // FirstProfitableWithMemory state, -1 = None (pivot_rule.rs:62-64)
    int32_t phase;          // 1 or 2
    int32_t owner_has_row;  // sharded: 1 iff this rank owns row r
    int32_t pad_;
    double  key1;           // selection key of q (d_q, or its position in the search order)
    int32_t n_eta;          // deferred update: columns of W in use (distinct pivot rows since the last flush)
    int32_t eta_target;     // deferred update: column of W that receives this pivot's u
    int32_t n_eta_old;      // deferred update: n_eta before this pivot
    int32_t degenerate;     // pivots so far whose ratio b_r / alpha_r was exactly 0 (reported for config C5)
    int32_t p_now;          // n_eta as the column kernel of this pivot saw it (the fused ratio + update launch reads this copy:
                            // its first workgroup rewrites n_eta while later ones may not have started)
    int32_t pad2_;
};

// Deferred (blocked) update of the explicit inverse:  B^-1 = (I + W S') B0inv, where B0inv is the
// dense inverse at the last flush, S = [e_s1 .. e_sp] selects the distinct pivot rows since then and
// W (m x p, column-major) accumulates the elementary matrices E_k = I + u_k e_rk'.  A flush folds the
// p pivots into B0inv with one m x p x m GEMM (B0inv += W (S' B0inv)), so the 16 m^2 bytes of
// basis_inverse_rows.rs:63-83 are paid once per K pivots instead of once per pivot.
struct DeferredUpdate {
    double*  W;             // m x kmax, column j contiguous with pitch ld
    int64_t  ld;
    int32_t  kmax;
    int32_t* S;             // column j of W belongs to pivot row S[j]            (kmax)
    int32_t* pos_of_row;    // row -> column of W, or -1                          (m)
    double*  wr;            // row r of W before this pivot's update              (kmax)
    double*  R;             // flush snapshot S' B0inv, kmax x ld (row j contiguous)
};

// Read-only description of the tableau's columns (Kind + MatrixData, partially.rs:72-80,
// matrix_data.rs:308-348).  Tableau column j: j < nr_artificial -> artificial unit column,
// else provider column p = j - nr_artificial; p < nr_normal structural, else virtual.
struct ColumnTable {
    int32_t nr_artificial;
    int32_t nr_normal;
    int32_t nr_virtual;           // provider columns that are slack / bound slack
    int32_t nr_constraints;       // rows of A
    const int32_t* column_to_row; // artificial k -> row                      (nr_artificial)
    const int32_t* bound_row;     // structural p -> bound row or -1          (nr_normal)
    const int32_t* vrow0;         // virtual v -> first row                   (nr_virtual)
    const int32_t* vrow1;         // virtual v -> second row or -1 (range slack)
    const int32_t* vsign;         // virtual v -> +1 / -1 on vrow0
    const double*  cost;          // structural p -> phase-2 cost             (nr_normal)
};

// ratio_rule: relp_ratio_rule_t.  pivot_guard (relp_config_t.pivot_rescue): a pivot row whose element is below `pivot` times the
// largest |entry| of the entering column ends the loop as "no row" -- the host looks at the column and pivots with a tolerance
// relative to it or bars the column (Engine::run) -- instead of being pivoted on
struct Tolerances { double cost, pivot, zero, tie; int32_t ratio_rule, pivot_guard; double guard_rel; };

// PRICE -> entering-column choice without a second pass over d: every PRICE workgroup leaves the
// best (key, j) of its own columns here (key as in k_select_column), k_select_partials reduces them.
struct SelectPartials {
    double*        k1;        // per workgroup: best key (+inf = none)
    int32_t*       j;         // per workgroup: its column
    const uint8_t* in_basis;
    double         tol_cost;
    int32_t        rule;      // relp_pivot_rule_t
    int32_t        n;         // tableau columns (search-order wrap for FirstProfitableWithMemory)
    int32_t        offset;    // first slot this launch may use
    int32_t        nb_struct; // slots [0, nb_struct) belong to structural workgroups of 8 columns from p_lo
    double         tol_tie;   // SteepestDescent: columns within tol_tie*max(1,|min|) of the minimum tie
    int32_t        p_lo;      // first structural column priced by this rank
    int32_t        cols_per_slot;  // structural columns behind one slot (8 dense PRICE, 256 CSC PRICE)
};

// ---- sparse LU engine (device side of relp_lu.hpp) ------------------------------------------------
// One row of a triangular factor in solve order (rows of a level are contiguous): unknown k becomes
// (x[k] - sum_{e in [e0, e1)} val[e] x[idx[e]]) * diag, diag = 1 / (diagonal entry).
struct LuRow { int32_t k, e0, e1, pad_; double diag; };
struct DeviceSchedule {
    const LuRow* rows;          // m rows, level by level
    const int32_t* idx; const double* val;
    const int32_t* level_ptr;   // n_levels + 1 offsets into rows
    int32_t n_levels; int32_t nnz;
    // runs of levels (begin, end, solo) from level 1 on: solo = 1 when every level of the run has at most 8 rows, so that
    // ONE wavefront walks the run without workgroup barriers (relp_lu_device.h: levels_pipelined)
    const int32_t* seg; int32_t n_seg; int32_t pad_;
};
struct DeviceLU {
    int32_t m; int32_t pad_;
    const int32_t* rowperm;   // pivot step -> original row
    const int32_t* colperm;   // pivot step -> basis position
    DeviceSchedule Lf, Uf, Ub, Lb;
};
// The same solve packed "ELL by pass" for the persistent pivot kernel (relp_lu.hpp: ell_pack; relp_lu_device.h: ell_solve):
// one image per schedule, contiguous in device memory in the order of the members below, every array padded to 16 bytes.
static constexpr int kEllLg = 13;                       // sidx = index | lg << kEllLg (relp_lu.hpp: kEllLgShift)
static constexpr int kEllLgWide = 24;                   // the same in the 32-bit images of large bases (FtState::big)
static constexpr int kEllPadHeaders = 4;                // empty pass headers behind the last one: the solves read ahead unchecked
struct EllPass { int32_t lane0, lanes, info, level; };     // info: max lg | last-of-level << 8 | overflow << 9
struct EllSchedule {
    const EllPass*  passes;      // n_passes (+ 3 empty headers)
    const int32_t*  lvl_pass;    // n_levels + 1: first pass of a level (= group of fused levels, relp_lu.hpp: fuse_levels)
    double*         rdiag;       // m + 1: 1 / diagonal by pivot; 0 = row masked by a Forrest-Tomlin update
    double*         sval;        // n_lanes; an update zeroes the entries whose substitution path runs through the masked pivot
    const double*   oval;        // n_ovf: entries beyond the 63rd of a row
    const int32_t*  rovf;        // 2 m (or nothing when n_ovf = 0): overflow range by pivot
    const uint16_t* sidx;        // n_lanes: index | lg << 13; index >= rhs_base: the copy of the right-hand side behind x
    const uint16_t* oidx;        // n_ovf
    const int32_t*  via_ptr;     // m + 1 (U, U' only, else null): positions in sval to zero when a pivot is masked ...
    const int32_t*  via_pos;     // ... (not part of the staged image)
    int32_t n_passes, n_levels, m, n_lanes, n_ovf;
    int32_t bytes;               // size of the image
    int32_t rhs_base;            // m + 1 when some entry reads the right-hand-side copy, else 0 (no copy is made)
    int32_t n_triv;              // rows without entries of U / U' (relp_lu.hpp: EllPacked::triv): x[k] *= rdiag[k] before the passes
    const int32_t* triv;         // n_triv pivots (global, not part of the staged image)
    const int32_t* reach;        // m: first group in which x[p] matters (EllPacked::reach) -> where a sweep may start
    const int32_t* rhs_src;      // n_rhs: slot index rhs_base + i reads the right-hand side of pivot rhs_src[i] (EllPacked::rhs_src)
    int32_t n_rhs, pad_;
    // layout 2 of the persistent kernel (x as a sparse vector: relp_kernels_ft.hip, "hs_"), else null:
    const int32_t* rhs_pos;      // m: i with rhs_src[i] == pivot, or -1 (the inverse of rhs_src)
    const uint32_t* triv_bits;   // m bits: pivot is in `triv`
};
// Column indices at or above this value are artificial variables that survived phase 1 (see
// Engine::switch_to_phase_two): INT32_MAX - (na - 1 - a).  They have no flag, no cost and no column.
static constexpr int32_t kWrappedArtificialBase = 0x40000000;

struct DeviceCSC { const int64_t* col_ptr; const int32_t* row_idx; const double* values; };

// ---- sparse LU engine with the Forrest-Tomlin update on the device (relp_kernels_ft.hip) -------------------------
// lower_upper/mod.rs:92-155 keeps `updates: Vec<(EtaFile, RotateToBack)>` and rewrites U at every basis change.  Here L and
// U0 of the last refactorisation are never modified (their level schedules stay valid).  A "pivot" is a step k of the
// factorisation P B Q = L U0; the reference's position of a pivot in U is its rank in the order
// [pivots never updated, ascending k | updated pivots, by the time of their last update], so its rotate-to-back needs no
// data movement.  Update number s ("slot" s, pivot p):
//   * row p and column p of U0 are masked (the pull rows of p get 1/diag = 0, x[p] reads as 0 in the U0 sweeps),
//   * the row eta r = u_bar U^-1 (R = I - e_p r', eta_file.rs:10-18): sparse part over never-updated pivots in the eta
//     pool, part over updated pivots in row s of TC (below the diagonal),
//   * the spike L^-1 a_q (after the earlier etas) becomes the last column of U: sparse part over never-updated pivots in
//     the spike pool, part over updated pivots in column s of TC (on and above the diagonal; TC[s][s] = new diagonal).
// A pivot updated again gets a new slot; its old slot is dead (row / column of TC zeroed, diagonal 1).
static constexpr int kFtThreads = 512;                 // the persistent pivot workgroup: 8 wavefronts
static constexpr int kFtWaves = kFtThreads / 64;       // sparse lists are bucketed by (pivot % kFtWaves): one wavefront per bucket
static constexpr int kFtMaxSlots = 64;                 // one lane of a wavefront per slot in the chains over TC
static constexpr int kFtMaxRows = 1 << 20;             // (layout 2 keeps no per-row array in LDS; the spike pool is tcap x m pairs)
static constexpr int kFtLdsBudget = 156 * 1024;        // of the CU's 160 KB
// Layout 2 keeps "may be non-zero" bitmaps of x (with its right-hand-side copies) and of the spike in LDS, one bit per 2^shift
// entries: the smallest shift with which both fit kFtBitmapBytes.
static constexpr int kFtBitmapBytes = 40 * 1024;
static inline __host__ __device__ int ft_bitmap_shift(int m, int rhs_cap) {
    int s = 0;
    while (((((long long)m + 1 + rhs_cap) >> s) + 64 + (((long long)m) >> s) + 64) / 8 > kFtBitmapBytes) ++s;
    return s;
}
// Everything the Forrest-Tomlin update needs to know about the leaving pivot p, together in one cache line segment (it was
// five dependent global round trips: task -> row header -> entries, via_ptr -> via_pos, twice).
struct FtPivotInfo {
    int32_t u_e0, u_e1;          // row p of U right of the diagonal: entries [u_e0, u_e1) of DeviceLU::Uf idx / val
    int32_t via_u0, via_u1;      // EllSchedule::via_pos range of p in the U image ...
    int32_t via_t0, via_t1;      // ... and in the U' image
    int32_t lev_ub, pad_;
};
struct FtState {
    int32_t  m, tcap, ldt, eta_cap;
    int32_t* hdr;            // [0] updates since the refactorisation (t), [1] eta pool entries in use, [2] refactor requested
    int32_t* slot_pivot;     // tcap
    int32_t* slot_prev;      // tcap: earlier slot of the same pivot or -1
    int32_t* slot_live;      // tcap
    int32_t* tslot;          // m: live slot of a pivot, -1 = never updated
    double*  TC;             // tcap x ldt
    int32_t* eta_off;        // tcap x (kFtWaves + 1): bucket offsets into the eta pool (absolute)
    int32_t* eta_idx; double* eta_val;     // eta_cap
    int32_t* spk_off;        // tcap x (kFtWaves + 1): bucket offsets relative to the slot's region s * m
    int32_t* spk_idx; double* spk_val;     // tcap * m
    const int32_t* inv_rowperm;            // original row -> pivot
    const int32_t* inv_colperm;            // basis position -> pivot
    const int32_t* task_uf;                // pivot -> index of its row in the U (FTRAN) schedule
    const int32_t* task_ub;                // pivot -> index of its row in the U' (BTRAN) schedule
    EllSchedule ell[4];                    // L, U, U', L' packed for the persistent kernel
    const int32_t* lev_ub;                 // pivot -> level of its row in the U' schedule (a solve with e_p or u_bar starts there)
    const struct FtPivotInfo* pinfo;       // per pivot: what an update of that pivot needs, one 32-byte load
    int32_t* journal;                      // (basis position, entering column) of every basis change of the last k_ft_run
                                           // launch, hdr[3] of them: what k_ft_replay applies to freshly computed factors
    double*  spike;          // m: the spike of the last FTRAN (step-wise API: consumed by the next update)
    double*  sp_work;        // m: the spike inside the kernels when it does not live in LDS (big)
    int32_t  hyper;          // bit k: sweep k (L, U, U', L') starts at the first group its right-hand side reaches (RELP_FT_HYPER)
    int32_t  pad2_;
    double*  x_work;         // m + 1 + rhs_cap: the work vector inside the kernels when it does not live in LDS (layout 2)
    unsigned long long* chunk_mask;    // ceil(m / 64) words of scratch (layout 2: ft_compact)
    int32_t* nz_idx; double* nz_val;   // m each (layout 2): the non-zeros of the entering column as (row, alpha), RATIO and the update of b
    int32_t* rho_idx;                  // m (layout 2): the rows where the last pivot row rho is not zero
    int32_t* nzc;                      // [0] entries of nz_idx that describe pb.alpha, [1] of rho_idx / pb.rho (-1: unknown, the
                                       // vectors are rewritten densely), [2] bits_save holds the bitmaps of x and the spike
    uint32_t* bits_save;               // the two LDS bitmaps between launches
    int32_t  big;            // layout (relp_kernels_ft.hip: ft_layout).  0: everything in LDS.  1: spike, permutations and eta pool in
                             // global memory (L2), 32-bit slot indices.  2: x, -pi and the pivot -> slot table there as well (no per-row
                             // array in LDS: any m), 32-bit row indices in the PRICE copy
    int32_t  rhs_cap;        // words behind x[m] for the right-hand-side copies of the fused schedules (0: levels are not fused)
    int32_t  stage[4];       // which of the four schedules (L, U, U', L') fit the LDS staging area
    int32_t  stage_bytes;
    int32_t  lds_bytes;      // dynamic LDS of every FT kernel
    int32_t  max_updates;    // refactor when this many updates are pending (<= tcap)
    int32_t  pad_;
    long long* prof;         // 16 phase accumulators of the persistent kernel (shader clocks, thread 0) or nullptr
};
// PRICE copy of the structural columns for the persistent kernel, k-major ("ELL"): slot k of column p at [k * n + p], so
// that a thread pricing column p issues all its loads at once, coalesced, with no column offsets and no entry loop.
// Tier A: the first kPriceSlots entries of EVERY column (bit 15 of slot 0's index marks a column that has more).  Tier B:
// the long columns once more, complete (up to kPriceLongSlots entries, k-major over n_long), with their column number, so a
// long column is priced by one thread in the column's own order from loads that do not depend on each other.  Columns
// beyond kPriceLongSlots entries (`very_long`) are priced from the CSC arrays.  Padding slots are (0, 0.0).
static constexpr int kPriceSlots = 8;
static constexpr int kPriceLongSlots = 24;
static constexpr int kPriceLongFlag = 0x8000;
static constexpr uint32_t kPriceLongFlag32 = 0x80000000u;
struct PriceEll {
    const uint16_t* idx; const double* val;            // kPriceSlots x nr_normal
    const uint16_t* lidx; const double* lval;          // kPriceLongSlots x n_long
    const uint32_t* idx32; const uint32_t* lidx32;     // the two index arrays 32 bits wide (FtState::big == 2: rows beyond 32,767), else null
    const int32_t* long_cols;                          // n_long
    const int32_t* very_long;                          // n_very_long
    const uint16_t* long_of;                           // nr_normal: index of a column in tier B, 0xFFFF = not there
    int32_t n_long, n_very_long;
};
// What the host wants to know when the pivot kernel returns, written by the kernel itself into pinned, device-mapped host
// memory: after the stream is synchronised it is simply there (no copies, no second synchronisation): the record, the
// header of the update file and the basis (the snapshot a look-ahead refactorisation factorises).
struct FtMirror {
    PivotRecord rec;
    int32_t hdr[4];
    int32_t walked[4], whole[4]; // this launch: passes walked / passes of the whole schedule, summed over its sweeps of L, U, U', L'
    int32_t sweeps[4];           // (the host switches the hyper-sparse start of a schedule off while it saves less than it costs)
    int32_t basis[1];            // m entries
};
struct FtProblem {           // what the persistent kernel needs besides the factors
    DeviceCSC csc; ColumnTable ct; PriceEll pe;
    double *minus_pi, *b, *alpha, *rho, *d;
    int32_t* basis; uint8_t* in_basis; int32_t* trace; int64_t trace_cap;
    PivotRecord* rec;
    FtMirror* mirror;        // or null
    Tolerances tol;
    int32_t rule, n, phase;
    int32_t external_price;  // 1: PRICE ran as a grid launch (k_price_csc + k_select_partials): the entering column is in the record
};

// ---- launchers (all asynchronous on `s`) -------------------------------------------------------
// cost_mode: 0 = no cost term (tableau row), 1 = phase-1 costs (artificial: 1), 2 = phase-2 costs
// PRICE, structural part: d[na + p] = c_p + (-pi[0:mc]) . A[:,p] (+ -pi[bound_row]) for p in [p_lo, p_hi)
void launch_price_structural(const double* A, int64_t ld_a, const ColumnTable& ct, const double* minus_pi,
                             double* d, int32_t p_lo, int32_t p_hi, int32_t cost_mode, const PivotRecord* rec,
                             hipStream_t s);
// structural columns outside [p_lo, p_hi) get +inf (sharded pricing)
void launch_price_mask_unowned(const ColumnTable& ct, double* d, int32_t p_lo, int32_t p_hi,
                               const PivotRecord* rec, hipStream_t s);
// fused PRICE + partial selection (hot loop): same results as the two-kernel form
void launch_price_structural_sel(const double* A, int64_t ld_a, const ColumnTable& ct, const double* minus_pi,
                                 double* d, int32_t p_lo, int32_t p_hi, int32_t cost_mode, SelectPartials sp,
                                 const PivotRecord* rec, hipStream_t s);
void launch_price_virtual_sel(const ColumnTable& ct, const double* minus_pi, double* d, int32_t cost_mode,
                              SelectPartials sp, const PivotRecord* rec, hipStream_t s);
// both of the above in one launch (unsharded loop): structural workgroups first, then the virtual ones
void launch_price_all_sel(const double* A, int64_t ld_a, const ColumnTable& ct, const double* minus_pi, double* d,
                          int32_t p_lo, int32_t p_hi, int32_t cost_mode, SelectPartials sp, const PivotRecord* rec,
                          hipStream_t s);
int32_t price_structural_blocks(int32_t p_lo, int32_t p_hi);
int32_t price_virtual_blocks(const ColumnTable& ct);
// reduce `count` partials, record q / d_q, and build aq (= k_select_column + k_build_column)
void launch_select_partials(SelectPartials sp, int32_t count, const double* d, const double* A, int64_t ld_a,
                            const ColumnTable& ct, int32_t m, double* aq, PivotRecord* rec, hipStream_t s);
// ratio test followed by the deferred-update bookkeeping of k_eta_prepare (du.kmax = 0: ratio only)
void launch_ratio_eta(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m, Tolerances tol,
                      const DeferredUpdate& du, PivotRecord* rec, hipStream_t s);
// PRICE, artificial + virtual columns
void launch_price_virtual(const ColumnTable& ct, const double* minus_pi, double* d, int32_t cost_mode,
                          const PivotRecord* rec, hipStream_t s);
// entering-column choice over d (masked by in_basis), rule = relp_pivot_rule_t
void launch_select_column(const double* d, const uint8_t* in_basis, int32_t n, int32_t rule, double tol_cost,
                          double tol_tie, PivotRecord* rec, hipStream_t s);
// aq := dense column rec->q in tableau row space (m entries)
void launch_build_column(const double* A, int64_t ld_a, const ColumnTable& ct, int32_t m, double* aq,
                         const PivotRecord* rec, hipStream_t s);
// FTRAN: alpha[i] = Binv[i,:] . aq for i in [row_lo, row_hi); out[i - out_offset]
void launch_ftran(const double* Binv, int64_t ld_b, int32_t m, int32_t row_lo, int32_t row_hi,
                  const double* aq, double* out, int32_t out_offset, const PivotRecord* rec, hipStream_t s);
// RATIO TEST over alpha/b/basis_indices (two-pass tie rule)
void launch_ratio(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m,
                  Tolerances tol, PivotRecord* rec, hipStream_t s);
// rho = Binv[r,:] / alpha_r if row r in [row_lo,row_hi) else 0   (ld_b entries, padding zero)
void launch_compute_rho(const double* Binv, int64_t ld_b, int32_t m, int32_t row_lo, int32_t row_hi,
                        double* rho, PivotRecord* rec, hipStream_t s);
// b, -pi, -obj, basis_indices, in_basis, trace, iteration counter
void launch_update_vectors(int32_t m, const double* alpha, const double* rho, double* b, double* minus_pi,
                           int32_t* basis_indices, uint8_t* in_basis, int32_t* trace, int64_t trace_cap,
                           PivotRecord* rec, hipStream_t s);
// rank-1 update of rows [row_lo,row_hi) of Binv: row_r = rho; row_i -= alpha_i * rho
void launch_update_inverse(double* Binv, int64_t ld_b, int32_t m, int32_t row_lo, int32_t row_hi,
                           const double* alpha, const double* rho, const PivotRecord* rec, hipStream_t s);
// launch_update_vectors + launch_update_inverse in one launch (unsharded explicit path)
void launch_update_inverse_vectors(double* Binv, int64_t ld_b, int32_t m, const double* alpha, const double* rho, double* b,
                                   double* minus_pi, int32_t* basis_indices, uint8_t* in_basis, int32_t* trace,
                                   int64_t trace_cap, PivotRecord* rec, hipStream_t s);

// phase switch: minus_pi[j] = -sum_i w[i] * Binv[i,j]  (w = cost of basis column of row i)
void launch_weighted_column_sums(const double* Binv, int64_t ld_b, int32_t m, const double* w,
                                 double* minus_pi, hipStream_t s);
void launch_set_identity(double* Binv_local, int64_t ld_b, int32_t row_lo, int32_t row_hi, hipStream_t s);
void launch_fill_dense(double* A, int64_t ld, int32_t m, int32_t n, uint64_t seed, int64_t first_column,
                       hipStream_t s);
// A (zeroed, column-major, ld) := the n_cols CSC columns whose pointers start at col_ptr[0] (row_idx / values hold their entries only)
void launch_csc_to_dense(const int64_t* col_ptr, const int32_t* row_idx, const double* values, int32_t n_cols, double* A,
                         int64_t ld, hipStream_t s);

// deferred update (see DeferredUpdate)
// alpha[i] = v[i] + sum_j W[i,j] v[S[j]] for every row
void launch_apply_w(const DeferredUpdate& du, int32_t m, const double* v, double* alpha, const PivotRecord* rec,
                    hipStream_t s);
// FTRAN / alpha = v + W (S'v) that also leave the minimum ratio b_i / alpha_i of every block of rows (8 rows per
// k_ftran workgroup, 256 per k_apply_w workgroup), and the ratio test that starts from those minima (same result
// as launch_ratio / launch_ratio_eta; re-reads only the blocks inside the tie band).  Unsharded engines.
int32_t ftran_rows_per_block();
void launch_ftran_rmin(const double* Binv, int64_t ld_b, int32_t m, const double* aq, double* out, const double* b,
                       Tolerances tol, double* rmin, const PivotRecord* rec, hipStream_t s);
void launch_apply_w_rmin(const DeferredUpdate& du, int32_t m, const double* v, double* alpha, const double* b, Tolerances tol,
                         double* rmin, const PivotRecord* rec, hipStream_t s);
void launch_ratio_rows(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m, Tolerances tol,
                       const DeferredUpdate& du, const double* rmin, int32_t rows_per_block, PivotRecord* rec, hipStream_t s);
// wr = W[r,:], pick the target column (existing column of row r or a new one), book-keeping
void launch_eta_prepare(const DeferredUpdate& du, PivotRecord* rec, hipStream_t s);
// W <- W + u W[r,:], then column target += / = u, with u = eta - e_r built from alpha
void launch_update_w(const DeferredUpdate& du, int32_t m, const double* alpha, const PivotRecord* rec, hipStream_t s);
// rho = sum over owned rows in {r} U S of coefficient * B0inv[row,:]  (coefficient 1 for r, W[r,j] for S[j])
void launch_rho_deferred(const DeferredUpdate& du, const double* Binv, int64_t ld_b, int32_t m, int32_t row_lo,
                         int32_t row_hi, double* rho, PivotRecord* rec, hipStream_t s);
// B0inv[rows] += W[rows,:] (S' B0inv); the snapshot R must hold S' B0inv of ALL ranks' rows
void launch_flush_snapshot(const DeferredUpdate& du, const double* Binv, int64_t ld_b, int32_t row_lo, int32_t row_hi,
                           const PivotRecord* rec, hipStream_t s);
void launch_flush_apply(const DeferredUpdate& du, double* Binv, int64_t ld_b, int32_t m, int32_t row_lo,
                        int32_t row_hi, const PivotRecord* rec, hipStream_t s);
void launch_flush_reset(const DeferredUpdate& du, PivotRecord* rec, hipStream_t s);

// ---- dense-tableau engine ------------------------------------------------------------------------
// T = B^-1 [all columns] is kept as T = (I + W S') T0 with T0 dense column-major (m x n_store, column
// pitch ld_t) and R0 = S' T0 (the rows of T0 at the distinct pivot rows of the block, row pitch ld_r).
// Storage column = tableau column + col_off (phase 2 skips the artificial block).
struct TableauView {
    double*  T0;      int64_t ld_t;
    double*  R0;      int64_t ld_r;
    double*  d;       // reduced cost per storage column, maintained incrementally
    int32_t  m;
    int32_t  n_store; // stored columns (artificial block + provider columns)
    int32_t  col_off; // first storage column of the current phase's tableau
    int32_t  n;       // tableau columns of the current phase
    int32_t  c_lo, c_hi;  // storage columns owned by this rank (T0, R0 hold only these; pointers are
                          // pre-shifted so that kernels index by global storage column)
};
// T0 := original matrix in row space (artificial unit columns | A + bound rows | virtual unit columns)
void launch_tab_build(const TableauView& tv, const double* A, int64_t ld_a, const ColumnTable& ct, hipStream_t s);
// d[c] = cost[c] - w . T0[:,c] for every stored column (w = cost of the basic variable of each row)
void launch_tab_price_init(const TableauView& tv, const double* w, const double* cost_store, hipStream_t s);
// partial argmin over d (one slot per 256 columns) -- used when the loop is (re)entered
// w[i] = cost of the basic variable of row i; with launch_tab_price_init it recomputes d from T0
void launch_tab_basis_costs(const TableauView& tv, const int32_t* basis_indices, const double* cost_store, double* w,
                            hipStream_t s);
void launch_tab_scan(const TableauView& tv, SelectPartials sp, const PivotRecord* rec, hipStream_t s);
int32_t tab_scan_blocks(int32_t n_owned_columns);
// re-tabulation: T0[:, c] = (LU)^-1 a_c for the stored columns [c_first, c_first + c_count) in one launch (a_c as in
// launch_tab_build); false when the solve vector does not fit into LDS
bool launch_lu_ftran_cols(const DeviceLU& lu, const TableauView& tv, const double* A, int64_t ld_a, const ColumnTable& ct,
                          int32_t c_first, int32_t c_count, hipStream_t s);
// entering column from the partials (no column build: the tableau column is read directly)
void launch_tab_select(const TableauView& tv, SelectPartials sp, int32_t count, PivotRecord* rec, hipStream_t s);
// alpha = T[:,q] = T0[:,q] + W R0[:,q]
void launch_tab_column(const TableauView& tv, const DeferredUpdate& du, double* alpha, const PivotRecord* rec,
                       hipStream_t s);
// row r of T (appending T0[r,:] to R0 when r is new in the block), d -= (d_q/alpha_r) row, partial argmin
void launch_tab_row_update(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, PivotRecord* rec,
                           hipStream_t s);
// b, -obj, basis_indices, in_basis, trace, iteration counter (no -pi: it is read off d)
void launch_tab_update_vectors(int32_t m, const double* alpha, double* b, int32_t* basis_indices, uint8_t* in_basis,
                               int32_t* trace, int64_t trace_cap, PivotRecord* rec, hipStream_t s);
// launch_update_w + launch_tab_update_vectors in one launch
void launch_tab_update_w_vectors(const DeferredUpdate& du, int32_t m, const double* alpha, double* b,
                                 int32_t* basis_indices, uint8_t* in_basis, int32_t* trace, int64_t trace_cap,
                                 PivotRecord* rec, hipStream_t s);
// launch_tab_select + launch_tab_column in one launch (every workgroup reduces the partials itself)
void launch_tab_select_column(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t count,
                              double* alpha, PivotRecord* rec, hipStream_t s);
// single-GPU loop: also leaves the minimum ratio of every block of 256 rows in `rmin`, and the ratio test that
// starts from those minima (re-reads only the row blocks inside the tie band)
void launch_tab_select_column_rmin(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t count,
                                   double* alpha, const double* b, Tolerances tol, double* rmin, PivotRecord* rec,
                                   hipStream_t s, const double* shadow = nullptr, int32_t* shadow_meta = nullptr);
void launch_ratio_blocks(const double* alpha, const double* b, const int32_t* basis_indices, int32_t m, Tolerances tol,
                         const DeferredUpdate& du, const double* rmin, PivotRecord* rec, hipStream_t s);
// sharded engines: this rank's candidate message [key, j, d_j, alpha (m), minimum ratio per block of 256 rows
// (cdiv(m, 256))] instead of the record, and the choice among the gathered messages + ratio test from the
// winner's block minima
void launch_tab_select_column_msg(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t count,
                                  double* msg, const double* b, Tolerances tol, PivotRecord* rec, hipStream_t s,
                                  const double* shadow = nullptr, int32_t* shadow_meta = nullptr);
// forced_row >= 0: the winner enters in that row at zero level (no ratio test), phase_one.rs:246-250
void launch_tab_select_candidate_ratio(const double* msgs, int32_t count, int64_t msg_len, int32_t m, double* alpha,
                                       const double* b, const int32_t* basis_indices, int32_t rule, Tolerances tol,
                                       const DeferredUpdate& du, int32_t forced_row, PivotRecord* rec, hipStream_t s);
// sharded removal of basic artificial variables: PRICE partials (key = column index) of the owned columns that may
// replace the variable basic in `row` at zero level
void launch_tab_zero_level_scan(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t row,
                                int32_t nr_artificial, Tolerances tol, const PivotRecord* rec, hipStream_t s);
// winner among the gathered candidates + ratio test + block bookkeeping in one launch (tableau engine)
void launch_select_candidate_ratio(const double* msgs, int32_t count, int64_t msg_len, int32_t m, double* alpha,
                                   const double* b, const int32_t* basis_indices, int32_t rule, Tolerances tol,
                                   const DeferredUpdate& du, PivotRecord* rec, hipStream_t s);
// launch_tab_row_update + launch_tab_update_w_vectors in one launch (disjoint workgroup ranges)
void launch_tab_update_all(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t m,
                           const double* alpha, double* b, int32_t* basis_indices, uint8_t* in_basis, int32_t* trace,
                           int64_t trace_cap, PivotRecord* rec, hipStream_t s);
// flush: T0 += W R0 with v_mfma_f64_16x16x4_f64 tiles
void launch_tab_flush(const TableauView& tv, const DeferredUpdate& du, const PivotRecord* rec, hipStream_t s);
// Ratio test + update in ONE launch (single-GPU loop): every workgroup repeats the ratio test from the block minima the column
// kernel left (same code, same answer), then does its share of the update.  What one workgroup rewrites while another may
// still read it is double-buffered: b and the basis array (in -> out, the caller swaps them), row r of W (its new values go
// to `shadow`, {row, length} to `shadow_meta`; the next column kernel or launch_tab_apply_shadow folds them into W), n_eta
// (read as PivotRecord::p_now).
void launch_tab_ratio_update_all(const TableauView& tv, const DeferredUpdate& du, SelectPartials sp, int32_t m,
                                 const double* alpha, const double* b_in, double* b_out, const int32_t* basis_in,
                                 int32_t* basis_out, uint8_t* in_basis, int32_t* trace, int64_t trace_cap, Tolerances tol,
                                 const double* rmin, double* shadow, int32_t* shadow_meta, PivotRecord* rec, hipStream_t s,
                                 const double* msgs = nullptr, int32_t count = 0, int64_t msg_len = 0, int32_t rule = 0);
// (msgs: sharded loop -- the gathered candidate messages; the winner's column and block minima are taken from there and
// `alpha` / `rmin` are ignored)
void launch_tab_apply_shadow(const DeferredUpdate& du, double* shadow, int32_t* shadow_meta, hipStream_t s);
// out[i, k] = T0[i, cols[k]] (row-major m x m): B^-1 from the identity columns
void launch_tab_gather_columns(const TableauView& tv, const int32_t* cols, double* out, hipStream_t s);
// tableau row `row` of the CURRENT T into out[0..n_store) (remove_artificial_basis_variables)
void launch_tab_row(const TableauView& tv, const DeferredUpdate& du, int32_t row, double* out, const PivotRecord* rec,
                    hipStream_t s);

// ---- sparse LU engine ------------------------------------------------------------------------------
// PRICE over CSC columns (thread per column), partial argmin per 256 columns
void launch_price_csc(const DeviceCSC& csc, const ColumnTable& ct, const double* vec, double* d, int32_t p_lo,
                      int32_t p_hi, int32_t cost_mode, SelectPartials sp, const PivotRecord* rec, hipStream_t s);
// the same plus the virtual columns (launch_price_virtual_sel) in one launch
void launch_price_csc_all(const DeviceCSC& csc, const ColumnTable& ct, const double* vec, double* d, int32_t nr_normal,
                          int32_t cost_mode, SelectPartials sp, int32_t nb_virtual, const PivotRecord* rec, hipStream_t s);
int32_t price_csc_blocks(int32_t p_lo, int32_t p_hi);
// k_select_partials with the entering column scattered from CSC instead of copied from dense A
void launch_select_partials_csc(SelectPartials sp, int32_t count, const double* d, const DeviceCSC& csc,
                                const ColumnTable& ct, int32_t m, double* aq, PivotRecord* rec, hipStream_t s);
void launch_build_column_csc(const DeviceCSC& csc, const ColumnTable& ct, int32_t m, double* aq, const PivotRecord* rec,
                             hipStream_t s);
// FTRAN: v = (LU)^-1 aq  (one persistent workgroup, level-scheduled pull solves, work vector in LDS)
void launch_lu_ftran(const DeviceLU& lu, const double* aq, double* v, double* scratch, const PivotRecord* rec,
                     hipStream_t s);
// BTRAN: rho' = z' (LU)^-1 with z = e_r + sum_j W[r,j] e_S[j] (rhs == nullptr, r from rec or `row` >= 0)
//        or z = rhs (dense, indexed by basis position)
void launch_lu_btran(const DeviceLU& lu, const DeferredUpdate& du, const double* rhs, int32_t row, double* rho,
                     double* scratch, const PivotRecord* rec, hipStream_t s);
// every row of B^-1 = (LU)^-1 in one launch (workgroup i: e_i' B^-1 -> out + i * ld); `none`: a DeferredUpdate with kmax = 0
void launch_lu_btran_rows(const DeviceLU& lu, const DeferredUpdate& none, double* out, int64_t ld, double* scratch,
                          hipStream_t s);

// ---- Forrest-Tomlin engine (relp_kernels_ft.hip) ---------------------------------------------------------------
// bytes of dynamic LDS the FT kernels need besides the staging area
size_t ft_lds_base_bytes(int32_t m, int32_t tcap, int32_t eta_cap, int32_t tier, int32_t rhs_cap);
// LDS bytes a schedule needs to be staged (relp_lu_device.h: schedule_lds_bytes)
int64_t ft_schedule_stage_bytes(int32_t m, int64_t nnz, int32_t n_levels, int32_t n_seg);
// up to `max_pivots` whole pivots (PRICE -> FTRAN -> RATIO -> FT update -> BTRAN -> b, -pi, basis) in ONE launch of one
// workgroup; stops early when the outcome is decided or a refactorisation is due (hdr[2] = 1)
void launch_ft_run(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int64_t max_pivots, hipStream_t s);
// alpha = B^-1 a for tableau column `column` (>= 0; -2: rec->q) or the dense right-hand side `rhs` (indexed by original
// row); leaves the spike in st.spike
void launch_ft_ftran(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int32_t column, const double* rhs, double* alpha,
                     hipStream_t s);
// rho' = c' B^-1 with c = e_row (row >= 0; -2: rec->r) or the dense `rhs` (indexed by basis position)
void launch_ft_btran(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int32_t row, const double* rhs, double* rho,
                     hipStream_t s);
// the Forrest-Tomlin update for the basis change in position rec->r with the spike in st.spike
void launch_ft_update(const DeviceLU& lu, const FtState& st, const FtProblem& pb, hipStream_t s);
// the first `count` basis changes of the journal as Forrest-Tomlin updates of the (fresh) factors; hdr[2] = 2 when one fails
void launch_ft_replay(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int32_t count, hipStream_t s);

// sharded helpers
void launch_pack_candidate(const double* aq, int32_t m, double* msg, PivotRecord* rec, hipStream_t s);
void launch_select_candidate(const double* msgs, int32_t count, int64_t msg_len, int32_t m, double* aq,
                             int32_t rule, double tol_tie, PivotRecord* rec, hipStream_t s);
void launch_gather_alpha(const double* slices, int32_t count, int32_t stride, int32_t m, double* alpha,
                         const PivotRecord* rec, hipStream_t s);
void launch_pad_slice(double* slice, int32_t valid, int32_t stride, hipStream_t s);

}  // namespace relp
