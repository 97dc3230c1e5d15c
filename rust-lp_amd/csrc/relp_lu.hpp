// relp_lu.hpp -- host-side sparse LU factorisation of the basis (P B Q = L U) and the level
// schedules the device solves use.
//
// Reference: tableau/inverse_maintenance/carry/lower_upper/decomposition/mod.rs:27-138 (right-looking
// elimination on a row-major working copy) and decomposition/pivoting.rs:45-81 (Markowitz: minimise
// (row count - 1)(column count - 1)).  This runs at every refactorisation (lower_upper/mod.rs:199-202:
// after more than 10 updates; carry/mod.rs:602-614 re-inverts from the original columns).
// f64 differences, none of which changes what is computed (B^-1 a_q is the same vector up to rounding):
// the Markowitz search is restricted to the sparsest active row and the sparsest active column (the
// reference scans every remaining entry, pivoting.rs:59), and a pivot must be at least 0.1 of the largest
// active entry of its column (threshold partial pivoting; exact arithmetic needs no threshold).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace relp {

// One triangular factor in the "pull" form the device kernel uses: row k lists the entries
// (idx, val) it must subtract before it is final, rows are grouped into levels (all rows of a level
// depend only on rows of earlier levels).
struct TriangularSchedule {
    std::vector<int32_t> ptr;        // m + 1
    std::vector<int32_t> idx;        // position (pivot step) of the other unknown
    std::vector<double>  val;
    std::vector<double>  diag;       // m (1.0 for the unit-diagonal factor)
    std::vector<int32_t> level_ptr;  // n_levels + 1
    std::vector<int32_t> level_rows; // rows sorted by level
};

struct LUFactors {
    int32_t m = 0;
    std::vector<int32_t> rowperm;    // pivot step k -> original row
    std::vector<int32_t> colperm;    // pivot step k -> basis position (column of B)
    TriangularSchedule Lf;           // FTRAN: L y = P a         (rows of L)
    TriangularSchedule Uf;           // FTRAN: U x = y           (rows of U, solved from the back)
    TriangularSchedule Ub;           // BTRAN: U' t = Q' c       (columns of U)
    TriangularSchedule Lb;           // BTRAN: L' z = t          (columns of L, solved from the back)
    int64_t nnz_l = 0, nnz_u = 0;
};

// columns[j] = sorted sparse column j of the basis matrix B (m columns over m rows).
// Returns false (and a message) when B is numerically singular.
bool lu_factor(int32_t m, const std::vector<std::vector<std::pair<int32_t, double>>>& columns, LUFactors* out,
               std::string* err);
// the same from a flat copy of the columns (column j = entries [ptr[j], ptr[j + 1]) of idx / val)
bool lu_factor_csc(int32_t m, const int64_t* ptr, const int32_t* idx, const double* val, LUFactors* out, std::string* err);

// Levels for a schedule whose rows (ptr / idx / val) are given, e.g. by the device factorisation (relp_lu_factor_core.h):
// fills diag, level_ptr, level_rows.  ascending: the dependencies of a row have smaller indices.
void lu_levels_from_rows(int32_t m, const std::vector<double>& diag, bool ascending, TriangularSchedule* s);

// Level fusion by local inversion.  A basis factor of an LP has dozens of levels of a handful of rows each, and on the device a
// level costs a fixed ~800 clocks whatever it holds, so consecutive levels are merged into GROUPS that one
// pass solves: a row r of the group that depends on a row j of the same group gets j's equation substituted,
//     x_r = (b_r - sum_k v_rk x_k - v_rj x_j) / d_r,   x_j = (b_j - sum_k v_jk x_k) / d_j
//  => x_r = (b_r - sum_k v_rk x_k - (v_rj / d_j) b_j + sum_k (v_rj v_jk / d_j) x_k) / d_r,
// recursively over the levels of the group: every row then reads solved values x_k of EARLIER groups and raw right-hand
// sides b_j of its own group only, so the rows of a group are independent of each other.  The solve keeps a copy of the
// right-hand side behind x for this: an entry index >= rhs_base means b[index - rhs_base].  A group is closed when a row
// would exceed 63 entries or the group one pass (lane_cap lanes).
// `maskable` (U, U'): a Forrest-Tomlin update deletes row and column p of U by masking pivot p (1 / d_p := 0), which must
// also cancel every substituted term that ran through p.  Such schedules keep one entry per substitution PATH (duplicates
// of an index are not combined) and list, per pivot, the entries whose path runs through it (`via`): the update zeroes them.
struct FusedSchedule {
    TriangularSchedule s;                // levels = groups; idx may be >= rhs_base
    int32_t rhs_base = 0;                // m + 1 (x[m] is the solve's scratch word)
    std::vector<int32_t> start_after;    // by pivot p: a right-hand side that is zero on p and on everything solved before p
                                         // lets the sweep start at group start_after[p] + 1
    std::vector<int32_t> via_ptr, via_ent;   // maskable: CSR by pivot of entry positions in s.idx / s.val
};
void fuse_levels(const TriangularSchedule& t, bool maskable, bool keep_trivial, int32_t lane_cap, FusedSchedule* out);

// A triangular schedule packed for the persistent pivot kernel ("ELL by pass"): a level (group) is executed in passes of up
// to 256 lanes; inside a pass every row owns 2^lg consecutive lanes (widths sorted descending so that every row starts at a
// multiple of its width).  Lane 0 of a row is the row's own unknown as a slot (k, -1.0), lanes 1.. hold its entries
// (index, value): the lane sum of -value * x[index] is x[k] - sum_e val_e x[idx_e], which the first lane multiplies by
// 1 / diagonal (rdiag, indexed by pivot) and stores to x[k].  Thread t of the workgroup finds its slot at lane0 + t: no
// ranges, no row descriptors.  Entries beyond the 63rd of a row live in an overflow list (rare).  Padding slots are (0, 0.0).
struct EllPassHost { int32_t lane0, lanes, info, level; };     // info: max lg | (last pass of its level) << 8 | (has overflow rows) << 9
constexpr int32_t kEllLgShift = 13;      // sidx = index | lg << 13: indices up to 2 m + 1 (right-hand-side copies) need m <= 4095
constexpr int32_t kEllLgShiftWide = 24;  // sidx32 = index | lg << 24: the images of larger bases
struct EllPacked {
    std::vector<EllPassHost> passes;
    std::vector<int32_t> lvl_pass;       // level -> first pass (n_levels + 1)
    std::vector<double> rdiag;           // m + 1: 1 / diagonal by pivot (1.0 for rows that are left out and for the scratch word)
    std::vector<double> sval, oval;
    std::vector<int32_t> rovf;           // 2 m: overflow entries [begin, end) by pivot; empty when no row overflows
    std::vector<uint16_t> sidx, oidx;    // sidx: index | lg << kEllLgShift
    std::vector<uint32_t> sidx32, oidx32;    // `wide` packing instead: index | lg << kEllLgShiftWide (sidx / oidx stay empty)
    size_t lanes() const { return sidx.empty() ? sidx32.size() : sidx.size(); }
    size_t overflow() const { return oidx.empty() ? oidx32.size() : oidx.size(); }
    std::vector<int32_t> via_ptr, via_pos;   // maskable schedules: CSR by pivot of positions in sval to zero when the pivot is masked
    // keep_trivial schedules (U, U'): the rows without entries are not packed as slots -- one lane each, they were a dozen
    // passes of a near-diagonal factor -- but listed here: x[k] *= rdiag[k], one loop in front of the passes
    std::vector<int32_t> triv;
    // The right-hand-side copies the fused rows read, compacted: slot index rhs_base + i means b[rhs_src[i]] (fuse_levels
    // writes rhs_base + pivot; only the pivots some row was substituted through need a copy -- a few hundred of thousands)
    std::vector<int32_t> rhs_src;
    // by pivot p: the first group (level of the fused schedule) in which x[p] matters -- p's own row if it is packed, and every
    // row that reads x[p] or its right-hand-side copy; 0x7fffffff = none.  A sweep whose right-hand side is zero wherever
    // reach[p] < g may start at group g (hyper-sparse start).
    std::vector<int32_t> reach;
};
// keep_trivial: also pack the rows without entries whose diagonal is 1 (needed when rows can be masked later: U, U')
// wide: 32-bit slots.  compact_rhs: number the right-hand-side copies consecutively (rhs_src) instead of rhs_base + pivot.
// triv_min: rows without entries are listed in `triv` when there are at least this many of them, else packed as slots.
void ell_pack(const FusedSchedule& f, bool keep_trivial, EllPacked* out, bool wide = false, bool compact_rhs = false,
              int32_t triv_min = 0x7fffffff);

// Factors given literally (P = Q = I), the way the reference's tests build a `LUDecomposition { lower_triangular,
// upper_triangular, .. }` (lower_upper/mod.rs:44-52): L column-major, unit diagonal implied, entries (row > column);
// U column-major, entries (row <= column) with the diagonal among them.
bool lu_from_triangles(int32_t m, const std::vector<std::vector<std::pair<int32_t, double>>>& lcols,
                       const std::vector<std::vector<std::pair<int32_t, double>>>& ucols, LUFactors* out, std::string* err);

// Host-side solves with the factors (warm starts, tests): x <- B^-1 a (x indexed by basis position on
// return, a by original row) and z' <- c' B^-1 (c indexed by basis position, z by original row).
void lu_ftran_host(const LUFactors& f, const std::vector<double>& a, std::vector<double>* x);
void lu_btran_host(const LUFactors& f, const std::vector<double>& c, std::vector<double>* z);

}  // namespace relp
