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

// A triangular schedule packed for the persistent pivot kernel ("ELL by pass"): a level is executed in passes of up to
// 256 lanes; inside a pass every row owns 2^lg consecutive lanes (widths sorted descending so that every row starts at a
// multiple of its width).  Lane 0 of a row is the row's own unknown as a slot (k, -1.0), lanes 1.. hold its entries
// (index, value): the lane sum of -value * x[index] is x[k] - sum_e val_e x[idx_e], which the first lane multiplies by
// 1 / diagonal (rdiag, indexed by pivot) and stores to x[k].  Thread t of the workgroup finds its slot at lane0 + t: no
// ranges, no row descriptors.  Entries beyond the 63rd of a row live in an overflow list (rare).  Padding slots are (0, 0.0).
struct EllPassHost { int32_t lane0, lanes, info, level; };     // info: max lg | (last pass of its level) << 8 | (has overflow rows) << 9
struct EllPacked {
    std::vector<EllPassHost> passes;
    std::vector<int32_t> lvl_pass;       // level -> first pass (n_levels + 1)
    std::vector<double> rdiag;           // m: 1 / diagonal by pivot (1.0 for rows that are left out)
    std::vector<double> sval, oval;
    std::vector<int32_t> rovf;           // 2 m: overflow entries [begin, end) by pivot; empty when no row overflows
    std::vector<uint16_t> sidx, oidx;    // sidx: index | lg << 12
};
// keep_trivial: also pack the rows without entries whose diagonal is 1 (needed when rows can be masked later: U, U')
void ell_pack(const TriangularSchedule& t, bool keep_trivial, EllPacked* out);

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
