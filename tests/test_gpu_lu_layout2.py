"""The LU engine beyond one CU's LDS (missing 2: "beyond one CU's LDS put x in L2 ... and measure that against the
product-form fallback instead of silently using the latter").

Layout 2 of the persistent pivot kernel keeps no per-row array in LDS: x, -pi and the pivot -> slot table live in global memory
(L2), the LDS holds the dense tail of U, the slot tables and the staged factor image.  With Dantzig's rule over >= 32,768
columns PRICE runs as a grid launch per pivot (one workgroup pricing 140,000 columns was 70 % of a pivot).  The reference's own
files end below 9,000 rows (the larger ones -- KEN-*, PDS-*, STOCFOR3 -- are rejected by its MPS reader, restated literally in
rust-lp_amd/mps.py), so the cases here are synthetic: `synthetic.multicommodity_lp` has the shape of the KEN / PDS files
(node-arc incidence blocks coupled by bundle capacity rows), `synthetic.sparse_lp` is the random sparse LP of the earlier rounds.

Parity: the same pivots as the f64 CPU oracle (carry/lower_upper/mod.rs:92-222 restated in oracle/relp_f64.c) on a prefix the
oracle finishes in seconds, the same pivots as the other device engines further on, and B^-1 B = I / primal feasibility of the
basis the run ends on.
"""
import os

import numpy as np
import pytest

import rust_lp_amd  # noqa: F401
from oracle import relp_f64
from rust_lp_amd import MatrixData, engine, synthetic

pytestmark = pytest.mark.gpu


def _prefix(md, pivots):
    ref = relp_f64.OracleF64(md)
    ref.run(max_iters=pivots)
    return ref


def _run(t, pivots):
    total = 0
    while total < pivots:
        done, oc = t.run(pivots - total)
        total += done
        if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE):
            break
    return total


def test_multicommodity_11k_rows_takes_layout_2_and_the_oracles_pivots():
    """m = 11,190 (10 commodities x 799 nodes + 3,200 arcs): beyond the 17 bytes per row layout 1 has room for."""
    md = MatrixData.from_sparse_dict(synthetic.multicommodity_lp(800, 3200, 10, 7))
    t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
    lay = t.lu_kernel_layout()
    assert lay["persistent_kernel"] and lay["layout"] == 2 and lay["tail_slots"] == 64, lay       # (layout 2: interval 64, look-ahead 16)
    n = 6000
    assert _run(t, n) == n
    ref = _prefix(md, n)
    assert t.trace() == ref.trace
    assert abs(t.objective_function_value() - ref.objective) <= 1e-9 * max(1.0, abs(ref.objective))
    ident, basic, min_b = t.check_basis()
    assert ident <= 1e-9 and basic <= 1e-9 and min_b >= -1e-9
    # the same pivots from the dense tableau engine (an independent device path) over the same stretch
    d = engine.Tableau(md, engine=engine.ENGINE_TABLEAU, trace_capacity=1 << 16)
    assert _run(d, n) == n and d.trace() == t.trace()


def test_random_sparse_11k_rows_layout_2_against_the_oracle():
    """synthetic.sparse_lp(7000, 21000): m = 11,234 with the bound rows; 1,500 pivots of the oracle take ~10 s."""
    md = MatrixData.from_sparse_dict(synthetic.sparse_lp(7000, 21000, 7))
    t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
    assert t.lu_kernel_layout()["layout"] == 2
    n = 1500
    assert _run(t, n) == n
    ref = _prefix(md, n)
    assert t.trace() == ref.trace
    assert np.allclose(t.b(), ref.b(), rtol=1e-9, atol=1e-9)


def test_phase_two_with_grid_price_takes_the_tableau_engines_pivots():
    """All rows <=, A >= 0, c < 0: Dantzig's rule from the first pivot over 33,000 + columns -> PRICE as a grid launch per
    pivot (k_price_csc + k_select_partials, then ONE pivot of the persistent kernel), incl. the launches of a look-ahead
    refactorisation and the one that only marks the refactorisation due."""
    d = synthetic.sparse_lp(10000, 26000, 11, frac_eq=0.0, frac_ge=0.0)
    d["values"] = np.abs(d["values"]); d["b"] = np.abs(d["b"]) + 1.0; d["c"] = -d["c"]
    md = MatrixData.from_sparse_dict(d)
    t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
    lay = t.lu_kernel_layout()
    assert lay["layout"] == 2 and lay["grid_price"], lay
    n = 2000
    assert _run(t, n) == n and t.phase == 2
    st = t.lu_stats()
    assert st["refactorisations"] >= n // 64 and st["lookahead_installs"] > 0 and st["lookahead"] == 16        # batches, look-ahead batches
    ref = _prefix(md, 600)
    assert t.trace()[:600] == ref.trace
    dense = engine.Tableau(md, engine=engine.ENGINE_TABLEAU, trace_capacity=1 << 16)
    assert _run(dense, n) == n and dense.trace() == t.trace()
    assert abs(dense.objective_function_value() - t.objective_function_value()) <= 1e-9 * abs(t.objective_function_value())


@pytest.mark.parametrize("forced", [{"RELP_FT_BIG": "2"}, {"RELP_FT_BIG": "2", "RELP_FT_GRID_PRICE": "1"}, {"RELP_FT_GRID_PRICE": "1"}])
def test_layout_2_and_grid_price_forced_on_25fv47(forced, monkeypatch):
    """Both switches are read at create: the reference's C3 file through layout 2 / the grid PRICE ends at the pin with the
    pivots of the default layout (tests/netlib/test.rs:157: 5501.8459)."""
    from lp_files import load
    gf, ex, md, emd = load("netlib/25FV47.SIF")
    base = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
    assert base.lu_kernel_layout()["layout"] == 0
    assert base.solve_relaxation() == engine.OPTIMAL
    for k, v in forced.items():
        monkeypatch.setenv(k, v)
    t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
    lay = t.lu_kernel_layout()
    assert lay["layout"] == (2 if "RELP_FT_BIG" in forced else 0) and lay["grid_price"] == ("RELP_FT_GRID_PRICE" in forced)
    assert t.solve_relaxation() == engine.OPTIMAL
    assert abs(t.objective_function_value() + float(gf.fixed_cost) - 5501.8459) < 1e-4
    same = sum(1 for a, b in zip(t.trace(), base.trace()) if a == b)
    assert same >= 1000, same                              # (the bucket order of the update lists differs between the layouts: sums
    #                                                        round differently, the traces part where a tie flips)


def test_the_fallback_is_what_a_forced_small_layout_leaves():
    """RELP_FT_BIG=0 on 11,190 rows: no layout may be taken, the engine says so (product-form loop), and takes the same pivots."""
    md = MatrixData.from_sparse_dict(synthetic.multicommodity_lp(800, 3200, 10, 7))
    os.environ["RELP_FT_BIG"] = "0"
    try:
        t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
    finally:
        del os.environ["RELP_FT_BIG"]
    assert not t.lu_kernel_layout()["persistent_kernel"]
    n = 600
    assert _run(t, n) == n
    assert t.trace() == _prefix(md, n).trace


def test_tableau_beyond_2_to_the_32_elements():
    """A launch takes at most 2^32 - 1 threads: the element-wise set-up kernels of a tableau with more stored elements than that
    (here 4,100 x 1,060,000 = 4.3e9, 35 GB) were refused without a word and the engine pivoted on an empty tableau (found at
    63,988 x 255,988, where the dense tableau engine declared a feasible LP infeasible).  Grid-stride loops now; the first
    pivots must be those of the explicit-inverse engine, which prices the same device-generated matrix directly."""
    import torch
    m, n, seed = 4100, 1060000, 12345
    free, _ = torch.cuda.mem_get_info()
    if free < 100 * (1 << 30):
        pytest.skip("needs 100 GB of device memory")
    lib = engine.load_library()
    A = torch.empty((n, m), dtype=torch.float64, device="cuda")                     # column-major m x n, 35 GB
    assert lib.relp_synth_fill_dense(A.data_ptr(), m, m, n, seed, 0, None) == 0     # (itself one of the element-wise kernels)
    torch.cuda.synchronize()
    # the last column as the generator defines it: the fill reached the end
    want = (1 + synthetic.splitmix64(seed, 0, np.uint64(n - 1) * np.uint64(m) + np.arange(m, dtype=np.uint64)) % np.uint64(999)) / 1000.0
    assert np.array_equal(A[n - 1].cpu().numpy(), want.astype(np.float64))
    nums_b = n * (1000 + (synthetic.splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64))
    nums_c = -(1000 + (synthetic.splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64))
    md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=nums_b / 4000.0, cost=nums_c / 1000.0,
                    upper_bound=np.full(n, np.inf))
    t = engine.Tableau(md, engine=engine.ENGINE_TABLEAU, trace_capacity=4096, device_dense_ptr=A.data_ptr(), device_dense_ld=m)
    assert _run(t, 300) >= 200
    tr, obj = t.trace(), t.objective_function_value()
    t.close()
    r = engine.Tableau(md, engine=engine.ENGINE_REVISED, trace_capacity=4096, device_dense_ptr=A.data_ptr(), device_dense_ld=m)
    assert _run(r, 300) == len(tr)
    assert tr == r.trace()
    assert abs(obj - r.objective_function_value()) <= 1e-9 * abs(obj)
    r.close()


@pytest.mark.parametrize("name", ["test_no_change", "test_from_identity_2", "test_from_5x5_identity_no_r", "test_from_4x4_identity",
                                  "test_from_5x5_elble_sahinidis", "test_reference_cadence_refactors_after_the_eleventh_update"])
def test_change_basis_known_answers_in_layout_2(name, monkeypatch):
    """The reference's five `change_basis` cases (lower_upper/mod.rs:605-867) and the cadence test of tests/test_gpu_lu_update.py
    once more with the persistent kernel forced into layout 2 (the step-wise kernels k_ft_ftran / k_ft_btran / k_ft_update keep
    the sparse-vector invariant too)."""
    import test_gpu_lu_update as base
    monkeypatch.setenv("RELP_FT_BIG", "2")
    getattr(base, name)()


@pytest.mark.parametrize("m,seed", [(13, 3), (97, 5), (300, 6)])
def test_random_replacement_sequences_in_layout_2(m, seed, monkeypatch):
    import test_gpu_lu_update as base
    monkeypatch.setenv("RELP_FT_BIG", "2")
    base.test_random_replacement_sequences_against_a_dense_inverse(m, seed)


def test_row_removal_and_both_phases_in_layout_2(monkeypatch):
    """BORE3D (artificials left basic at zero level, redundant rows removed at the phase switch: the bitmaps saved between
    launches describe another m afterwards) and 50v-10 (upper bounds, 1,647 bound rows) to their pins in layout 2."""
    from lp_files import load
    monkeypatch.setenv("RELP_FT_BIG", "2")
    for path, fixed, pin, tol in (("netlib/BORE3D.SIF", True, 0.13730803942084927e4, 1e-2), ("miplib/50v-10.mps", False, 2879.065687, 1e-3)):
        gf, ex, md, emd = load(path, fixed=fixed)
        t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
        assert t.lu_kernel_layout()["layout"] == 2
        assert t.solve_relaxation() == engine.OPTIMAL
        assert abs(t.objective_function_value() + float(gf.fixed_cost) - pin) < tol, path
        ref = relp_f64.OracleF64(md)
        assert ref.run() == "optimal"
        same = sum(1 for a, b in zip(t.trace(), ref.trace) if a == b)
        assert same >= min(len(ref.trace), 300), (path, same)
