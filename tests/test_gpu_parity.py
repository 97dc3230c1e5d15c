"""Parity tests proper: the HIP engine (through the C ABI) against the oracles on the same seeded
inputs.  Index work (pivot sequence, basis) must be identical; f64 values within the tolerances
written here: |dobj|/|obj| <= 1e-9, |db|_inf <= 1e-7 * max(1, |b|_inf).
"""
import os

import numpy as np
import pytest

import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic
from oracle import relp_exact as ox
from oracle import relp_f64

pytestmark = pytest.mark.gpu

OBJ_RTOL = 1e-9
VEC_TOL = 1e-7


def assert_state_close(t, ref):
    assert abs(t.objective_function_value() - ref.objective) <= OBJ_RTOL * max(1.0, abs(ref.objective))
    bref = ref.b()
    assert np.max(np.abs(t.b() - bref)) <= VEC_TOL * max(1.0, np.max(np.abs(bref)))
    pref = ref.minus_pi()
    assert np.max(np.abs(t.minus_pi() - pref)) <= VEC_TOL * max(1.0, np.max(np.abs(pref)))
    assert t.basis_indices().tolist() == ref.basis().tolist()


def dense_problem(m, n, seed):
    lp = synthetic.dense_lp(m, n, seed)
    return MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])


# update_block: 0 = rank-1 update of the explicit inverse per pivot (reference-literal), K > 0 =
# deferred update folded in every K pivots (K = 3 forces frequent flushes and repeated pivot rows)
BLOCKS = [0, 3, 64]
# engine kinds: revised (explicit / deferred inverse) and the dense tableau (always blocked)
# and the sparse LU engine (block = pivots between refactorisations; 1 = refactor at every pivot)
KINDS = [(engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 3), (engine.ENGINE_REVISED, 64),
         (engine.ENGINE_TABLEAU, 3), (engine.ENGINE_TABLEAU, 64),
         (engine.ENGINE_LU, 1), (engine.ENGINE_LU, 3), (engine.ENGINE_LU, 64)]


@pytest.mark.parametrize("kind,block", KINDS)
@pytest.mark.parametrize("m,n,seed", [(8, 8, 1), (32, 48, 7), (128, 128, 20250001), (257, 131, 3), (300, 700, 11)])
def test_dense_trace_matches_f64_oracle(m, n, seed, kind, block):
    md = dense_problem(m, n, seed)
    t = engine.Tableau(md, trace_capacity=1 << 16, update_block=block, engine=kind)
    assert t.update_block() == block
    assert t.solve_relaxation() == engine.OPTIMAL
    ref = relp_f64.OracleF64(md.ensure_csc())
    assert ref.run() == "optimal"
    assert t.trace() == ref.trace
    assert_state_close(t, ref)
    ident, basic, min_b = t.check_basis()
    assert ident <= 1e-8 and basic <= 1e-8 and min_b >= -1e-9


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 16), (engine.ENGINE_TABLEAU, 16),
                                        (engine.ENGINE_LU, 11)])
@pytest.mark.parametrize("m,n,seed", [(8, 8, 1), (32, 48, 7), (128, 128, 20250001)])
def test_dense_trace_matches_exact_oracle(m, n, seed, kind, block):
    """Parity shadows of config C2: the f64 GPU pivot sequence equals the exact-rational trace."""
    md = dense_problem(m, n, seed)
    t = engine.Tableau(md, trace_capacity=1 << 16, update_block=block, engine=kind)
    assert t.solve_relaxation() == engine.OPTIMAL
    cols, b, c = synthetic.dense_lp_exact(m, n, seed)
    emd = ox.MatrixData(cols, b, [], 0, 0, m, 0, c, [None] * n)
    tr = []
    out = ox.solve_relaxation(emd, trace=tr.append)
    assert out["status"] == "optimal"
    assert t.trace() == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
    obj = float(out["objective"])
    assert abs(t.objective_function_value() - obj) <= OBJ_RTOL * max(1.0, abs(obj))
    bfs = dict(t.current_bfs())
    for j, v in out["bfs"]:
        assert abs(bfs[j] - float(v)) <= VEC_TOL * max(1.0, abs(float(v)))


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 5), (engine.ENGINE_REVISED, 64),
                                        (engine.ENGINE_TABLEAU, 5), (engine.ENGINE_TABLEAU, 64),
                                        (engine.ENGINE_LU, 1), (engine.ENGINE_LU, 5), (engine.ENGINE_LU, 64)])
@pytest.mark.parametrize("m,n,seed", [(20, 30, 5), (60, 90, 2), (150, 220, 9)])
def test_sparse_two_phase_matches_f64_oracle(m, n, seed, kind, block):
    """==, <=, >= rows and upper bounds: phase 1 (FirstProfitableWithMemory), the phase switch and
    phase 2 (SteepestDescent), CSC input."""
    md = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed))
    t = engine.Tableau(md, trace_capacity=1 << 16, update_block=block, engine=kind)
    outcome = t.solve_relaxation()
    ref = relp_f64.OracleF64(md)
    status = ref.run()
    assert engine.OUTCOME_NAMES[outcome] == status
    assert t.trace() == ref.trace
    if status == "optimal":
        assert_state_close(t, ref)


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 4), (engine.ENGINE_TABLEAU, 4),
                                        (engine.ENGINE_LU, 4)])
def test_stepwise_api_matches_loop(kind, block):
    """The step-by-step entry points (select column / generate column / select row / bring into
    basis) walk the same path as relp_run."""
    md = dense_problem(40, 60, 13)
    loop = engine.Tableau(md, trace_capacity=4096)
    assert loop.solve_relaxation() == engine.OPTIMAL
    t = engine.Tableau(md, trace_capacity=4096, update_block=block, engine=kind)
    assert t.run(0)[1] in (engine.RUNNING, engine.PHASE_ONE_DONE)
    if t.phase == 1:
        assert t.run(1 << 20)[1] == engine.PHASE_ONE_DONE
    steps = []
    while True:
        sel = t.select_primal_pivot_column(engine.STEEPEST_DESCENT)
        if sel is None:
            break
        q, dq = sel
        col = t.generate_column(q)
        r = t.select_primal_pivot_row()
        assert r is not None and col[r] > 0
        leaving = t.bring_into_basis(q, r, dq)
        steps.append((2, q, r, leaving))
    assert steps == loop.trace()
    assert abs(t.objective_function_value() - loop.objective_function_value()) <= 1e-12 * abs(loop.objective_function_value())


@pytest.mark.parametrize("kind", [engine.ENGINE_REVISED, engine.ENGINE_TABLEAU, engine.ENGINE_LU])
def test_relative_costs_and_generate_element(kind):
    md = dense_problem(24, 36, 21)
    t = engine.Tableau(md, update_block=4, engine=kind)
    t.run(1 << 20)          # finishes the (empty) phase 1
    t.run(5)
    ref = relp_f64.OracleF64(md.ensure_csc())
    ref.run(5)
    d = t.relative_costs()
    binv = ref.basis_inverse()
    a = np.hstack([md.dense, np.eye(24)])
    c = np.concatenate([md.cost, np.zeros(24)])
    expect = c + ref.minus_pi() @ a
    np.testing.assert_allclose(d, expect, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(t.basis_inverse(), binv, rtol=1e-9, atol=1e-12)
    col = binv @ a[:, 3]
    np.testing.assert_allclose(t.generate_column(3), col, rtol=1e-9, atol=1e-12)
    assert abs(t.generate_element(2, 3) - col[2]) <= 1e-9


def test_unbounded_and_infeasible_outcomes():
    # unbounded: min -x0 - x1  s.t.  x0 - x1 <= 1
    md = MatrixData(nr_normal=2, nr_eq=0, nr_range=0, nr_le=1, nr_ge=0, b=np.array([1.0]), cost=np.array([-1.0, -1.0]),
                    upper_bound=np.array([np.inf, np.inf]), dense=np.asfortranarray([[1.0, -1.0]]))
    assert engine.Tableau(md).solve_relaxation() == engine.UNBOUNDED
    # infeasible: x0 + x1 == 5, x0 + x1 <= 3
    md = MatrixData(nr_normal=2, nr_eq=1, nr_range=0, nr_le=1, nr_ge=0, b=np.array([5.0, 3.0]), cost=np.array([1.0, 1.0]),
                    upper_bound=np.array([np.inf, np.inf]), dense=np.asfortranarray([[1.0, 1.0], [1.0, 1.0]]))
    assert engine.Tableau(md).solve_relaxation() == engine.INFEASIBLE


def test_reference_problem_1_and_2_on_gpu():
    """src/tests/problem_1.rs / problem_2.rs (FirstProfitable in both phases): optimum pins."""
    cons = np.asfortranarray([[3.0, 2, 1, 0, 0], [5, 1, 1, 1, 0], [2, 5, 1, 0, 1]])
    md = MatrixData(nr_normal=5, nr_eq=3, nr_range=0, nr_le=0, nr_ge=0, b=np.array([1.0, 3, 4]), cost=np.ones(5),
                    upper_bound=np.full(5, np.inf), dense=cons)
    t = engine.Tableau(md, phase_one_rule=engine.FIRST_PROFITABLE, phase_two_rule=engine.FIRST_PROFITABLE)
    assert t.run(1 << 20)[1] == engine.PHASE_ONE_DONE
    # post-phase-1 carry, src/tests/problem_2.rs:141-174
    assert abs(t.objective_function_value() - 4.5) < 1e-12
    np.testing.assert_allclose(t.minus_pi(), [2.5, -1, -1], atol=1e-12)
    np.testing.assert_allclose(t.b(), [0.5, 2.5, 1.5], atol=1e-12)
    assert t.basis_indices().tolist() == [1, 3, 4]
    assert t.run(1 << 20)[1] == engine.OPTIMAL
    assert [(j, round(v, 12)) for j, v in t.current_bfs()] == [(1, 0.5), (3, 2.5), (4, 1.5)]

    cons = np.asfortranarray([[0.0, -1, 1], [1, 0, 1]])
    md = MatrixData(nr_normal=3, nr_eq=1, nr_range=0, nr_le=0, nr_ge=1, b=np.array([6.0, 10]), cost=np.array([1.0, 4, 9]),
                    upper_bound=np.array([4.0, 2, np.inf]), dense=cons)
    t = engine.Tableau(md, phase_one_rule=engine.FIRST_PROFITABLE, phase_two_rule=engine.FIRST_PROFITABLE)
    assert t.nr_rows() == 4 and t.nr_columns() == 2 + 6
    assert t.run(1 << 20)[1] == engine.PHASE_ONE_DONE
    # src/tests/problem_1.rs:403-431
    assert abs(t.objective_function_value() - 58) < 1e-12
    np.testing.assert_allclose(t.minus_pi(), [4, -13, 12, 0], atol=1e-12)
    np.testing.assert_allclose(t.b(), [6, 0, 4, 2], atol=1e-12)
    assert t.basis_indices().tolist() == [2, 1, 0, 5]
    np.testing.assert_allclose(t.basis_inverse(), [[0, 1, -1, 0], [-1, 1, -1, 0], [0, 0, 1, 0], [1, -1, 1, 1]], atol=1e-12)
    assert t.run(1 << 20)[1] == engine.OPTIMAL
    assert [(j, round(v, 12)) for j, v in t.current_bfs()] == [(0, 4.0), (2, 6.0), (5, 2.0)]


def test_device_resident_matrix_and_synth_fill():
    """The on-device synthetic fill equals the numpy generator, and an adopted device matrix
    (zero-copy) gives the same solve."""
    import ctypes as C
    lib = engine.load_library()
    m, n, seed = 64, 96, 99
    lp = synthetic.dense_lp(m, n, seed)
    ptr = C.c_void_p()
    assert lib.relp_device_alloc(C.byref(ptr), m * n * 8) == 0
    assert lib.relp_synth_fill_dense(ptr, m, m, n, seed, 0, None) == 0
    md = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])
    t_dev = engine.Tableau(md, device_dense_ptr=ptr.value, device_dense_ld=m, trace_capacity=4096)
    t_host = engine.Tableau(md, trace_capacity=4096)
    assert t_dev.solve_relaxation() == engine.OPTIMAL
    assert t_host.solve_relaxation() == engine.OPTIMAL
    assert t_dev.trace() == t_host.trace()
    assert t_dev.objective_function_value() == t_host.objective_function_value()
    t_dev.close()
    assert lib.relp_device_free(ptr) == 0


def test_dense_2000_properties():
    """Config C2 size (2,000 x 2,000): too slow for a full CPU solve in a unit test, so check a
    bounded prefix against the CPU oracle and size-independent invariants after more pivots."""
    m = n = 2000
    md = dense_problem(m, n, 20250001)
    t = engine.Tableau(md, trace_capacity=1 << 16)
    assert t.update_block() == 0                   # automatic choice at this size (64 from m = 4096)
    assert t.run(1 << 20)[1] == engine.PHASE_ONE_DONE
    done, outcome = t.run(60)
    assert done == 60 and outcome == engine.RUNNING
    ref = relp_f64.OracleF64(md.ensure_csc())
    ref.run(60)
    assert t.trace() == ref.trace
    assert_state_close(t, ref)
    obj_60 = t.objective_function_value()
    done, outcome = t.run(300)
    assert done == 300
    assert t.objective_function_value() <= obj_60 + 1e-9          # monotone objective
    b = t.b()
    assert b.min() >= -1e-8                                       # primal feasibility kept
    basis = t.basis_indices()
    assert len(set(basis.tolist())) == m                          # a basis
    d = t.relative_costs()
    assert np.max(np.abs(d[basis])) <= 1e-7                       # basic reduced costs vanish
    # B^-1 B = I on a sample of basis columns
    a = np.hstack([md.dense, np.eye(m)])
    binv = t.basis_inverse()
    for i in (0, 17, 999, 1999):
        e = binv @ a[:, basis[i]]
        e[i] -= 1.0
        assert np.max(np.abs(e)) <= 1e-8


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 4), (engine.ENGINE_TABLEAU, 4),
                                        (engine.ENGINE_TABLEAU, 64)])
@pytest.mark.parametrize("world,m,n,seed", [(2, 48, 64, 17), (3, 61, 45, 23), (8, 40, 300, 31)])
def test_shard_entry_points_on_one_gpu(world, m, n, seed, kind, block):
    """`relp_shard_*` with G engines in one process on one GPU: the exchange steps (all-gather of
    candidates; for the revised engine also the all-gather of alpha slices and the SUM all-reduce of
    rho) are done with torch ops on a shared stream.  Every shard must walk the single-engine pivot
    sequence."""
    import torch
    from rust_lp_amd.sharded import HipShardOps
    lp = synthetic.dense_lp(m, n, seed)
    full = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])
    single = engine.Tableau(full, trace_capacity=4096)
    assert single.solve_relaxation() == engine.OPTIMAL
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    tabs, ops = [], []
    covered = []
    for r in range(world):
        counts = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=lp["b"], cost=lp["c"],
                            upper_bound=np.full(n, np.inf))
        cfg = engine.default_config(shard_rank=r, shard_count=world, engine=kind, update_block=block, trace_capacity=4096)
        lo, hi = engine.shard_plan(counts, cfg)
        covered.append((lo, hi))
        counts.dense = np.asfortranarray(lp["A"][:, lo:hi]) if hi > lo else np.zeros((m, 1), order="F")
        t = engine.Tableau(counts, config=cfg)
        t.set_stream(stream)
        tabs.append(t)
        ops.append(HipShardOps(t))
    assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    L, S, RL = ops[0].candidate_len, ops[0].row_stride, ops[0].rho_len
    cands = torch.zeros(world * L, dtype=torch.float64, device=dev)
    slices = torch.zeros(world * S, dtype=torch.float64, device=dev)
    rhos = torch.zeros((world, RL), dtype=torch.float64, device=dev)

    def iteration():
        for r, o in enumerate(ops):
            o.price(cands[r * L:(r + 1) * L])
        for o in ops:
            o.select_column(cands, world)
        if kind == engine.ENGINE_TABLEAU:
            for o in ops:
                o.pivot()                          # one collective per pivot: nothing else to exchange
            return
        for r, o in enumerate(ops):
            o.ftran(slices[r * S:(r + 1) * S])
        for r, o in enumerate(ops):
            o.ratio(slices, world, rhos[r])
        rho = rhos.sum(dim=0)
        for o in ops:
            o.update(rho)

    def flush():
        if kind == engine.ENGINE_TABLEAU:
            return                                 # the tableau flush is local and done inside relp_shard_pivot
        snaps = [o.flush_begin(torch, dev) for o in ops]
        if snaps[0] is None:
            return
        total = torch.stack(snaps).sum(dim=0)
        for sn, o in zip(snaps, ops):
            sn.copy_(total)
            o.flush_end()

    iteration()                                   # phase 1: no candidate anywhere
    assert [o.poll()[0] for o in ops] == [engine.PHASE_ONE_DONE] * world
    for k in range(len(single.trace()) + 3):
        iteration()
        if block and (k + 1) % block == 0:
            flush()
    torch.cuda.synchronize()
    for o, t in zip(ops, tabs):
        oc, it = o.poll()
        assert oc == engine.OPTIMAL and it == len(single.trace())
        assert t.trace() == single.trace()
        assert abs(t.objective_function_value() - single.objective_function_value()) <= 1e-9 * abs(single.objective_function_value())
        np.testing.assert_allclose(t.b(), single.b(), rtol=1e-9, atol=1e-9)


def _native_ranks(world, make_md, kind, block, poll_interval=16, return_calls=False, limit=None):
    """G engines on one GPU, each driven by `relp_shard_run` on its own thread with the in-process collectives
    of tests/shard_threads.py; returns [(outcome of phase 1, pivots, outcome, trace, objective, b)] per rank."""
    import ctypes as C
    import torch
    from shard_threads import ThreadRank, ThreadWorld, run_ranks
    lib = engine.load_library()
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)                          # initialise torch's HIP state on the main thread
    torch.cuda.synchronize()
    shared = ThreadWorld(world)
    tabs, ranks = [], []
    for r in range(world):
        cfg = engine.default_config(shard_rank=r, shard_count=world, engine=kind, update_block=block, trace_capacity=1 << 14,
                                    poll_interval=poll_interval)
        made = make_md(cfg)                              # MatrixData, or (MatrixData, Tableau keyword arguments)
        md_r, kw_r = made if isinstance(made, tuple) else (made, {})
        t = engine.Tableau(md_r, config=cfg, **kw_r)
        tabs.append(t)
        ranks.append(ThreadRank(shared, r, lib, t.handle, torch, dev))

    def body(r):
        t = tabs[r]
        done, oc = C.c_int64(), C.c_int32()
        if limit is not None:                              # a bounded number of pivots of the current phase
            assert lib.relp_shard_run(t.handle, limit, C.byref(done), C.byref(oc)) == 0, (lib.relp_last_error(t.handle).decode(), shared.errors)
            return oc.value, done.value, oc.value, t.trace(), t.objective_function_value(), t.b()
        assert lib.relp_shard_run(t.handle, 1 << 20, C.byref(done), C.byref(oc)) == 0, (lib.relp_last_error(t.handle).decode(), shared.errors)
        first = oc.value
        total = done.value
        if first == engine.PHASE_ONE_DONE:
            assert lib.relp_shard_run(t.handle, 1 << 20, C.byref(done), C.byref(oc)) == 0, lib.relp_last_error(t.handle).decode()
            total += done.value
        return first, total, oc.value, t.trace(), t.objective_function_value(), t.b()
    out = run_ranks(world, body)
    if return_calls:
        out = [o + (rk.calls["allgather"],) for o, rk in zip(out, ranks)]
    assert not shared.errors, shared.errors
    assert all(rk.calls["allgather"] > 0 for rk in ranks)
    for t in tabs:
        t.close()
    return out


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_TABLEAU, 64), (engine.ENGINE_TABLEAU, 4), (engine.ENGINE_REVISED, 0),
                                        (engine.ENGINE_REVISED, 4)])
@pytest.mark.parametrize("world,m,n,seed", [(2, 48, 64, 17), (3, 61, 45, 23), (5, 90, 140, 3)])
def test_native_shard_loop_with_several_ranks_on_one_gpu(world, m, n, seed, kind, block):
    """The loop inside the library (`relp_shard_run`) with G > 1: one thread per rank, collectives through the
    `relp_shard_set_collectives` hooks (tests/shard_threads.py stands in for RCCL, which refuses two ranks on one
    device).  Every rank walks the single-engine pivot sequence."""
    import torch  # noqa: F401  (before the first engine: PyTorch's HIP runtime has to be the one in the process)
    lp = synthetic.dense_lp(m, n, seed)
    single = engine.Tableau(MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]), trace_capacity=1 << 14)
    assert single.solve_relaxation() == engine.OPTIMAL

    def make_md(cfg):
        md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=lp["b"], cost=lp["c"], upper_bound=np.full(n, np.inf))
        lo, hi = engine.shard_plan(md, cfg)
        md.dense = np.asfortranarray(lp["A"][:, lo:hi]) if hi > lo else np.zeros((m, 1), order="F")
        return md
    for first, total, oc, trace, obj, b in _native_ranks(world, make_md, kind, block):
        assert first == engine.PHASE_ONE_DONE and oc == engine.OPTIMAL
        assert trace == single.trace() and total == len(trace)
        assert abs(obj - single.objective_function_value()) <= 1e-9 * abs(single.objective_function_value())
        np.testing.assert_allclose(b, single.b(), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_TABLEAU, 8), (engine.ENGINE_REVISED, 4)])
def test_native_shard_loop_agrees_on_a_failure_of_one_rank(kind, block):
    """A rank whose step fails in the middle of a chunk (relp_shard_inject_failure: after 5 pivots, polls every 8) must not
    leave the others inside a collective: it keeps the chunk's collectives going, and at the poll one all-gather of the
    statuses makes EVERY rank return an error."""
    import ctypes as C
    import torch
    from shard_threads import ThreadRank, ThreadWorld, run_ranks
    lib = engine.load_library()
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    world, m, n = 3, 60, 90
    lp = synthetic.dense_lp(m, n, 41)
    shared = ThreadWorld(world)
    tabs, ranks = [], []
    for r in range(world):
        cfg = engine.default_config(shard_rank=r, shard_count=world, engine=kind, update_block=block, poll_interval=8)
        md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=lp["b"], cost=lp["c"], upper_bound=np.full(n, np.inf))
        lo, hi = engine.shard_plan(md, cfg)
        md.dense = np.asfortranarray(lp["A"][:, lo:hi]) if hi > lo else np.zeros((m, 1), order="F")
        t = engine.Tableau(md, config=cfg)
        tabs.append(t)
        ranks.append(ThreadRank(shared, r, lib, t.handle, torch, dev))

    def body(r):
        t = tabs[r]
        done, oc = C.c_int64(), C.c_int32()
        assert lib.relp_shard_run(t.handle, 1, C.byref(done), C.byref(oc)) == 0           # empty phase 1
        if r == 1:
            assert lib.relp_shard_inject_failure(t.handle, 5) == 0
        st = lib.relp_shard_run(t.handle, 1 << 20, C.byref(done), C.byref(oc))
        return st, lib.relp_last_error(t.handle).decode()
    out = run_ranks(world, body)
    assert not shared.errors, shared.errors
    assert all(st != 0 for st, _ in out), out                     # everybody stops, nobody hangs
    assert "injected failure" in out[1][1] and all("rank 1" in msg for k, (st, msg) in enumerate(out) if k != 1), out
    for t in tabs:
        t.close()


def test_sharded_tableau_runs_both_phases_on_general_lps():
    """LPs with ==, >= rows and bounded variables (artificial variables, phase 1, zero-level pivots that remove
    basic artificials, phase switch) on the column-sharded tableau engine, 2 to 4 ranks through the native loop:
    every rank ends with the single-engine outcome, pivot sequence and objective, rank-deficient cases (redundant
    rows removed from every rank's columns at the phase switch) included."""
    import ctypes as C
    import torch  # noqa: F401
    rng = np.random.default_rng(77)
    checked = removed = 0
    outcomes = set()
    for case in range(48):
        m, n = int(rng.integers(8, 60)), int(rng.integers(6, 90))
        if case % 4 == 3:                                      # every row kind incl. ranges; some infeasible
            d = synthetic.mixed_lp(m, n, 8100 + case, nnz_per_col=int(rng.integers(2, 6)), frac_negative_cost=0.1,
                                   infeasible=(case % 8 == 7))
        else:
            d = synthetic.sparse_lp(m, n, 8100 + case, nnz_per_col=int(rng.integers(2, 6)), frac_eq=float(rng.uniform(0.1, 0.5)),
                                    frac_ge=float(rng.uniform(0, 0.4)), frac_bounded=float(rng.uniform(0, 0.5)))
            d.setdefault("ranges", np.zeros(0))
            if case % 6 == 5:
                d["c"] = -d["c"]                              # unbounded / other ends
        full = MatrixData.from_sparse_dict(d)
        ref = relp_f64.OracleF64(full)
        status = ref.run(200000)
        single = engine.Tableau(full, engine=engine.ENGINE_TABLEAU, update_block=4, trace_capacity=1 << 14)
        single_outcome = single.solve_relaxation()
        dense = np.zeros((m, n))
        for j in range(n):
            for e in range(d["col_ptr"][j], d["col_ptr"][j + 1]):
                dense[d["row_idx"][e], j] = d["values"][e]
        world = int(rng.integers(2, 5))

        def make_md(cfg, d=d, dense=dense, m=m, n=n):
            md = MatrixData(nr_normal=n, nr_eq=d["nr_eq"], nr_range=d["nr_range"], nr_le=d["nr_le"], nr_ge=d["nr_ge"], b=d["b"],
                            cost=d["c"], upper_bound=d["ub"], ranges=d["ranges"])
            lo, hi = engine.shard_plan(md, cfg)
            md.dense = np.asfortranarray(dense[:, lo:hi]) if hi > lo else np.zeros((m, 1), order="F")
            return md
        # rows removed at the phase switch: when one of them is not an == / range row the reference deletes a
        # non-redundant row (see test_random_mixed_lps_...): defined by the tableau engine itself only
        rows = ref.filtered_rows()
        quirk = any(r >= d["nr_eq"] + d["nr_range"] for r in rows)
        results = _native_ranks(world, make_md, engine.ENGINE_TABLEAU, int(rng.choice([2, 5, 64])), poll_interval=int(rng.choice([3, 16])))
        for first, total, oc, trace, obj, b in results:
            assert (oc if first == engine.PHASE_ONE_DONE else first) == single_outcome, case
            assert trace == single.trace(), case
            if single_outcome == engine.OPTIMAL:
                assert abs(obj - single.objective_function_value()) <= OBJ_RTOL * max(1.0, abs(obj)), case
                np.testing.assert_allclose(b, single.b(), rtol=1e-9, atol=1e-9)
            if not quirk:
                assert engine.OUTCOME_NAMES[single_outcome] == status and trace == ref.trace, case
                if status == "optimal":
                    assert abs(obj - ref.objective) <= OBJ_RTOL * max(1.0, abs(ref.objective)), case
                    assert np.max(np.abs(b - ref.b())) <= VEC_TOL * max(1.0, np.max(np.abs(ref.b()))), case
        outcomes.add(status)
        removed += bool(rows)
        checked += 1
        single.close()
    assert checked == 48 and removed >= 5 and {"optimal", "infeasible", "unbounded"} <= outcomes, (checked, removed, outcomes)


@pytest.mark.parametrize("path,fixed", [("netlib/BOEING2.SIF", True), ("netlib/BORE3D.SIF", True)])
def test_python_sharded_loop_two_processes_general_lp(path, fixed):
    """`ShardedPivotLoop` (collectives from Python through torch.distributed, gloo) in two processes that share the
    GPU, on files of the reference with ==, range and >= rows: phase 1, the removal of basic artificial variables
    (BOEING2: the library calls back into torch.distributed through the collective hooks), redundant rows removed
    (BORE3D: the message buffers shrink with the rows), phase switch, phase 2."""
    import torch.multiprocessing as mp
    from lp_files import load
    from shard_gloo_worker import gloo_rank
    gf, ex, md, emd = load(path, fixed=fixed)
    ref = relp_f64.OracleF64(md)
    assert ref.run(400000) == "optimal"
    assert ref.nr_zero_level_pivots > 0 or ref.filtered_rows()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 300)
    procs = [ctx.Process(target=gloo_rank, args=(r, world, port, path, fixed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, done, oc, trace, obj, hook_calls in results:
        assert (hook_calls > 0) == (ref.nr_zero_level_pivots > 0 or bool(ref.filtered_rows()))
        assert oc == engine.OPTIMAL and trace == ref.trace and done == len(trace)
        assert abs(obj - ref.objective) <= OBJ_RTOL * max(1.0, abs(ref.objective))


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_TABLEAU, 64), (engine.ENGINE_TABLEAU, 5), (engine.ENGINE_REVISED, 0),
                                        (engine.ENGINE_REVISED, 4)])
def test_native_shard_loop_with_rccl_on_one_rank(kind, block):
    """`relp_shard_run` (the loop inside the library, RCCL called from C++ between the kernels) on a
    communicator of one rank: same pivot sequence as the unsharded engine.  The exchange logic for G > 1 is the
    same shard entry points the test above drives with G engines; this one covers the native driver and the
    RCCL adaptor (`relp_rccl_unique_id`, `relp_rccl_attach`)."""
    import ctypes as C
    import torch  # noqa: F401  (PyTorch's librccl.so.1 is then in the process; the adaptor takes that one)
    m, n, seed = 96, 130, 29
    lp = synthetic.dense_lp(m, n, seed)
    full = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])
    single = engine.Tableau(full, trace_capacity=4096)
    assert single.solve_relaxation() == engine.OPTIMAL
    lib = engine.load_library()
    cfg = engine.default_config(shard_rank=0, shard_count=1, engine=kind, update_block=block, trace_capacity=4096,
                                poll_interval=16)
    t = engine.Tableau(MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]), config=cfg)
    done, oc = C.c_int64(), C.c_int32()
    assert lib.relp_shard_run(t.handle, 1, C.byref(done), C.byref(oc)) == engine.E_STATE     # no collectives yet
    ident = (C.c_uint8 * 128)()
    assert lib.relp_rccl_unique_id(ident) == 0
    assert lib.relp_rccl_attach(t.handle, ident) == 0, lib.relp_last_error(t.handle).decode()
    assert lib.relp_shard_run(t.handle, 1, C.byref(done), C.byref(oc)) == 0
    assert oc.value == engine.PHASE_ONE_DONE and done.value == 0
    assert lib.relp_shard_run(t.handle, 7, C.byref(done), C.byref(oc)) == 0                  # bounded call
    assert oc.value == engine.RUNNING and done.value == 7
    assert lib.relp_shard_run(t.handle, 1 << 20, C.byref(done), C.byref(oc)) == 0
    assert oc.value == engine.OPTIMAL and done.value == len(single.trace()) - 7
    assert t.trace() == single.trace()
    assert abs(t.objective_function_value() - single.objective_function_value()) <= 1e-9 * abs(single.objective_function_value())
    np.testing.assert_allclose(t.b(), single.b(), rtol=1e-9, atol=1e-9)
    t.close()


# ------------------------------------------------------------------------------------------------
# File-driven configs (MPS reader + standardisation feed the engine)
# ------------------------------------------------------------------------------------------------
def test_adlittle_gpu_pivot_sequence_equals_exact_trace():
    """Config C1: burkardt adlittle.mps.  The f64 GPU engine walks the exact-rational pivot sequence
    (127 pivots, 2 phases) and lands on the reference's pinned optimum
    24975305659811992079614961229/120651674036153428931840 (tests/burkardt/test.rs:34-54)."""
    from fractions import Fraction as Fr
    from lp_files import exact_solve, load
    gf, ex, md, emd = load("burkardt/adlittle.mps")
    tr = []
    status, obj, sol = exact_solve(gf, emd, trace=tr.append)
    assert status == "optimal" and obj == Fr(24975305659811992079614961229, 120651674036153428931840)
    for kind, block in ((engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 8), (engine.ENGINE_TABLEAU, 8),
                        (engine.ENGINE_LU, 11)):
        t = engine.Tableau(md, trace_capacity=4096, update_block=block, engine=kind)
        assert t.solve_relaxation() == engine.OPTIMAL
        assert t.trace() == [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
        got = t.objective_function_value() + float(gf.fixed_cost)
        assert abs(got - float(obj)) <= 1e-9 * float(obj)
        reduced = {j: v for j, v in t.current_bfs() if j < md.nr_normal}
        _, x = gf.compute_full_solution(reduced)
        for name, v in sol.items():
            assert abs(x[name] - float(v)) <= 1e-7 * max(1.0, abs(float(v)))


FILES = [("burkardt/afiro.mps", False, -406659 / 875, 1e-9), ("burkardt/testprob.mps", False, 54.0, 1e-9),
         ("burkardt/maros.mps", False, 385 / 3, 1e-9), ("cook/small_example.mps", False, -243 / 4, 1e-9),
         ("netlib/AFIRO.SIF", True, -464.75314, 1e-3), ("netlib/SC50A.SIF", True, -6.457507706e+01, 1e-5),
         ("netlib/SC50B.SIF", True, -70.0, 1e-9), ("netlib/KB2.SIF", True, -1.749900130e+03, 1e-3),
         ("netlib/SC105.SIF", True, -5.220206121e+01, 1e-3), ("netlib/ADLITTLE.SIF", True, 2.254949632e+05, 1e-3),
         ("netlib/STOCFOR1.SIF", True, -4.113197622e+04, 1e-3), ("netlib/BLEND.SIF", True, -30.81215, 1e-3),
         ("netlib/SCAGR7.SIF", True, -2.331389824e+06, 1e-1), ("netlib/SC205.SIF", True, -5.220206121e+01, 1e-5),
         ("netlib/SHARE2B.SIF", True, -4.157322407e+02, 1e-5), ("netlib/RECIPELP.SIF", True, -0.266616e3, 1e-2),
         ("netlib/LOTFI.SIF", True, -0.2526470606188e2, 1e-6), ("netlib/VTP-BASE.SIF", True, 0.1298314624613613e6, 1e-2),
         ("netlib/SHARE1B.SIF", True, -0.76589318579185e5, 1e-3), ("netlib/BORE3D.SIF", True, 0.13730803942084927e4, 1e-2),
         ("netlib/BOEING2.SIF", True, -0.31501872801520287e3, 1e-3),
         ("miplib/50v-10.mps", False, 2879.065687, 1e-3)]


# (Round 1 exempted LOTFI, SHARE1B and BORE3D on the LU engine from the trace comparison: its product-form updates rounded
# differently from the oracle on bases with cond(B) ~ 1e6 .. 1e8.  With the Forrest-Tomlin update and the 1e-5 pivot
# tolerance all three walk the oracle's path, at the reference's refactorisation cadence and at the engine's own.)


@pytest.mark.parametrize("path,fixed,objective,tol,world", [
    ("netlib/BOEING2.SIF", True, -0.31501872801520287e3, 1e-3, 3),     # a zero-level pivot; an artificial survives it
    ("miplib/50v-10.mps", False, 2879.065687, 1e-3, 4),                 # config C5: 1,647 bound rows, one redundant row
    ("netlib/BORE3D.SIF", True, 0.13730803942084927e4, 1e-2, 2),        # two redundant rows
    ("burkardt/adlittle.mps", False, 24975305659811992079614961229 / 120651674036153428931840, 1e-9, 8)])
def test_reference_problem_files_on_the_sharded_tableau_engine(path, fixed, objective, tol, world):
    """BASELINE config C5 (MIPLIB relaxation with sharded pricing) and other files of the reference on the
    column-sharded tableau engine, through the native loop: both phases, the removal of basic artificial variables
    through the candidate exchange, redundant rows deleted on every rank.  Pivot sequence of the f64 oracle, the
    reference's objective pin."""
    import torch  # noqa: F401
    from lp_files import load
    gf, ex, md, emd = load(path, fixed=fixed)
    ref = relp_f64.OracleF64(md)
    assert ref.run() == "optimal"
    full = md.ensure_dense()
    dense = np.array(full.dense)

    def make_md(cfg):
        part = MatrixData(nr_normal=md.nr_normal, nr_eq=md.nr_eq, nr_range=md.nr_range, nr_le=md.nr_le, nr_ge=md.nr_ge, b=md.b,
                          cost=md.cost, upper_bound=md.upper_bound, ranges=md.ranges)
        lo, hi = engine.shard_plan(part, cfg)
        part.dense = np.asfortranarray(dense[:, lo:hi]) if hi > lo else np.zeros((dense.shape[0], 1), order="F")
        return part
    for first, total, oc, trace, obj, b in _native_ranks(world, make_md, engine.ENGINE_TABLEAU, 32):
        assert first == engine.PHASE_ONE_DONE and oc == engine.OPTIMAL
        assert trace == ref.trace and total == len(trace)
        got = obj + float(gf.fixed_cost)
        assert abs(got - objective) < max(tol, 1e-9 * abs(objective))


@pytest.mark.parametrize("kind", [engine.ENGINE_REVISED, engine.ENGINE_TABLEAU, engine.ENGINE_LU])
@pytest.mark.parametrize("path,fixed,objective,tol", FILES)
def test_reference_problem_files_on_gpu(path, fixed, objective, tol, kind):
    """The reference's own problem files (tests/{burkardt,cook,netlib,miplib}) through the GPU engine:
    the pivot sequence equals the f64 CPU oracle's and the objective meets the reference's pin with the
    reference's tolerance."""
    from lp_files import load
    gf, ex, md, emd = load(path, fixed=fixed)
    t = engine.Tableau(md, trace_capacity=1 << 15, engine=kind, update_block={engine.ENGINE_TABLEAU: 32, engine.ENGINE_LU: 11}.get(kind, -1))
    outcome = t.solve_relaxation()
    assert outcome == engine.OPTIMAL
    ref = relp_f64.OracleF64(md)
    assert ref.run() == "optimal"
    assert t.trace() == ref.trace
    got = t.objective_function_value() + float(gf.fixed_cost)
    assert abs(got - objective) < max(tol, 1e-9 * abs(objective))
    assert abs(got - (ref.objective + float(gf.fixed_cost))) <= 1e-9 * max(1.0, abs(objective))
    # primal feasibility of the final b (the ratio test clamps b_i <= tol_zero to 0, so a b that rounding or a skipped
    # small pivot pushed below zero would be hidden from later pivots; nothing may be left of that at the end)
    b = t.b()
    assert b.min() >= -1e-7 * max(1.0, np.abs(b).max())


def test_nazareth_is_unbounded_on_gpu():
    from lp_files import load
    gf, ex, md, emd = load("burkardt/nazareth.mps")
    assert engine.Tableau(md).solve_relaxation() == engine.UNBOUNDED           # tests/burkardt/test.rs:149-157


def test_c2_full_solve_both_engines_agree_to_optimality():
    """Config C2 (2,000 x 2,000) solved to optimality (~5,900 pivots) by both engines: identical pivot
    sequences over the whole solve, and the final basis checked against a numpy solve:
    B x_B = rhs, x_B >= 0, basic reduced costs vanish, no negative reduced cost is left."""
    m = n = 2000
    md = dense_problem(m, n, 20250001)
    traces, objs = [], []
    for kind in (engine.ENGINE_TABLEAU, engine.ENGINE_REVISED):
        t = engine.Tableau(md, engine=kind, trace_capacity=1 << 16)
        assert t.solve_relaxation() == engine.OPTIMAL
        basis, b, d = t.basis_indices(), t.b(), t.relative_costs()
        a = np.hstack([md.dense, np.eye(m)])
        x_b = np.linalg.solve(a[:, basis], md.b)
        assert np.max(np.abs(x_b - b)) <= 1e-7 * max(1.0, np.max(np.abs(x_b)))
        assert b.min() >= -1e-9
        nonbasic = np.ones(len(d), dtype=bool)
        nonbasic[basis] = False
        assert d[nonbasic].min() >= -1e-9 and np.max(np.abs(d[basis])) <= 1e-9
        c = np.concatenate([md.cost, np.zeros(m)])
        assert abs(t.objective_function_value() - c[basis] @ x_b) <= 1e-9 * abs(c[basis] @ x_b)
        traces.append(t.trace())
        objs.append(t.objective_function_value())
    assert traces[0] == traces[1] and len(traces[0]) > 1000
    assert abs(objs[0] - objs[1]) <= 1e-11 * abs(objs[0])


# ------------------------------------------------------------------------------------------------
# Sparse LU engine (Carry<_, LUDecomposition<_>>)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", [engine.ENGINE_LU, engine.ENGINE_REVISED, engine.ENGINE_TABLEAU])
@pytest.mark.parametrize("path,fixed", [("burkardt/adlittle.mps", False), ("netlib/SC205.SIF", True), ("netlib/SHARE1B.SIF", True)])
def test_from_basis_restarts_at_the_optimum(path, fixed, kind):
    """InverseMaintener::from_basis (carry/mod.rs:428-463) with a general basis: the LU engine factorises the
    optimal basis found by a cold solve (the revised engine inverts it on the host) and reports optimality
    without a single pivot, with the same b, -pi and objective."""
    from lp_files import load
    gf, ex, md, emd = load(path, fixed=fixed)
    first = engine.Tableau(md)
    assert first.solve_relaxation() == engine.OPTIMAL
    basis = first.basis_indices()
    t = engine.Tableau(md, engine=kind, trace_capacity=64)
    if t.nr_rows() != first.nr_rows():
        pytest.skip("phase 1 removed redundant rows: the basis refers to the reduced problem")
    t.from_basis(basis)
    done, outcome = t.run(1 << 20)
    assert (done, outcome) == (0, engine.OPTIMAL)
    assert abs(t.objective_function_value() - first.objective_function_value()) <= 1e-9 * max(1.0, abs(first.objective_function_value()))
    np.testing.assert_allclose(t.b(), first.b(), rtol=1e-7, atol=1e-7 * max(1.0, np.max(np.abs(first.b()))))
    scale = max(1.0, np.max(np.abs(first.minus_pi())))
    np.testing.assert_allclose(t.minus_pi(), first.minus_pi(), rtol=1e-7, atol=1e-7 * scale)


@pytest.mark.parametrize("kind", [engine.ENGINE_LU, engine.ENGINE_REVISED, engine.ENGINE_TABLEAU])
def test_from_basis_mid_solve_continues_the_same_path(kind):
    """Warm start from an intermediate basis: the continuation walks the same pivots as the run it was
    taken from (Dantzig's rule has no memory)."""
    md = MatrixData.from_sparse_dict(synthetic.sparse_lp(60, 90, 2))
    full = engine.Tableau(md, engine=engine.ENGINE_LU, update_block=7, trace_capacity=4096)
    assert full.run(1 << 20)[1] == engine.PHASE_ONE_DONE
    start = len(full.trace())
    assert full.run(10)[0] == 10
    if full.nr_rows() != engine.Tableau(md, engine=engine.ENGINE_LU).nr_rows():
        pytest.skip("rows were removed at the phase switch")
    basis = full.basis_indices()
    assert full.run(1 << 20)[1] == engine.OPTIMAL
    t = engine.Tableau(md, engine=kind, update_block=7, trace_capacity=4096)
    t.from_basis(basis)
    assert t.run(1 << 20)[1] == engine.OPTIMAL
    assert t.trace() == full.trace()[start + 10:]
    assert abs(t.objective_function_value() - full.objective_function_value()) <= 1e-9 * max(1.0, abs(full.objective_function_value()))


def test_lu_engine_degenerate_sparse_reaches_the_oracle_optimum():
    """A Netlib-shaped sparse LP (400 x 800, ~11,000 heavily degenerate pivots, hundreds of
    refactorisations): the first 500 pivots equal the CPU oracle's, and the solve ends at the oracle's
    optimum.  (Over 11,000 degenerate pivots f64 rounding may legitimately break a tie differently, so
    the whole trace is not compared; the optimum is unique.)"""
    md = MatrixData.from_sparse_dict(synthetic.sparse_lp(400, 800, 77))
    ref = relp_f64.OracleF64(md)
    assert ref.run() == "optimal"
    t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=1 << 16)
    assert t.solve_relaxation() == engine.OPTIMAL
    assert t.trace()[:500] == ref.trace[:500]
    assert abs(t.objective_function_value() - ref.objective) <= 1e-7 * abs(ref.objective)
    ident, basic, min_b = t.check_basis()
    # -pi is updated pivot by pivot like the reference's (carry/mod.rs:559-563), never recomputed: 11,000
    # updates leave ~1e-7 of drift in the basic reduced costs
    assert ident <= 1e-7 and basic <= 1e-6 and min_b >= -1e-7
    st = t.lu_stats()
    assert st["refactorisations"] > 50 and st["m"] == t.nr_rows()


def test_acc_tight4_degenerate_stress_prefix_matches_the_oracle():
    """BASELINE config C5's degenerate stress: MIPLIB acc-tight4 (3,285 constraints + 1,620 bound rows, all columns
    bounded; tests/miplib/test.rs:14-18, `#[ignore = "Too computationally expensive"]` in the reference: phase 1 alone
    needs far more than 400,000 pivots).  No optimum to pin, so the check is the path: the first 1,500 pivots of every
    engine equal the f64 oracle's, ties and all (more than half of them have ratio exactly 0), and the phase-1
    objective never increases over 6,000 pivots."""
    from lp_files import load
    gf, ex, md, emd = load("miplib/acc-tight4.mps", fixed=False)
    assert (md.nr_eq, md.nr_le, md.nr_ge, md.nr_normal) == (297, 756, 2232, 1620)
    ref = relp_f64.OracleF64(md)
    assert ref.run(1500) == "iteration_limit"
    degenerate = []
    for kind, block in ((engine.ENGINE_TABLEAU, 64), (engine.ENGINE_REVISED, -1), (engine.ENGINE_LU, -1)):
        t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 13)
        objs = []
        for k in range(12):
            done, oc = t.run(500)
            assert done == 500 and oc == engine.RUNNING
            objs.append(t.objective_function_value())
            if k == 2:
                degenerate.append(t.degenerate_pivots())     # after the 1,500 pivots every engine shares
        assert t.trace()[:1500] == ref.trace
        assert all(b <= a + 1e-7 for a, b in zip(objs, objs[1:])) and objs[-1] < objs[0]
        t.close()
    # the count SURVEY 8d asks for: pivots with ratio exactly 0 -- the same on every engine, and most of them
    assert degenerate[0] == degenerate[1] == degenerate[2] and degenerate[0] > 750, degenerate


def test_acc_tight4_on_the_sharded_tableau_engine():
    """Config C5 as BASELINE.json words it -- MIPLIB relaxation, degenerate pivoting stress, sharded pricing: the
    first 1,000 phase-1 pivots of acc-tight4 on 4 ranks (native loop) are the f64 oracle's on every rank."""
    import torch  # noqa: F401
    from lp_files import load
    gf, ex, md, emd = load("miplib/acc-tight4.mps", fixed=False)
    ref = relp_f64.OracleF64(md)
    assert ref.run(1000) == "iteration_limit"
    dense = np.array(md.ensure_dense().dense)

    def make_md(cfg):
        part = MatrixData(nr_normal=md.nr_normal, nr_eq=md.nr_eq, nr_range=md.nr_range, nr_le=md.nr_le, nr_ge=md.nr_ge, b=md.b,
                          cost=md.cost, upper_bound=md.upper_bound, ranges=md.ranges)
        lo, hi = engine.shard_plan(part, cfg)
        part.dense = np.asfortranarray(dense[:, lo:hi]) if hi > lo else np.zeros((dense.shape[0], 1), order="F")
        return part
    for first, total, oc, trace, obj, b in _native_ranks(4, make_md, engine.ENGINE_TABLEAU, 64, poll_interval=100, limit=1000):
        assert oc == engine.RUNNING and total == 1000
        assert trace[:1000] == ref.trace


# ------------------------------------------------------------------------------------------------
# Config C3: Netlib 25FV47 (BASELINE.json configs[2])
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,block", [(engine.ENGINE_LU, -1), (engine.ENGINE_REVISED, 0), (engine.ENGINE_TABLEAU, 32)])
def test_25fv47_reaches_the_netlib_optimum_under_the_default_config(kind, block):
    """25FV47 after presolve: m = 790 rows, 1,539 structural columns, ~12,000 pivots over two phases, solved with
    `relp_default_config` (no hand-set tolerances) by all three engines: the LU engine (the configuration
    BASELINE.json names: eta-file basis maintenance), the explicit-inverse engine and the dense tableau reach the
    optimum the reference pins (tests/netlib/test.rs:152-158: 5.5018459e+03, given to 8 digits; the reference is
    exact and `#[ignore]`s this file as too expensive)."""
    from lp_files import load
    gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
    t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 15)
    assert t.config.tol_pivot == 1e-5 and t.config.tol_cost == 1e-7            # relp_default_config
    assert t.solve_relaxation() == engine.OPTIMAL
    got = t.objective_function_value() + float(gf.fixed_cost)
    # tests/netlib/test.rs:157 writes `< 1e-5` beside a pin of 8 digits, 5.5018459e+03: the optimum is 5501.845888.. (all three
    # engines, the f64 oracle, scipy's LU in scripts/f64_lab.py and HiGHS agree on these digits), 1.2e-5 away from the pin as
    # printed, so the reference's own assert could not hold for the exact optimum either (its test is `#[ignore]`d and has
    # never run).  1e-4 is the pin's rounding; the agreement on the full value is checked to 1e-8 relative.
    assert abs(got - 5.5018459e+03) < 1e-4
    assert abs(got - 5501.845888286) <= 1e-8 * 5501.845888286
    ident, basic, min_b = t.check_basis()
    assert ident <= 1e-6 and min_b >= -1e-6
    t.close()


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_LU, -1), (engine.ENGINE_REVISED, 0), (engine.ENGINE_TABLEAU, 32)])
def test_25fv47_phase_two_pivot_by_pivot_from_the_oracles_phase_one_basis(kind, block):
    """Phase 2 of C3 from a COMMON start: the f64 oracle's basis at the end of its phase 1 is handed to every engine through
    `relp_from_basis` (InverseMaintener::from_basis, carry/mod.rs:428-463), then engine and oracle run phase 2 (SteepestDescent)
    side by side.  Identical pivots for at least the first 100 (measured: 111 on the LU engine, whose factors are fresh every 48
    pivots while the oracle's inverse is only ever updated; Dantzig's rule on this LP has reduced costs that tie within rounding
    early in phase 2), and the same optimum at the end on every path."""
    from lp_files import load
    gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
    ref = relp_f64.OracleF64(md)
    assert ref.run(through_phases=False) == "phase_one_done"
    n1 = len(ref.trace)
    assert ref.filtered_rows() == []
    basis, b_start = ref.basis(), ref.b()
    assert ref.run() == "optimal"
    phase2 = ref.trace[n1:]
    t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 14)
    if kind == engine.ENGINE_REVISED:
        t.set_reinversion_interval(0)              # like the oracle: an inverse that is only ever updated
    t.from_basis(basis)
    assert t.phase == 2
    np.testing.assert_allclose(t.b(), b_start, rtol=1e-6, atol=1e-6 * max(1.0, np.max(np.abs(b_start))))
    assert t.run(1 << 20)[1] == engine.OPTIMAL
    tr = t.trace()
    same = next((k for k, (a, b) in enumerate(zip(tr, phase2)) if a != b), min(len(tr), len(phase2)))
    print(f"25FV47 phase 2: engine {len(tr)} pivots, oracle {len(phase2)}, identical for the first {same}")
    assert same >= 100
    got = t.objective_function_value() + float(gf.fixed_cost)
    assert abs(got - (ref.objective + float(gf.fixed_cost))) <= 1e-8 * abs(got)
    t.close()


@pytest.mark.parametrize("kind,block", [(engine.ENGINE_REVISED, 0), (engine.ENGINE_LU, -1)])
def test_25fv47_phase_one_against_the_f64_oracle(kind, block):
    """Phase 1 of 25FV47 (about 5,600 pivots, FirstProfitableWithMemory, 446 artificial variables driven out) against
    oracle/relp_f64.c under the default tolerances: the first 1,000 pivots are identical, and both end phase 1
    feasible.  The whole phase cannot be compared pivot by pivot in f64: this LP has reduced costs and ratios that
    tie to within rounding, the first one that resolves differently appears after ~1,300 pivots, and the oracle
    itself takes 5,602 pivots on one host and 5,639 on another (-march=native changes its FMA contraction).  The
    explicit-inverse engine runs without its periodic re-inversion here (the oracle is `Carry<_, BasisInverseRows>`
    literally: an inverse that is only ever updated)."""
    from lp_files import load
    gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
    ref = relp_f64.OracleF64(md)
    assert ref.run(through_phases=False) == "phase_one_done"
    t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 14)
    if kind == engine.ENGINE_REVISED:
        t.set_reinversion_interval(0)
    done, oc = t.run(1 << 20)
    assert oc == engine.PHASE_ONE_DONE
    tr = t.trace()
    first_diff = next((k for k, (a, b) in enumerate(zip(tr, ref.trace)) if a != b), min(len(tr), len(ref.trace)))
    print(f"25FV47 phase 1: engine {len(tr)} pivots, oracle {len(ref.trace)}, identical for the first {first_diff}")
    assert first_diff >= 1000, f"traces differ at pivot {first_diff}"
    assert t.phase == 2 and t.b().min() >= -1e-7
    t.close()


# ------------------------------------------------------------------------------------------------
# Full size of the north-star workload: 10,000 x 10,000 dense (bench.py's default)
# ------------------------------------------------------------------------------------------------
def test_dense10k_full_size_engines_agree_and_match_the_cpu_oracle_prefix():
    """The LP bench.py measures, generated in HBM by relp_synth_fill_dense (bit-identical to the numpy
    generator).  Too large for a CPU solve, so: the first 40 pivots equal the CPU oracle's, the tableau and
    the revised engine walk the same 320 pivots (5 flushes of the blocked update each), and the
    size-independent invariants hold afterwards."""
    import ctypes as C
    lib = engine.load_library()
    m = n = 10000
    seed = 20250002
    lp = synthetic.dense_lp(m, n, seed)
    md = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])
    ref = relp_f64.OracleF64(md.ensure_csc())
    ref.run(40)
    ptr = C.c_void_p()
    assert lib.relp_device_alloc(C.byref(ptr), m * n * 8) == 0
    assert lib.relp_synth_fill_dense(ptr, m, m, n, seed, 0, None) == 0
    counts = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=md.b, cost=md.cost, upper_bound=md.upper_bound)
    traces = []
    for kind in (engine.ENGINE_TABLEAU, engine.ENGINE_REVISED):
        t = engine.Tableau(counts, engine=kind, device_dense_ptr=ptr.value, device_dense_ld=m, trace_capacity=1024)
        assert t.update_block() == 64
        assert t.run(1)[1] == engine.PHASE_ONE_DONE
        objs = []
        for _ in range(5):
            done, outcome = t.run(64)
            assert done == 64 and outcome == engine.RUNNING
            objs.append(t.objective_function_value())
        assert all(b <= a + 1e-9 for a, b in zip(objs, objs[1:]))         # monotone objective
        tr = t.trace()
        assert tr[:40] == ref.trace
        b = t.b()
        basis = t.basis_indices()
        assert b.min() >= -1e-8 and len(set(basis.tolist())) == m
        d = t.relative_costs()
        assert np.max(np.abs(d[basis])) <= 1e-7                           # basic reduced costs vanish
        # objective = c_B . x_B
        c = np.concatenate([md.cost, np.zeros(m)])
        assert abs(c[basis] @ b - objs[-1]) <= 1e-9 * abs(objs[-1])
        traces.append(tr)
        t.close()
    assert traces[0] == traces[1] and len(traces[0]) == 320
    assert lib.relp_device_free(ptr) == 0


# ------------------------------------------------------------------------------------------------
# Config C4: synthetic dense 10,000 x 50,000 (BASELINE.json configs[3]), one GPU and 8 shards on one GPU
# ------------------------------------------------------------------------------------------------
def _c4_fixture():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c4_prefix.json")
    fx = json.load(open(path))
    assert (fx["m"], fx["n"], fx["seed"]) == (10000, 50000, 20250003) and fx["pivots"] >= 40
    return fx, [tuple(t) for t in fx["trace"]]


def _c4_counts():
    m, n, seed = 10000, 50000, 20250003
    b = n * (1000 + (synthetic.splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 4000.0
    c = -(1000 + (synthetic.splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 1000.0
    return m, n, seed, b, c


def test_c4_full_size_on_one_gpu_matches_the_oracle_prefix_and_both_engines_agree():
    """10,000 x 50,000 generated in HBM (4 GB; the tableau is 4.8 GB).  The first 40 pivots equal the f64 CPU
    oracle's (tests/golden/c4_prefix.json, scripts/gen_c4_prefix.py: too large to re-run here), the tableau and the
    revised engine walk the same 192 pivots (three flush blocks of 64), and the size-independent invariants hold."""
    import ctypes as C
    lib = engine.load_library()
    fx, prefix = _c4_fixture()
    m, n, seed, b, c = _c4_counts()
    ptr = C.c_void_p()
    assert lib.relp_device_alloc(C.byref(ptr), m * n * 8) == 0
    assert lib.relp_synth_fill_dense(ptr, m, m, n, seed, 0, None) == 0
    counts = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=b, cost=c, upper_bound=np.full(n, np.inf))
    traces = []
    for kind in (engine.ENGINE_TABLEAU, engine.ENGINE_REVISED):
        t = engine.Tableau(counts, engine=kind, device_dense_ptr=ptr.value, device_dense_ld=m, trace_capacity=1024)
        assert t.update_block() == (96 if kind == engine.ENGINE_TABLEAU else 64)       # the automatic choices at this size
        assert t.run(1)[1] == engine.PHASE_ONE_DONE
        objs = []
        for _ in range(3):
            done, outcome = t.run(64)
            assert done == 64 and outcome == engine.RUNNING
            objs.append(t.objective_function_value())
        assert all(y <= x + 1e-9 for x, y in zip(objs, objs[1:]))           # monotone objective
        tr = t.trace()
        assert tr[:40] == prefix[:40]
        bb, basis = t.b(), t.basis_indices()
        assert bb.min() >= -1e-8 and len(set(basis.tolist())) == m
        d = t.relative_costs()
        assert np.max(np.abs(d[basis])) <= 1e-7                              # basic reduced costs vanish
        cc = np.concatenate([c, np.zeros(m)])
        assert abs(cc[basis] @ bb - objs[-1]) <= 1e-9 * abs(objs[-1])       # objective = c_B . x_B
        traces.append(tr)
        t.close()
    assert traces[0] == traces[1] and len(traces[0]) == 192
    assert lib.relp_device_free(ptr) == 0


def test_c4_full_size_column_sharded_over_eight_ranks_on_one_gpu():
    """BASELINE.json configs[3] as it is meant to run: the 60,000 stored tableau columns split over 8 ranks (7,500
    each), every rank's structural columns generated in HBM, one all-gather of the PRICE candidates per pivot, the
    native loop (`relp_shard_run`) on one thread per rank with tests/shard_threads.py standing in for RCCL (which
    refuses two ranks on one device).  Every rank walks the oracle's 40-pivot prefix and the same 192 pivots
    (three local flushes) with the same objective and b."""
    import ctypes as C
    import torch  # noqa: F401
    lib = engine.load_library()
    fx, prefix = _c4_fixture()
    m, n, seed, b, c = _c4_counts()
    owned = []

    def make_md(cfg):
        md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=b, cost=c, upper_bound=np.full(n, np.inf))
        lo, hi = engine.shard_plan(md, cfg)
        ptr = C.c_void_p()
        assert lib.relp_device_alloc(C.byref(ptr), max(hi - lo, 1) * m * 8) == 0
        assert lib.relp_synth_fill_dense(ptr, m, m, hi - lo, seed, lo, None) == 0
        owned.append(ptr)
        return md, dict(device_dense_ptr=ptr.value, device_dense_ld=m)
    results = _c4_sharded_run(lib, make_md, 192)
    for p in owned:
        assert lib.relp_device_free(p) == 0
    first = results[0]
    assert first[0][:40] == prefix[:40] and len(first[0]) == 192
    for tr, obj, bb in results[1:]:
        assert tr == first[0] and obj == first[1]
        np.testing.assert_array_equal(bb, first[2])
    assert first[2].min() >= -1e-8


def _c4_sharded_run(lib, make_md, pivots, world=8):
    import ctypes as C
    import torch
    from shard_threads import ThreadRank, ThreadWorld, run_ranks
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    shared = ThreadWorld(world)
    tabs, ranks = [], []
    for r in range(world):
        cfg = engine.default_config(shard_rank=r, shard_count=world, engine=engine.ENGINE_TABLEAU, update_block=64,
                                    trace_capacity=1024, poll_interval=64)
        md, kw = make_md(cfg)
        t = engine.Tableau(md, config=cfg, **kw)
        tabs.append(t)
        ranks.append(ThreadRank(shared, r, lib, t.handle, torch, dev))

    def body(r):
        t = tabs[r]
        done, oc = C.c_int64(), C.c_int32()
        assert lib.relp_shard_run(t.handle, 1, C.byref(done), C.byref(oc)) == 0, lib.relp_last_error(t.handle).decode()
        assert oc.value == engine.PHASE_ONE_DONE
        assert lib.relp_shard_run(t.handle, pivots, C.byref(done), C.byref(oc)) == 0, lib.relp_last_error(t.handle).decode()
        assert done.value == pivots and oc.value == engine.RUNNING
        return t.trace(), t.objective_function_value(), t.b()
    out = run_ranks(world, body)
    assert not shared.errors, shared.errors
    assert all(rk.calls["allgather"] >= pivots for rk in ranks)
    for t in tabs:
        t.close()
    return out


@pytest.mark.parametrize("degenerate_rows", ["all", "most", "few"])
def test_ratio_test_from_block_minima_on_many_degenerate_rows(degenerate_rows):
    """17,000 rows = 67 blocks of 256: the tableau engine's ratio test starts from the per-block minimum ratios
    (k_ratio_blocks) and re-reads only the blocks inside the tie band.  With b = 0 on all / most rows every block
    is inside the band (more than the 64 the block list holds: the kernel then walks all blocks), with b = 0 on a
    few rows a handful are.  The revised engine's ratio test scans every row (k_ratio); both must walk the same
    pivots, and the leaving column must be the smallest one among the tied rows (tableau/mod.rs:229-239)."""
    m, n, seed = 17000, 24, 41
    lp = synthetic.dense_lp(m, n, seed)
    b = lp["b"].copy()
    rng = np.random.default_rng(5)
    if degenerate_rows == "all":
        b[:] = 0.0
    elif degenerate_rows == "most":
        b[rng.random(m) < 0.97] = 0.0
    else:
        b[rng.choice(m, size=9, replace=False)] = 0.0
    traces, objs = [], []
    for kind, block in ((engine.ENGINE_TABLEAU, 64), (engine.ENGINE_TABLEAU, 3), (engine.ENGINE_REVISED, 0)):
        t = engine.Tableau(MatrixData.from_dense_le(lp["A"], b, lp["c"]), engine=kind, update_block=block, trace_capacity=4096)
        assert t.solve_relaxation() == engine.OPTIMAL
        traces.append(t.trace())
        objs.append(t.objective_function_value())
        t.close()
    assert traces[0] == traces[1] == traces[2] and len(traces[0]) >= 1
    assert max(objs) - min(objs) <= 1e-9 * max(1.0, abs(objs[0]))
    if degenerate_rows == "all":
        assert abs(objs[0]) <= 1e-12                       # x = 0 is the only feasible point
        # every row ties at ratio 0 in the first pivot: the smallest leaving column is row 0's slack
        assert traces[0][0][2] == 0 and traces[0][0][3] == n


@pytest.mark.parametrize("block", [8, 32, 64])
def test_25fv47_on_the_tableau_engine_with_periodic_retabulation(block):
    """The dense tableau is only ever updated (T0 += W R0 at every flush); on 25FV47 it drifts until phase 1 ends in
    `no_row_phase_one` (K = 32, 64 without re-tabulation).  Every 1,000 pivots (default for sparse input below 4,097 rows) T0 is
    recomputed column by column from a fresh factorisation of the basis (k_lu_ftran_cols: one workgroup per stored
    column), b and d with it: the reference's pin, and a tableau that still is B^-1 [A | I]."""
    from lp_files import load
    gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
    t = engine.Tableau(md, engine=engine.ENGINE_TABLEAU, update_block=block)
    assert t.solve_relaxation() == engine.OPTIMAL
    assert t.reinversions() >= 8
    got = t.objective_function_value() + float(gf.fixed_cost)
    assert abs(got - 5.5018459e+03) < 1e-4
    ident, basic, min_b = t.check_basis()
    assert ident <= 1e-6 and min_b >= -1e-6
    t.close()


@pytest.mark.parametrize("block", [0, 5, 16, 20])
def test_25fv47_with_periodic_reinversion_is_block_independent(block):
    """An explicit inverse that is only ever updated loses accuracy on 25FV47 (max |B^-1 B - I| reaches 1e-6 .. 1e-4
    after ~2,000 pivots); without re-inversion the blocked update with K = 5, 16, 20 blows up in phase 1.  The
    revised engine rebuilds B^-1, b, -pi from the basis columns every 1,000 pivots (default for sparse input below 4,097 rows):
    every block length reaches the reference's pin, and the inverse stays accurate."""
    from lp_files import load
    gf, ex, md, emd = load("netlib/25FV47.SIF", fixed=True)
    t = engine.Tableau(md, engine=engine.ENGINE_REVISED, update_block=block)
    assert t.solve_relaxation() == engine.OPTIMAL
    assert t.reinversions() >= 8
    got = t.objective_function_value() + float(gf.fixed_cost)
    assert abs(got - 5.5018459e+03) < 1e-4
    ident, basic, min_b = t.check_basis()
    assert ident <= 1e-6 and min_b >= -1e-6
    t.close()


def test_reinversion_every_few_pivots_keeps_the_oracle_path():
    """Re-inversion only refreshes numbers: with B^-1, b and -pi rebuilt every 4 pivots (both phases, artificial and
    wrapped columns in the basis, redundant rows removed) well-conditioned LPs still walk the oracle's pivots."""
    rng = np.random.default_rng(99)
    checked = 0
    for case in range(40):
        m, n = int(rng.integers(5, 70)), int(rng.integers(5, 100))
        d = synthetic.mixed_lp(m, n, 9300 + case, nnz_per_col=int(rng.integers(2, 6)), frac_negative_cost=0.1,
                               infeasible=(case % 9 == 8))
        md = MatrixData.from_sparse_dict(d)
        ref = relp_f64.OracleF64(md)
        status = ref.run(200000)
        # (when the reference deletes a non-redundant row - see test_random_mixed_lps_... - the engine keeps the
        # literal state and stops re-inverting; those cases pass here because of that)
        quirk = any(r >= md.nr_eq + md.nr_range for r in ref.filtered_rows())
        for kind, block in ((engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 3), (engine.ENGINE_TABLEAU, 3)):
            if quirk and kind == engine.ENGINE_TABLEAU:
                continue                                   # defined by the explicit inverse only
            t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 15)
            t.set_reinversion_interval(4)
            assert engine.OUTCOME_NAMES[t.solve_relaxation()] == status, case
            assert t.trace() == ref.trace, case
            if len(ref.trace) >= 8:
                assert t.reinversions() >= 1
            if status == "optimal":
                assert abs(t.objective_function_value() - ref.objective) <= OBJ_RTOL * max(1.0, abs(ref.objective)), case
            t.close()
            checked += 1
    assert checked >= 100
    with pytest.raises(engine.RelpError):
        engine.Tableau(md, engine=engine.ENGINE_LU).set_reinversion_interval(10)


# ------------------------------------------------------------------------------------------------
# Randomised differential test (a fixed slice of tests/tools/fuzz_gpu.py)
# ------------------------------------------------------------------------------------------------
def test_random_mixed_lps_every_engine_matches_the_oracle():
    """120 random LPs with ==, <=, >= rows, bounds and rank deficiencies (rows removed at the phase switch,
    including rows that own a slack: its column becomes empty, matrix_data.rs:592-614), random engine and
    block length: outcome, pivot trace, objective and b equal the f64 oracle's.  Cases in which the reference
    removes a non-redundant row (it pushes the artificial's INDEX, phase_one.rs:252, which differs from its row
    for >= rows) have no defined answer outside the explicit-inverse back end and are checked on the revised
    engine only - see tests/tools/fuzz_gpu.py."""
    rng = np.random.default_rng(20250003)
    kinds = [(engine.ENGINE_REVISED, (0, 1, 3, 7, 64)), (engine.ENGINE_TABLEAU, (1, 2, 5, 64)), (engine.ENGINE_LU, (1, 2, 6, 64))]
    checked = removed = 0
    for case in range(120):
        m, n = int(rng.integers(4, 80)), int(rng.integers(4, 120))
        md = MatrixData.from_sparse_dict(synthetic.sparse_lp(
            m, n, 7000 + case, nnz_per_col=int(rng.integers(2, 7)), frac_eq=float(rng.uniform(0, 0.5)),
            frac_ge=float(rng.uniform(0, 0.4)), frac_bounded=float(rng.uniform(0, 0.6))))
        kind, blocks = kinds[int(rng.integers(0, 3))]
        block = int(blocks[int(rng.integers(0, len(blocks)))])
        ref = relp_f64.OracleF64(md)
        status = ref.run(200000)
        rows = ref.filtered_rows()
        removed += bool(rows)
        if any(r >= md.nr_eq + md.nr_range for r in rows):
            kind, block = engine.ENGINE_REVISED, block if kind == engine.ENGINE_REVISED else 0
        t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 16)
        try:
            outcome = engine.OUTCOME_NAMES[t.solve_relaxation()]
        except engine.RelpError as e:
            assert kind == engine.ENGINE_LU and "singular" in str(e) and rows, (case, str(e))
            continue
        assert outcome == status, case
        assert t.trace() == ref.trace, case
        if status == "optimal":
            assert abs(t.objective_function_value() - ref.objective) <= OBJ_RTOL * max(1.0, abs(ref.objective)), case
            assert np.max(np.abs(t.b() - ref.b())) <= VEC_TOL * max(1.0, np.max(np.abs(ref.b()))), case
        checked += 1
        t.close()
    assert checked >= 110 and removed >= 5


def test_random_lps_with_ranges_unbounded_and_infeasible_outcomes():
    """100 random LPs from `synthetic.mixed_lp`: every row kind including ranges (two-entry slack columns,
    range-bound rows), bounded variables, negative costs (some LPs are unbounded) and contradictory equality
    rows (some are infeasible).  Outcome and pivot trace equal the f64 oracle's on a random engine."""
    rng = np.random.default_rng(20250004)
    kinds = [(engine.ENGINE_REVISED, (0, 1, 3, 7, 64)), (engine.ENGINE_TABLEAU, (1, 2, 5, 64)), (engine.ENGINE_LU, (1, 2, 6, 64))]
    seen = {"optimal": 0, "unbounded": 0, "infeasible": 0}
    for case in range(100):
        m, n = int(rng.integers(6, 70)), int(rng.integers(4, 100))
        md = MatrixData.from_sparse_dict(synthetic.mixed_lp(
            m, n, 9000 + case, nnz_per_col=int(rng.integers(2, 6)), frac_eq=float(rng.uniform(0, 0.3)),
            frac_range=float(rng.uniform(0, 0.3)), frac_ge=float(rng.uniform(0, 0.3)), frac_bounded=float(rng.uniform(0, 0.6)),
            frac_negative_cost=float(rng.choice([0.0, 0.0, 0.1, 0.3])), infeasible=bool(rng.random() < 0.15)))
        kind, blocks = kinds[int(rng.integers(0, 3))]
        block = int(blocks[int(rng.integers(0, len(blocks)))])
        ref = relp_f64.OracleF64(md)
        status = ref.run(200000)
        seen[status] += 1
        rows = ref.filtered_rows()
        if any(r >= md.nr_eq + md.nr_range for r in rows):
            kind, block = engine.ENGINE_REVISED, block if kind == engine.ENGINE_REVISED else 0
        t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 16)
        try:
            outcome = engine.OUTCOME_NAMES[t.solve_relaxation()]
        except engine.RelpError as e:
            assert kind == engine.ENGINE_LU and "singular" in str(e) and rows, (case, str(e))
            continue
        assert outcome == status, case
        assert t.trace() == ref.trace, case
        if status == "optimal":
            assert abs(t.objective_function_value() - ref.objective) <= OBJ_RTOL * max(1.0, abs(ref.objective)), case
        t.close()
    assert min(seen.values()) >= 3, seen


def test_edge_case_lps_on_every_engine():
    """tests/edge_lps.py: 1 x 1 problems, empty columns, an all-zero right-hand side, zero costs, duplicate and
    contradictory equalities, a range row, bounds only - outcome, trace and objective equal the oracle's."""
    from edge_lps import CASES
    for name, problem in CASES.items():
        ref = relp_f64.OracleF64(problem.ensure_csc())
        status = ref.run(10000)
        for kind, block in ((engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 4), (engine.ENGINE_TABLEAU, 4), (engine.ENGINE_LU, 4)):
            t = engine.Tableau(problem, engine=kind, update_block=block, trace_capacity=256)
            assert engine.OUTCOME_NAMES[t.solve_relaxation(max_iters=1000)] == status, (name, kind)
            assert t.trace() == ref.trace, (name, kind)
            if status == "optimal":
                assert abs(t.objective_function_value() - ref.objective) <= OBJ_RTOL * max(1.0, abs(ref.objective)), (name, kind)
            t.close()


@pytest.mark.parametrize("name,objective", [("model_data_1", 123 / 38), ("model_data_3_1", 70.0), ("model_data_3_2", 180.0),
                                            ("model_data_3_3", 245.0), ("model_data_3_4", 2250.0), ("model_data_4", 7.0),
                                            ("model_data_6", 28.0)])
def test_unicamp_files_on_every_engine(name, objective):
    """tests/unicamp/test.rs (the reference solves these with `Carry<_, LUDecomposition<_>>`): exact optimum pins
    on all three GPU engines, pivot sequence equal to the oracle's.  Presolve may solve a file completely."""
    from lp_files import load
    from rust_lp_amd import general_form
    try:
        gf, ex, md, emd = load(f"unicamp/{name}.mps")
    except general_form.Solved as solved:
        assert abs(float(solved.objective) - objective) <= 1e-9 * max(1.0, abs(objective))
        return
    ref = relp_f64.OracleF64(md)
    status = ref.run()
    for kind in (engine.ENGINE_LU, engine.ENGINE_REVISED, engine.ENGINE_TABLEAU):
        t = engine.Tableau(md, engine=kind, trace_capacity=4096)
        assert engine.OUTCOME_NAMES[t.solve_relaxation()] == status == "optimal"
        assert t.trace() == ref.trace
        assert abs(t.objective_function_value() + float(gf.fixed_cost) - objective) <= 1e-9 * max(1.0, abs(objective))


@pytest.mark.parametrize("path,fixed,objective,tol", FILES)
def test_reference_problem_files_on_the_lu_engine_with_lookahead_refactorisation(path, fixed, objective, tol):
    """The LU engine at a refactorisation interval of 24 (from 24 on the pivot kernel is relaunched on the old factors while
    the host factorises a snapshot of the basis, and the basis changes made meanwhile are replayed on the new factors,
    `Engine::lu_refactor_lookahead`): the pivot sequence of the f64 oracle and the reference's objective pin, like at the
    reference's own cadence."""
    from lp_files import load
    gf, ex, md, emd = load(path, fixed=fixed)
    t = engine.Tableau(md, trace_capacity=1 << 15, engine=engine.ENGINE_LU, update_block=24)
    assert t.solve_relaxation() == engine.OPTIMAL
    ref = relp_f64.OracleF64(md)
    assert ref.run() == "optimal"
    assert t.trace() == ref.trace
    got = t.objective_function_value() + float(gf.fixed_cost)
    assert abs(got - objective) < max(tol, 1e-9 * abs(objective))
    if len(ref.trace) > 40:
        stats = t.lu_stats()
        assert stats["refactorisations"] >= 2
        # the look-ahead path itself ran: factorisations installed beside the running kernel, journal entries replayed
        assert stats["lookahead"] == 8 and stats["lookahead_installs"] >= 1
        if len(ref.trace) > 200:
            assert stats["replayed_changes"] >= 1
    t.close()


@pytest.mark.parametrize("path,fixed", [("netlib/SHARE1B.SIF", True), ("netlib/BOEING2.SIF", True), ("netlib/SC205.SIF", True)])
def test_lookahead_refactorisation_on_and_off_walk_the_same_pivots(path, fixed, monkeypatch):
    """RELP_LU_LOOKAHEAD is read when the engine is created: 0 = every refactorisation synchronous, 8 = prepared beside the
    running kernel and completed by a replay of the journal.  Same pivot sequence and the same objective to rounding."""
    from lp_files import load
    gf, ex, md, emd = load(path, fixed=fixed)
    out = {}
    for la in (0, 8):
        monkeypatch.setenv("RELP_LU_LOOKAHEAD", str(la))
        t = engine.Tableau(md, trace_capacity=1 << 15, engine=engine.ENGINE_LU, update_block=24)
        assert t.solve_relaxation() == engine.OPTIMAL
        stats = t.lu_stats()
        assert stats["lookahead"] == la
        assert (stats["lookahead_installs"] > 0) == (la > 0)
        out[la] = (t.trace(), t.objective_function_value())
        t.close()
    assert out[0][0] == out[8][0]
    assert abs(out[0][1] - out[8][1]) <= 1e-9 * max(1.0, abs(out[0][1]))


def test_two_launch_pivot_equals_three_launch_pivot_bit_for_bit(monkeypatch):
    """The tableau engine's loop runs the ratio test inside the update launch (k_tab_ratio_update_all: every workgroup repeats
    it; b, the basis array, row r of W and n_eta double-buffered); RELP_FUSED_UPDATE=0 is the three-launch pivot of round 1
    (k_ratio_blocks + k_tab_update_all, still the path of the step-wise API and the Python sharded loop).  Same arithmetic:
    identical traces, objectives and right-hand sides on a dense LP and on a two-phase sparse one, over several flushes."""
    from lp_files import load
    cases = []
    lp = synthetic.dense_lp(300, 500, 4242)
    cases.append(MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]))
    cases.append(load("netlib/SC205.SIF", fixed=True)[2])
    for md in cases:
        got = []
        for fused in ("1", "0"):
            monkeypatch.setenv("RELP_FUSED_UPDATE", fused)
            t = engine.Tableau(md, trace_capacity=1 << 15, engine=engine.ENGINE_TABLEAU, update_block=16)
            assert t.solve_relaxation() == engine.OPTIMAL
            got.append((t.trace(), t.objective_function_value(), np.array(t.b())))
            t.close()
        assert got[0][0] == got[1][0] and len(got[0][0]) > 64
        assert got[0][1] == got[1][1]
        assert np.array_equal(got[0][2], got[1][2])
