"""Differential fuzz of the column-sharded tableau engine on general LPs (every row kind, bounds, infeasible /
unbounded / rank-deficient cases): G ranks on one GPU through the native loop (`relp_shard_run`, threads +
in-process collectives of tests/shard_threads.py) against the single-GPU tableau engine and the f64 oracle.
Usage: python tests/tools/fuzz_shards_general.py [N_CASES] [SEED0]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa: F401  (before the first engine)
import numpy as np
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic
from oracle import relp_f64
from test_gpu_parity import _native_ranks

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
rng = np.random.default_rng(seed0)
bad = removed = zero_level = 0
outcomes = {}
for case in range(N):
    m, n = int(rng.integers(6, 90)), int(rng.integers(6, 140))
    if rng.random() < 0.4:
        d = synthetic.mixed_lp(m, n, seed0 + case, nnz_per_col=int(rng.integers(2, 7)), frac_negative_cost=float(rng.choice([0.0, 0.1])),
                               infeasible=bool(rng.random() < 0.15))
    else:
        d = synthetic.sparse_lp(m, n, seed0 + case, nnz_per_col=int(rng.integers(2, 7)), frac_eq=float(rng.uniform(0, 0.6)),
                                frac_ge=float(rng.uniform(0, 0.4)), frac_bounded=float(rng.uniform(0, 0.6)))
        d.setdefault("ranges", np.zeros(0))
        if rng.random() < 0.3:                      # degenerate right-hand sides
            d["b"][rng.random(m) < 0.4] = 0.0
    full = MatrixData.from_sparse_dict(d)
    ref = relp_f64.OracleF64(full)
    status = ref.run(200000)
    rows = ref.filtered_rows()
    quirk = any(r >= d["nr_eq"] + d["nr_range"] for r in rows)
    removed += bool(rows)
    zero_level += ref.nr_zero_level_pivots > 0
    single = engine.Tableau(full, engine=engine.ENGINE_TABLEAU, update_block=4, trace_capacity=1 << 15)
    single_outcome = single.solve_relaxation()
    dense = np.zeros((m, n))
    for j in range(n):
        for e in range(d["col_ptr"][j], d["col_ptr"][j + 1]):
            dense[d["row_idx"][e], j] = d["values"][e]
    world = int(rng.integers(2, 7))
    block = int(rng.choice([1, 3, 8, 64]))

    def make_md(cfg):
        md = MatrixData(nr_normal=n, nr_eq=d["nr_eq"], nr_range=d["nr_range"], nr_le=d["nr_le"], nr_ge=d["nr_ge"], b=d["b"],
                        cost=d["c"], upper_bound=d["ub"], ranges=d["ranges"])
        lo, hi = engine.shard_plan(md, cfg)
        md.dense = np.asfortranarray(dense[:, lo:hi]) if hi > lo else np.zeros((m, 1), order="F")
        return md
    try:
        results = _native_ranks(world, make_md, engine.ENGINE_TABLEAU, block, poll_interval=int(rng.choice([1, 5, 32])))
    except Exception as e:  # noqa: BLE001
        print("case", case, "seed", seed0 + case, "world", world, "FAILED", repr(e)[:300], flush=True)
        bad += 1
        continue
    for first, total, oc, trace, obj, b in results:
        final = oc if first == engine.PHASE_ONE_DONE else first
        ok = final == single_outcome and trace == single.trace()
        if ok and not quirk:
            ok = engine.OUTCOME_NAMES[final] == status and trace == ref.trace
            if ok and status == "optimal":
                ok = abs(obj - ref.objective) <= 1e-7 * max(1.0, abs(ref.objective))
        if not ok:
            print("case", case, "seed", seed0 + case, "world", world, "block", block, "MISMATCH", engine.OUTCOME_NAMES.get(final), status,
                  len(trace), len(single.trace()), len(ref.trace), flush=True)
            bad += 1
            break
    outcomes[status] = outcomes.get(status, 0) + 1
    single.close()
    if (case + 1) % 25 == 0:
        print("...", case + 1, "cases,", bad, "mismatches so far", flush=True)
print(N, "cases,", bad, "mismatches;", removed, "with redundant rows,", zero_level, "with zero-level pivots; oracle outcomes", outcomes)
sys.exit(1 if bad else 0)
