"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of rust_lp_amd.sharded.ShardedPivotLoop with the
numpy stand-in for the shard entry points.  Every rank must walk the pivot sequence of the
single-process C oracle and end with the same objective."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, m, n, seed, q, tableau=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rust_lp_amd  # noqa: F401
    from rust_lp_amd import synthetic
    from rust_lp_amd.sharded import ShardedPivotLoop
    from shard_standin import NumpyShardOps, NumpyTableauShardOps
    lp = synthetic.dense_lp(m, n, seed)
    per = -(-n // world)
    lo, hi = min(n, rank * per), min(n, rank * per + per)
    if tableau:
        ops = NumpyTableauShardOps(rank, world, m, n, lp["A"], lp["b"], lp["c"])
    else:
        ops = NumpyShardOps(rank, world, m, n, lp["A"][:, lo:hi], lo, lp["b"], lp["c"])
    loop = ShardedPivotLoop(ops, dist, torch.device("cpu"), poll_interval=7)
    oc1 = loop.finish_phase_one()
    done, oc = loop.run(1 << 20)
    q.put((rank, oc1, done, oc, ops.trace, -ops.minus_obj))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tableau", [False, True])
@pytest.mark.parametrize("world,m,n,seed", [(2, 24, 36, 5), (2, 33, 20, 8), (3, 40, 50, 2)])
def test_sharded_loop_matches_oracle(world, m, n, seed, tableau):
    sys.path.insert(0, ROOT)
    import rust_lp_amd  # noqa: F401
    from rust_lp_amd import MatrixData, synthetic
    from oracle import relp_f64
    lp = synthetic.dense_lp(m, n, seed)
    ref = relp_f64.OracleF64(MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]).ensure_csc())
    assert ref.run() == "optimal"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world + (10 if tableau else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, seed, q, tableau)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, oc1, done, oc, trace, obj in results:
        assert oc1 == 4 and oc == 1, (rank, oc1, oc)          # PHASE_ONE_DONE then OPTIMAL
        assert trace == ref.trace, f"rank {rank} diverged"
        assert done == len(ref.trace)
        assert abs(obj - ref.objective) <= 1e-9 * max(1.0, abs(ref.objective))


def test_shard_column_range_partition():
    import rust_lp_amd  # noqa: F401
    from rust_lp_amd import engine
    for n, g in [(10, 3), (10000, 8), (7, 8), (50000, 8)]:
        spans = [engine.shard_column_range(n, r, g) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
