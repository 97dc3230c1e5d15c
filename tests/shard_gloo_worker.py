"""Worker process of tests/test_gpu_parity.py::test_python_sharded_loop_two_processes_general_lp (its own module so
that the spawned process imports torch before anything touches the HIP runtime)."""
import os
import sys

import torch
import torch.distributed as dist
import numpy as np


def gloo_rank(rank, world, port, path, fixed, q):
    """One process of the Python loop (torch.distributed, gloo) with the HIP shard entry points on cuda:0."""
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rust_lp_amd  # noqa: F401
    from rust_lp_amd import MatrixData as MD, engine as eng
    from rust_lp_amd.sharded import ShardedPivotLoop
    from lp_files import load
    gf, ex, md, emd = load(path, fixed=fixed)
    dense = np.array(md.ensure_dense().dense)
    cfg = eng.default_config(shard_rank=rank, shard_count=world, engine=eng.ENGINE_TABLEAU, update_block=5, trace_capacity=1 << 14)
    part = MD(nr_normal=md.nr_normal, nr_eq=md.nr_eq, nr_range=md.nr_range, nr_le=md.nr_le, nr_ge=md.nr_ge, b=md.b, cost=md.cost,
              upper_bound=md.upper_bound, ranges=md.ranges)
    lo, hi = eng.shard_plan(part, cfg)
    part.dense = np.asfortranarray(dense[:, lo:hi]) if hi > lo else np.zeros((dense.shape[0], 1), order="F")
    t = eng.Tableau(part, config=cfg)
    loop = ShardedPivotLoop(t, dist, torch.device("cuda", 0), poll_interval=8)
    done, oc = loop.solve_relaxation()
    torch.cuda.synchronize()
    q.put((rank, done, oc, t.trace(), t.objective_function_value(), loop.hook_calls))
    dist.barrier()
    dist.destroy_process_group()
