"""Randomised run of the sharded entry points (G engines on one GPU, exchanges done with torch ops): calls
tests/test_gpu_parity.py::test_shard_entry_points_on_one_gpu with random world sizes, LP sizes, engines and
block lengths.  Usage: python tests/tools/fuzz_shards.py [N_CASES] [SEED]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import rust_lp_amd  # noqa: F401
from rust_lp_amd import engine
from test_gpu_parity import test_shard_entry_points_on_one_gpu as run_case

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
KINDS = [(engine.ENGINE_REVISED, (0, 2, 5, 64)), (engine.ENGINE_TABLEAU, (1, 3, 8, 64))]
bad = 0
for case in range(N):
    world = int(rng.integers(2, 6))
    m, n = int(rng.integers(3, 400)), int(rng.integers(world, 600))
    kind, blocks = KINDS[int(rng.integers(0, 2))]
    block = int(blocks[int(rng.integers(0, len(blocks)))])
    try:
        run_case(world, m, n, 100 + case, kind, block)
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH case {case}: world {world} {m}x{n} kind {kind} block {block}: {str(e)[:200]}", flush=True)
    if case % 10 == 9:
        print(f"... {case + 1} cases, {bad} mismatches", flush=True)
print(f"{N} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
