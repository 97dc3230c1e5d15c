#!/usr/bin/env python3
"""Generates tests/golden/c4_prefix.json: the first pivots of the f64 CPU oracle (oracle/relp_f64.c, default
tolerances) on BASELINE.json configs[3], the synthetic dense LP 10,000 x 50,000 (rust-lp_amd/synthetic.py, seed
20250003).  The LP needs ~17 GB of host memory and a few minutes on the oracle, so the GPU tier compares the engines'
traces with this committed vector instead of re-running the oracle at full size (tests/test_gpu_parity.py,
test_c4_*); the same script with --check re-derives and compares.

  python scripts/gen_c4_prefix.py [--pivots 40] [--check]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rust_lp_amd import MatrixData, synthetic  # noqa: E402
from oracle import relp_f64  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "c4_prefix.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pivots", type=int, default=40)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    m, n, seed = 10000, 50000, 20250003
    # the dense matrix as CSC directly (every entry is non-zero): column j holds rows 0..m-1
    values = np.empty(m * n, dtype=np.float64)
    chunk = 1000 * m
    for lo in range(0, m * n, chunk):
        idx = np.arange(lo, min(lo + chunk, m * n), dtype=np.uint64)
        values[lo:lo + len(idx)] = (1 + (synthetic.splitmix64(seed, 0, idx) % np.uint64(999)).astype(np.int64)) / 1000.0
    b = n * (1000 + (synthetic.splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 4000.0
    c = -(1000 + (synthetic.splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 1000.0
    md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=b, cost=c, upper_bound=np.full(n, np.inf),
                    col_ptr=np.arange(n + 1, dtype=np.int64) * m, row_idx=np.tile(np.arange(m, dtype=np.int32), n), values=values)
    ref = relp_f64.OracleF64(md)
    ref.run(args.pivots)
    out = {"workload": "c4", "m": m, "n": n, "seed": seed, "tolerances": relp_f64.DEFAULT_TOLERANCES,
           "generator": "scripts/gen_c4_prefix.py (oracle/relp_f64.c)", "pivots": len(ref.trace),
           "trace": [list(t) for t in ref.trace], "objective": ref.objective}
    if args.check:
        old = json.load(open(OUT))
        assert old["trace"][:len(out["trace"])] == out["trace"][:len(old["trace"])], "trace differs from the committed vector"
        print("ok: committed vector reproduced")
        return
    with open(OUT, "w") as f:
        json.dump(out, f)
        f.write("\n")
    print(f"wrote {OUT}: {len(ref.trace)} pivots, objective {ref.objective!r}")


if __name__ == "__main__":
    main()
