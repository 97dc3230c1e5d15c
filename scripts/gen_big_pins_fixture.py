"""Generates tests/golden/big_pins.npz: what oracle/relp_f64.c (the f64 CPU restatement of the reference path) does on the
three Netlib files the reference `#[ignore]`s as too expensive (tests/netlib/test.rs:137-166), for the GPU tier to compare
with -- the oracle needs minutes per file, the GPU tests seconds.

  80BAU3B, reference rules literally: the whole pivot trace, the "rows" removed at the phase switch (phase_one.rs:252
      pushes artificial INDICES: 330, 375, 390 name `<=` rows here), the objective it ends at (964,593.50, NOT the pin);
  80BAU3B, artificial_removal = 1: the objective (the pin, 987,224.1924);
  GREENBEA / GREENBEB, ratio_rule = 1 and artificial_removal = 1: the first 1,500 pivots.

usage: python scripts/gen_big_pins_fixture.py   (about 5 minutes)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa: E402,F401
from oracle import relp_f64  # noqa: E402
from lp_files import load  # noqa: E402


def main():
    out = {}
    gf, ex, md, emd = load("netlib/80BAU3B.SIF", fixed=True)
    o = relp_f64.OracleF64(md)
    assert o.run() == "optimal"
    out["bau_literal_trace"] = np.array(o.trace, dtype=np.int32)
    out["bau_literal_filtered"] = np.array(o.filtered_rows(), dtype=np.int32)
    out["bau_literal_objective"] = np.array([o.objective + float(gf.fixed_cost)])
    print("80BAU3B literal:", len(o.trace), "pivots, objective", out["bau_literal_objective"][0], "removed", o.filtered_rows())
    o = relp_f64.OracleF64(md, artificial_removal=1)
    assert o.run() == "optimal"
    out["bau_textbook_trace"] = np.array(o.trace, dtype=np.int32)
    out["bau_textbook_objective"] = np.array([o.objective + float(gf.fixed_cost)])
    print("80BAU3B textbook:", len(o.trace), "pivots, objective", out["bau_textbook_objective"][0], "removed", o.filtered_rows())
    for name in ("GREENBEA", "GREENBEB"):
        gf, ex, md, emd = load(f"netlib/{name}.SIF", fixed=True)
        o = relp_f64.OracleF64(md, ratio_rule=1, artificial_removal=1)
        assert o.run(max_iters=1500) == "iteration_limit"
        out[name.lower() + "_prefix"] = np.array(o.trace, dtype=np.int32)
        print(name, "prefix", len(o.trace))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "big_pins.npz"), **out)


if __name__ == "__main__":
    main()
