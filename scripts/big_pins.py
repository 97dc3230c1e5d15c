"""Probe: the reference's big `#[ignore]`d Netlib pins (tests/netlib/test.rs:137-166) on each engine.
usage: python scripts/big_pins.py NAME ENGINE [update_block] [max_seconds] [ratio_rule] [artificial_removal] [reinversion]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa: E402,F401
from rust_lp_amd import engine, general_form, mps  # noqa: E402

PINS = {"GREENBEA": -0.72555248129845987457557870574845e8, "GREENBEB": -0.43022602612065867539213672544432e7,
        "80BAU3B": 9.872241924e+05, "25FV47": 5.5018459e+03, "DFL001": 1.12664e7}
ENG = {"lu": engine.ENGINE_LU, "revised": engine.ENGINE_REVISED, "tableau": engine.ENGINE_TABLEAU}


def main():
    name, kind = sys.argv[1], sys.argv[2]
    block = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    max_s = float(sys.argv[4]) if len(sys.argv) > 4 else 120.0
    m = mps.import_file(os.path.join(ROOT, "tests", "golden", "mps", "netlib", name + ".SIF"), True)
    gf = general_form.GeneralForm.from_mps(m)
    ex = gf.derive_matrix_data_exact()
    md = gf.to_matrix_data(ex)
    rr = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    ar = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    t = engine.Tableau(md, engine=ENG[kind], update_block=block, trace_capacity=0, ratio_rule=rr, artificial_removal=ar)
    if len(sys.argv) > 7 and kind != "lu":
        t.set_reinversion_interval(int(sys.argv[7]))
    t0 = time.time()
    total = 0
    while True:
        done, oc = t.run(20000)
        total += done
        el = time.time() - t0
        obj = t.objective_function_value() + float(gf.fixed_cost)
        print(f"{name} {kind} K={block} rr={rr} ar={ar}: {total} pivots, {el:.2f}s, {total / max(el, 1e-9):.0f} it/s, phase {t.phase}, "
              f"objective {obj:.10g}, outcome {engine.OUTCOME_NAMES.get(oc, oc)}", flush=True)
        if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE):
            break
        if el > max_s:
            break
    print(f"{name} {kind}: pin {PINS[name]:.10g}, diff {obj - PINS[name]:.3g}, min b {t.b().min():.3g}")
    if kind == "lu":
        print(t.lu_stats())


if __name__ == "__main__":
    main()
