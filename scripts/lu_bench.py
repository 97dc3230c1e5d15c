"""Pivots per second of the three engines on sparse problems (synthetic Netlib-shaped LPs and the
reference's own files).  Usage: python scripts/lu_bench.py [synth M N SEED | FILE] ..."""
import sys
import time
sys.path.insert(0, ".")
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic


def load_file(rel):
    """MPS / SIF fixture -> presolved, standardised f64 MatrixData (the product path, no oracle involved)."""
    import os
    from rust_lp_amd import general_form, mps
    path = os.path.join("tests", "golden", "mps", rel)
    gf = general_form.GeneralForm.from_mps(mps.import_file(path, rel.endswith(".SIF")))
    return gf.to_matrix_data(gf.derive_matrix_data_exact())


def problems(argv):
    i = 0
    while i < len(argv):
        if argv[i] == "synth":
            m, n, s = map(int, argv[i + 1:i + 4])
            yield f"synth{m}x{n}", MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, s))
            i += 4
        else:
            yield argv[i], load_file(argv[i])
            i += 1


def main():
    argv = sys.argv[1:] or ["synth", "400", "800", "77"]
    for name, md in problems(argv):
        for label, kw in (("lu", dict(engine=engine.ENGINE_LU)), ("lu64", dict(engine=engine.ENGINE_LU, update_block=64)), ("lu32", dict(engine=engine.ENGINE_LU, update_block=32)),
                          ("lu11", dict(engine=engine.ENGINE_LU, update_block=11)),
                          ("revised", dict(engine=engine.ENGINE_REVISED, update_block=0)),
                          ("tableau", dict(engine=engine.ENGINE_TABLEAU, update_block=32))):
            t = engine.Tableau(md, **kw)
            if label.startswith("lu"):
                t.profile_enable(True, 100000, 1)
            t0 = time.perf_counter()
            try:
                oc = t.solve_relaxation()
            except engine.RelpError as e:
                print(f"{name:28s} {label:8s} failed: {e}", flush=True)
                continue
            dt = time.perf_counter() - t0
            it = t.iterations()
            extra = t.lu_stats() if label.startswith("lu") else ""
            if label.startswith("lu"):
                prof = {k: (c, round(ms, 2)) for k, (c, ms) in t.profile_read().items() if c}
                extra = f"{extra} events {prof}"
                ph = t.lu_phase_cycles()
                tot = sum(ph.values()) or 1
                extra += " phases% " + " ".join(f"{k}={100.0 * v / tot:.1f}" for k, v in ph.items()) + f" cycles/pivot={tot / max(it, 1):.0f}"
            print(f"{name:28s} {label:8s} {engine.OUTCOME_NAMES[oc]:10s} pivots {it:7d}  {dt:8.3f} s  {it / dt:9.0f} it/s  "
                  f"obj {t.objective_function_value():.9g} {extra}", flush=True)


if __name__ == "__main__":
    main()
