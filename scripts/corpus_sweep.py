"""Every LP of the reference's Netlib directory the build's front end accepts (tests/golden/corpus/, made by
scripts/gen_corpus_fixture.py) on the three engines, under `relp_default_config` (the reference's rules literally) and under the
f64 safeguards of `relp_robust_config` (largest-pivot ratio rule, textbook artificial removal, pivot rescue, adaptive re-inversion).

Per file, engine and configuration: outcome, pivots, objective, `check_basis` residuals, seconds; then per file the agreement of
the engines among themselves, with the reference's pin where tests/netlib/test.rs holds one, and with HiGHS on the same
standardised LP (an independent check, NOT the reference: "parity unpinned" for the ~60 files without a pin).

usage (GPU box): python scripts/corpus_sweep.py [--out gpurun_out/r4/corpus_sweep.json] [--max-pivots 400000] [--seconds 60] [NAME ...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_lp_amd  # noqa: E402,F401
from rust_lp_amd import engine  # noqa: E402
import corpus  # noqa: E402

ENGINES = (("lu", engine.ENGINE_LU, -1), ("tableau", engine.ENGINE_TABLEAU, -1), ("revised", engine.ENGINE_REVISED, -1))
CONFIGS = (("default", {}), ("robust", dict(ratio_rule=engine.RATIO_LARGEST_PIVOT, artificial_removal=engine.ARTIFICIAL_TEXTBOOK,
                                           pivot_rescue=1, auto_reinversion=1)))       # relp_robust_config with the engine of the leg


def solve(md, fixed, kind, block, cfg, max_pivots, seconds):
    t0 = time.perf_counter()
    out = {}
    try:
        t = engine.Tableau(md, engine=kind, update_block=block, **cfg)
    except engine.RelpError as e:
        return {"outcome": "create_failed", "error": str(e)[:160]}
    try:
        oc, total = engine.RUNNING, 0
        while True:
            if t.phase == 1:
                done, oc = t.run(20000)
                total += done
                if oc == engine.PHASE_ONE_DONE:
                    continue
            else:
                done, oc = t.run(20000)
                total += done
            if oc != engine.RUNNING or total >= max_pivots or time.perf_counter() - t0 > seconds:
                break
        out["outcome"] = engine.OUTCOME_NAMES.get(oc, str(oc)) if oc != engine.RUNNING else "limit"
        out["pivots"] = int(t.iterations())
        out["rows"] = int(t.nr_rows())
        if oc == engine.OPTIMAL:
            out["objective"] = t.objective_function_value() + fixed
            ident, basic, min_b = t.check_basis()
            out["check_basis"] = [ident, basic, min_b]
        if kind == engine.ENGINE_LU:
            out["layout"] = t.lu_kernel_layout()["layout"]
        if cfg.get("pivot_rescue"):
            out["robust_stats"] = t.robust_stats()
    except engine.RelpError as e:
        out["outcome"] = "error"
        out["error"] = str(e)[:160]
    finally:
        t.close()
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out


def rel(a, b):
    return abs(a - b) / max(1.0, abs(a), abs(b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r4", "corpus_sweep.json"))
    ap.add_argument("--max-pivots", type=int, default=400000)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--verified", action="store_true", help="engine.solve_verified per file instead of the engine x configuration grid")
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    idx = corpus.index()
    names = args.names or sorted(n for n, r in idx.items() if "nr_rows" in r)
    names.sort(key=lambda n: idx[n]["nr_rows"] * idx[n]["nr_columns"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    results = {}
    if args.verified:
        return verified(names, idx, args)
    for name in names:
        rec = idx[name]
        md, fixed = corpus.load(name)
        res = {"m": rec["nr_rows"], "n": rec["nr_columns"], "nnz": rec["nnz"], "highs": rec.get("highs_objective"),
               "pin": rec.get("reference_pin"), "pin_tol": rec.get("reference_tolerance"), "runs": {}}
        for cname, cfg in CONFIGS:
            for ename, kind, block in ENGINES:
                res["runs"][f"{cname}/{ename}"] = solve(md, fixed, kind, block, cfg, args.max_pivots, args.seconds)
        # agreement
        for cname, _ in CONFIGS:
            objs = {e: res["runs"][f"{cname}/{e}"].get("objective") for e, _, _ in ENGINES}
            have = [v for v in objs.values() if v is not None]
            res[f"{cname}_all_optimal"] = len(have) == len(ENGINES)
            res[f"{cname}_engines_agree"] = bool(have) and len(have) == len(ENGINES) and max(rel(a, have[0]) for a in have) <= 1e-6
            if res["highs"] is not None and have:
                res[f"{cname}_vs_highs"] = max(rel(a, res["highs"]) for a in have)
            if res["pin"] is not None and have:
                res[f"{cname}_meets_pin"] = all(abs(a - res["pin"]) <= max(res["pin_tol"], 1e-9 * abs(res["pin"])) for a in have)
        results[name] = res
        line = " ".join(f"{k}={v.get('outcome')}:{v.get('pivots')}:{v.get('objective', float('nan')):.9g}:{v.get('seconds')}s" for k, v in res["runs"].items())
        print(f"{name} m={res['m']} n={res['n']} highs={res['highs']} pin={res['pin']} | {line}", flush=True)
        json.dump(results, open(args.out, "w"), indent=1)
    markdown(results, os.path.splitext(args.out)[0] + ".md")


def verified(names, idx, args):
    """engine.solve_verified on every file: outcome, the leg that was accepted, agreement with HiGHS."""
    out, rows = {}, ["| file | m x n | outcome | accepted leg | legs tried | objective | HiGHS (same LP) |", "|---|---|---|---|---|---|---|"]
    for name in names:
        rec = idx[name]
        md, fixed = corpus.load(name)
        t0 = time.perf_counter()
        oc, t, report = engine.solve_verified(md, seconds_per_leg=args.seconds)
        obj = t.objective_function_value() + fixed if (t is not None and oc == engine.OPTIMAL) else None
        if t is not None:
            t.close()
        # (2e-6: PILOT87's verified optimum is Netlib's published 301.71072827 to ten digits; HiGHS's value of the same LP is
        # 301.7103473, 1.1e-6 away)
        ok = obj is not None and rec.get("highs_objective") is not None and rel(obj, rec["highs_objective"]) <= 2e-6
        last = report["legs"][-1]
        out[name] = {"outcome": engine.OUTCOME_NAMES.get(oc, str(oc)), "verified": report["verified"], "objective": obj, "agrees_with_highs": ok,
                     "legs": report["legs"], "seconds": round(time.perf_counter() - t0, 2)}
        rows.append(f"| {name} | {rec['nr_rows']} x {rec['nr_columns']} | {out[name]['outcome']}{'' if report['verified'] else ' (unverified)'} | "
                    f"{last['data'] + ' / ' + last['config'] + ' / ' + last['engine'] if report['verified'] else ''} | {len(report['legs'])} | {'' if obj is None else f'{obj:.10g}'} | {rec.get('highs_objective'):.10g} |")
        print(name, out[name]["outcome"], report["verified"], ok, len(report["legs"]), out[name]["seconds"], flush=True)
        json.dump(out, open(args.out, "w"), indent=1)
    n_ok = sum(1 for v in out.values() if v["agrees_with_highs"] and v["verified"])
    n_wrong = sum(1 for v in out.values() if v["verified"] and v["outcome"] == "optimal" and not v["agrees_with_highs"])
    rows.append("")
    rows.append(f"{n_ok} of {len(out)} files: a verified optimum that agrees with HiGHS (2e-6 relative); {n_wrong} verified optima that do not.")
    with open(os.path.splitext(args.out)[0] + ".md", "w") as f:
        f.write("\n".join(rows) + "\n")
    print(rows[-1])


def markdown(results, path):
    """The table of DESIGN.md 6.3 / profiles/r04_corpus_sweep.md."""
    rows = ["| file | m x n | HiGHS (same LP) | reference pin | default: lu / tableau / revised | robust: lu / tableau / revised |", "|---|---|---|---|---|---|"]

    def cell(res, cfg):
        out = []
        for e in ("lu", "tableau", "revised"):
            r = res["runs"][f"{cfg}/{e}"]
            if r.get("outcome") == "optimal":
                ok = res["highs"] is not None and rel(r["objective"], res["highs"]) <= 1e-6
                out.append(f"{'ok' if ok else 'OPT?'} {r['pivots']}")
            else:
                out.append(f"{r.get('outcome')} {r.get('pivots', '')}".strip())
        return " / ".join(out)
    for name in sorted(results):
        res = results[name]
        rows.append(f"| {name} | {res['m']} x {res['n']} | {res['highs']:.10g} | {res['pin'] if res['pin'] is not None else ''} | {cell(res, 'default')} | {cell(res, 'robust')} |")
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
