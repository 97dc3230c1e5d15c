import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import rust_lp_amd  # noqa
    from rust_lp_amd import engine
    import corpus
    idx = corpus.index()
    for name in sys.argv[2:]:
        md, fixed = corpus.load(name)
        t = engine.Tableau(md, config=engine.robust_config())
        t0 = time.time(); tot = 0; oc = engine.RUNNING
        while time.time() - t0 < 45 and tot < 300000:
            done, oc = t.run(20000); tot += done
            if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE) or done == 0: break
        ok = oc == engine.OPTIMAL and abs(t.objective_function_value() + fixed - idx[name]["highs_objective"]) <= 1e-6 * max(1, abs(idx[name]["highs_objective"]))
        print(f"  {name}: {'OK' if ok else engine.OUTCOME_NAMES.get(oc, oc)} {tot} pivots {time.time()-t0:.1f}s {t.robust_stats()}", flush=True)
        t.close()
else:
    files = ["GREENBEA", "GREENBEB", "BNL1", "25FV47", "STAIR", "PILOT4", "TUFF", "MAROS", "MODSZK1", "SCFXM2", "PEROLD", "D2Q06C", "80BAU3B", "SCSD8", "WOODW", "PILOT-WE"]
    for g in ("1e-5", "1e-7", "1e-9", "0"):
        print("guard", g, flush=True)
        env = dict(os.environ, RELP_PIVOT_GUARD=g)
        subprocess.run([sys.executable, __file__, "child"] + files, env=env)
