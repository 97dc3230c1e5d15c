#!/usr/bin/env python3
"""bench.py -- simplex iterations/s of the MI355X pivot engine on the synthetic dense LP.

A "step" is one pass of the hot path (PRICE -> FTRAN -> RATIO -> UPDATE = one basis change) of
the simplex on the synthetic dense LP of BASELINE.json (rust-lp_amd/synthetic.py), inputs
resident in HBM before the timed region.  One JSON line on stdout (rank 0).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N            (no WORLD_SIZE in the environment: starts the line above as a child)

Timed region (warm-up, then the median of 5 runs).  The engines fold their pivots
into the stored representation once per `update_block` pivots (the "flush"), so a window that is not a
whole number of blocks would leave amortised work outside the clock.  The timed region is therefore
5 windows of `steps_per_window` = max(K, 4 * update_block) rounded UP to whole blocks, the first one
starting right after a flush; `steps` in the JSON is that actual pivot count per window, `value` =
steps / median window time.

Workloads (config.workload):
  dense10k  (default) m = 10,000 rows, n = 10,000 structural columns (+10,000 slacks), f64,
            SteepestDescent (Dantzig): the LP BASELINE.json's target is quoted on.
  c2        2,000 x 2,000 (BASELINE.json configs[1]); c4: 10,000 x 50,000 (configs[3]).
Engines (--engine):
  tableau   (default) dense tableau T = B^-1 [A | I] kept as (I + W S') T0: PRICE is one tableau row,
            FTRAN one tableau column per pivot; T0 += W R0 (m x K x n GEMM on the f64 matrix cores)
            every K pivots.  Same pivots as the revised engine (parity-tested).
  revised   explicit dense inverse `Carry<_, BasisInverseRows<_>>`: PRICE streams A (8 m n bytes) and
            FTRAN streams B^-1 (8 m^2 bytes) at every pivot -- the FTRAN/PRICE HBM-roofline numbers.
  At N = 1 the default run measures the tableau engine as `value` and adds the revised engine's
  numbers under "revised_engine" and inside "roofline".
N > 1: the same LP (strong scaling); the stored tableau columns are split contiguously over the
ranks, ONE all-gather of [key, j, d_j, alpha(m)] candidates per pivot over RCCL, everything else local.
Every N also reports "c4" (10,000 x 50,000, the shape whose flush is large enough for sharding to pay).
"""
import argparse
import json
import math
import os
import shutil
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {"dense10k": (10000, 10000, 20250002), "c2": (2000, 2000, 20250001), "c4": (10000, 50000, 20250003)}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F64_MFMA_PEAK_TF = 78.6    # FP64 matrix 78.6 TFLOP/s
WINDOWS = 5                # median of 5 runs
CPU_BUDGET_S = 20.0        # cpu_baseline: bounded sample (10-30 s of CPU work)


# ------------------------------------------------------------------------------------------------------------
# Pure helpers (no GPU, no torch): window plan, algorithmic bytes, JSON assembly.  tests/test_bench_json.py
# feeds these with profile dictionaries for K in {1, 20, 64, 200}.
# ------------------------------------------------------------------------------------------------------------
def coprime_stride(stride, block):
    """The per-pivot kernels of every `stride`-th pivot are bracketed by HIP events.  A stride that shares a factor with the
    update block samples the same few positions of a block over and over (64 on 64: always its first pivot, where PRICE runs
    on cold caches and the tableau engine's per-pivot kernels carry no pending updates): the next stride co-prime with the block
    walks through every position p = 0 .. K - 1."""
    stride = max(int(stride), 1)
    if block <= 1:
        return stride
    while math.gcd(stride, block) != 1:
        stride += 1
    return stride


def window_steps(requested, block):
    """Pivots per timed window: at least `requested`, at least 4 update blocks, a whole number of blocks."""
    requested = max(int(requested), 1)
    if block <= 0:
        return requested
    want = max(requested, 4 * block)
    return (want + block - 1) // block * block


def algorithmic_bytes(kind, m, n, world, block):
    """Algorithmic HBM bytes (and flops) per launch of the streaming kernels, and per pivot, for the local shard.
    DESIGN.md section 5 derives them; n = structural columns, the stored tableau has n + m columns."""
    if kind == "tableau":
        n_owned = (n + m + world - 1) // world
        p_avg = max(block - 1, 0) / 2.0                              # average number of pending pivots in a block
        per_pivot = 8.0 * n_owned * (p_avg + 1) + 8.0 * m * (p_avg + 1) + 16.0 * m * p_avg + 24.0 * m
        flush = 16.0 * m * n_owned
        return {"launch": {"flush": flush}, "flops": {"flush": 2.0 * m * n_owned * max(block, 1)},
                "pivot": per_pivot + (flush / block if block > 0 else 0.0),
                "pivot_formula": "8 n_s (p+1) [tableau row] + 8 m (p+1) [tableau column] + 16 m p [W update] + 24 m "
                                 "[b, alpha, basis] with p = (K-1)/2 pending pivots on average, + 16 m n_s / K [flush]; "
                                 "n_s = stored columns of this rank"}
    rows_local = (m + world - 1) // world
    n_local = (n + world - 1) // world
    launch = {"price": 8.0 * m * n_local, "ftran": 8.0 * rows_local * m + 16.0 * m,
              "update_inverse": 16.0 * rows_local * m, "flush": 16.0 * rows_local * m}
    if block > 0:
        p_avg = max(block - 1, 0) / 2.0
        per_pivot = launch["price"] + launch["ftran"] + 24.0 * m * p_avg + 8.0 * m * (p_avg + 1) + launch["flush"] / block
        formula = "8 m n_local [PRICE] + 8 m_local m [FTRAN] + 24 m p [W] + 8 m (p+1) [pivot row] + 16 m_local m / K [flush]"
    else:
        per_pivot = launch["price"] + launch["ftran"] + launch["update_inverse"] + 40.0 * m
        formula = "8 m n_local [PRICE] + 8 m_local m [FTRAN] + 16 m_local m [rank-1 update] + 40 m (SURVEY.md section 8d)"
    return {"launch": launch, "flops": {"flush": 2.0 * rows_local * m * max(block, 1)}, "pivot": per_pivot,
            "pivot_formula": formula}


def kernel_table(prof, alg_bytes, alg_flops):
    """prof: {kernel class: (bracketed launches, total ms)} -> per-class averages (+ GB/s, TFLOP/s where defined)."""
    kernels = {}
    for name, (cnt, ms) in (prof or {}).items():
        if cnt and cnt > 0 and ms > 0:
            avg_us = ms * 1e3 / cnt
            entry = {"launches": int(cnt), "avg_us": round(avg_us, 3)}
            if name in alg_bytes:
                entry["GBps"] = round(alg_bytes[name] / (avg_us * 1e-6) / 1e9, 1)
            if name in alg_flops:
                entry["TFLOPs"] = round(alg_flops[name] / (avg_us * 1e-6) / 1e12, 2)
            kernels[name] = entry
    return kernels


def load_traffic(workload, kernel):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            return json.load(open(tpath)).get(workload, {}).get(kernel)
        except Exception:
            return None
    return None


TRAFFIC_KEY = {"price": "price_all", "ftran": "ftran", "update_inverse": "update_inverse_vectors", "flush": "flush_apply"}


def kernel_roofline(kind, workload, kernels, alg):
    """The streaming kernel that dominates the pivot, priced against HBM.  None when no bracketed launch of a
    streaming kernel exists (events switched off)."""
    if kind == "tableau":
        fl = kernels.get("flush")
        if not fl or "GBps" not in fl:
            return None
        return {"kernel": "k_tab_flush_lds", "bound": "hbm", "achieved": fl["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(fl["GBps"] / HBM_PEAK_GBS, 4), "traffic": load_traffic(workload, "tab_flush_lds"),
                "avg_us": fl["avg_us"], "launches_timed": fl["launches"],
                "algorithmic_bytes_per_launch": alg["launch"]["flush"],
                "mfma": {"achieved": fl.get("TFLOPs"), "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s (f64 matrix)",
                         "frac": round((fl.get("TFLOPs") or 0.0) / F64_MFMA_PEAK_TF, 4)}}
    streaming = [k for k in ("price", "ftran", "update_inverse", "flush") if "GBps" in kernels.get(k, {})]
    if not streaming:
        return None
    dom = max(streaming, key=lambda k: kernels[k]["avg_us"] * kernels[k]["launches"])
    return {"kernel": "k_" + TRAFFIC_KEY[dom], "bound": "hbm", "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(kernels[dom]["GBps"] / HBM_PEAK_GBS, 4),
            "traffic": load_traffic(workload, TRAFFIC_KEY[dom]), "avg_us": kernels[dom]["avg_us"],
            "launches_timed": kernels[dom]["launches"], "algorithmic_bytes_per_launch": alg["launch"][dom]}


def pivot_roofline(alg, ms_per_step):
    """Whole pivot against HBM: algorithmic bytes of one pivot (flush share included) / measured time per pivot."""
    gbps = alg["pivot"] / (ms_per_step * 1e-3) / 1e9 if ms_per_step > 0 else None
    return {"algorithmic_bytes": round(alg["pivot"]), "formula": alg["pivot_formula"],
            "achieved": round(gbps, 1) if gbps is not None else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbps / HBM_PEAK_GBS, 4) if gbps is not None else None}


def section(res, workload, m, n, world, kind):
    """One measured engine -> its JSON section (value, window statistics, kernels, rooflines)."""
    steps = res["steps"]
    med = statistics.median(res["window_s"])
    ms_per_step = med * 1e3 / steps
    alg = algorithmic_bytes(kind, m, n, world, res["block"])
    kernels = kernel_table(res.get("prof"), alg["launch"], alg["flops"])
    roof = kernel_roofline(kind, workload, kernels, alg)
    out = {"value": steps / med, "unit": "iterations/s", "steps": steps, "windows": len(res["window_s"]),
           "ms_per_step": ms_per_step, "window_ms": [round(w * 1e3, 4) for w in res["window_s"]],
           "update_block": res["block"], "kernels": kernels, "roofline": roof,
           "pivot_roofline": pivot_roofline(alg, ms_per_step), "objective_after_run": res.get("objective"),
           "kernel_event_stride": res.get("event_stride")}
    return out


def assemble(args_ns, world, primary_kind, primary, secondary=None, c2=None, c4=None, sparse=None, cpu=None,
             loop_kind=None):
    """The ONE JSON line.  `primary` etc. are outputs of section(); any of the optional parts may be None, and no
    kernel class is assumed to be present."""
    m, n, seed = WORKLOADS[args_ns.workload]
    roofline = dict(primary["roofline"]) if primary.get("roofline") else \
        {"kernel": None, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
         "note": "no bracketed launch of a streaming kernel in the timed windows (kernel events off)"}
    roofline["pivot"] = primary["pivot_roofline"]
    if secondary is not None:
        k2 = secondary.get("kernels", {})
        roofline["revised_engine"] = {
            name: {"kernel": "k_" + TRAFFIC_KEY[name], "achieved": k2[name]["GBps"], "unit": "GB/s", "peak": HBM_PEAK_GBS,
                   "frac": round(k2[name]["GBps"] / HBM_PEAK_GBS, 4), "avg_us": k2[name]["avg_us"],
                   "traffic": load_traffic(args_ns.workload, TRAFFIC_KEY[name])}
            for name in ("ftran", "price", "update_inverse", "flush") if "GBps" in k2.get(name, {})}
        roofline["revised_engine"]["pivot"] = secondary["pivot_roofline"]
    out = {
        "metric": "simplex iterations/sec", "value": primary["value"], "unit": "iterations/s", "n_gpus": world,
        "steps": primary["steps"], "steps_requested": args_ns.steps, "warmup": args_ns.warmup, "ms_per_step": primary["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args_ns.workload}: synthetic dense LP {m}x{n} f64 (+{m} slack columns), SteepestDescent, "
                               + ("dense tableau with blocked f64-MFMA updates" if primary_kind == "tableau"
                                  else "explicit dense basis inverse"),
                   "m": m, "n": n, "seed": seed, "engine": primary_kind, "update_block": primary["update_block"],
                   "steps_note": "--steps is honoured as AT LEAST: a timed window is a whole number of update blocks (max(steps, 4 blocks) "
                                 "rounded up to the block), because a window that ends inside a block would leave the block's one "
                                 "O(m n) flush out of the time; `steps` is what was timed, `steps_requested` what was asked for",
                   "parallelism": "single GPU" if world == 1 else
                   (f"stored tableau columns sharded x{world}, one all-gather per pivot" if primary_kind == "tableau"
                    else f"columns of A and rows of B^-1 sharded x{world}")},
        "timing": {"steps_requested": args_ns.steps, "windows": primary["windows"], "window_ms": primary["window_ms"],
                   "statistic": "median window; every window is a whole number of update blocks and starts right after a flush"},
        "roofline": roofline, "kernels": primary["kernels"],
        "kernel_event_stride": primary.get("kernel_event_stride"),
        "objective_after_run": primary.get("objective_after_run"),
        "reference_iteration_algorithmic_bytes": 8.0 * m * n + 24.0 * m * m,
    }
    if loop_kind:
        out["config"]["shard_loop"] = loop_kind
    if secondary is not None:
        out["revised_engine"] = dict(secondary, note="explicit dense inverse (Carry<_, BasisInverseRows<_>>): PRICE streams A, "
                                     "FTRAN streams B^-1 every pivot; GB/s = algorithmic bytes / HIP-event duration")
    if c2 is not None:
        out["c2"] = c2
    if c4 is not None:
        out["c4"] = c4
    if sparse is not None:
        out["sparse_engine"] = sparse
    if world == 1:
        out["cpu_baseline"] = cpu
    return out


# ------------------------------------------------------------------------------------------------------------
# CPU baseline and the sparse path
# ------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(m, n, seed, warmup, budget_s=None, max_pivots=2000):
    """The C (f64) restatement of the reference path (oracle/relp_f64.c, kind "port"), one core, timed on the same
    LP from iteration `warmup` on until ~budget_s of CPU time (independent of --steps)."""
    from rust_lp_amd import MatrixData, synthetic
    budget_s = CPU_BUDGET_S if budget_s is None else budget_s
    from oracle import relp_f64
    lp = synthetic.dense_lp(m, n, seed)
    md = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]).ensure_csc()
    md.dense = None
    del lp
    ref = relp_f64.OracleF64(md)
    ref.run(max_iters=1 << 40, through_phases=False, record=False)      # empty phase 1 -> phase 2
    ref.run(max_iters=warmup, record=False)
    done, t0 = 0, time.perf_counter()
    while done < max_pivots and time.perf_counter() - t0 < budget_s:
        ref.run(max_iters=5, record=False)
        done += ref.last_n_done
        if ref.last_n_done == 0:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt if dt > 0 and done else None, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{done} pivots (iterations {warmup}..{warmup + done}) of the same {m}x{n} LP, "
                      f"oracle/relp_f64.c (f64 restatement of Carry<_, BasisInverseRows>), {dt:.1f} s",
            "nproc": os.cpu_count(), "cpu_model": cpu_model(),
            "reference_toolchain": "cargo " + ("present (not used: the crate needs a 2021 nightly and crates.io)"
                                               if shutil.which("cargo") else "absent on this host")}


def sparse_path(events, with_cpu):
    """The sparse path (BASELINE.json configs[2]): Netlib 25FV47 through the build's MPS reader, presolve and
    standardisation, solved to optimality by the LU engine.  The whole solve is timed; it is latency-bound (the
    factors are a few hundred KB), so the HBM figure is reported for what it is."""
    from rust_lp_amd import engine, general_form, mps
    path = os.path.join(ROOT, "tests", "golden", "mps", "netlib", "25FV47.SIF")
    if not os.path.exists(path):
        return None
    gf = general_form.GeneralForm.from_mps(mps.import_file(path, True))
    md = gf.to_matrix_data(gf.derive_matrix_data_exact())
    out = {"reference_objective": 5.5018459e+03, "tolerances": "relp_default_config"}

    def solve(kind, **cfg):
        t = engine.Tableau(md, engine=kind, **cfg)
        if events and kind == engine.ENGINE_LU and not cfg:
            t.profile_enable(True, 40000, 1)
        t0 = time.perf_counter()
        outcome = t.solve_relaxation()
        dt = time.perf_counter() - t0
        its = t.iterations()
        res = {"outcome": engine.OUTCOME_NAMES.get(outcome), "pivots": its, "value": its / dt if dt > 0 else None,
               "unit": "iterations/s", "seconds": dt, "objective": t.objective_function_value() + float(gf.fixed_cost),
               "degenerate_pivots": t.degenerate_pivots()}
        return t, res

    t, res = solve(engine.ENGINE_LU)
    stats = t.lu_stats()
    prof = t.profile_read() if events else {}
    phases = t.lu_phase_cycles()
    block = t.update_block()
    t.close()
    out.update(res)
    out.update({"workload": f"Netlib 25FV47 after presolve: {stats['m']} rows, {md.nr_normal} structural columns, "
                            f"{len(md.values)} nonzeros; FirstProfitableWithMemory / SteepestDescent, whole two-phase solve",
                "engine": "lu (Forrest-Tomlin update on the device, persistent pivot kernel)", "update_block": block,
                "refactorisations": stats["refactorisations"],
                "last_factor": {k: stats[k] for k in ("nnz_l", "nnz_u", "levels_l", "levels_u")}})
    kt = {name: {"launches": cnt, "avg_us": round(ms * 1e3 / cnt, 3)} for name, (cnt, ms) in prof.items() if cnt}
    if kt:
        out["kernels"] = kt
    if "ft_run" in kt and res["pivots"]:
        runs = kt["ft_run"]["launches"]
        # k_ft_run launches plus, with the look-ahead refactorisation, one k_ft_replay per refactorisation (bracketed as "flush")
        out["kernel_launches_per_pivot"] = round((runs + kt.get("flush", {}).get("launches", 0)) / res["pivots"], 4)
        out["pivot_kernel_us_per_pivot"] = round(kt["ft_run"]["avg_us"] * runs / res["pivots"], 2)
    tot = sum(phases.values())
    if tot:
        out["pivot_kernel_phase_share"] = {k: round(v / tot, 4) for k, v in phases.items() if v}
        out["pivot_kernel_clocks_per_pivot"] = round(tot / max(res["pivots"], 1))
    # bytes of one pivot of the sparse engine = PRICE 12 nnz(A) + 4 (n + 1) + 8 m; FTRAN and BTRAN
    # 12 (nnz L + nnz U) + 16 m each; UPDATE ~ 40 m.  The working set lives in L2 / LDS: latency-bound, reported as such
    nnz_a, n_cols, mm = len(md.values), md.nr_normal, stats["m"]
    pivot_bytes = 12.0 * nnz_a + 4.0 * (n_cols + 1) + 8.0 * mm + 2 * (12.0 * (stats["nnz_l"] + stats["nnz_u"]) + 16.0 * mm) + 40.0 * mm
    out["algorithmic_bytes_per_pivot"] = round(pivot_bytes)
    out["achieved_GBps"] = round(pivot_bytes * res["value"] / 1e9, 3) if res["value"] else None
    out["hbm_frac"] = round(pivot_bytes * res["value"] / 1e9 / HBM_PEAK_GBS, 6) if res["value"] else None
    out["bound"] = "latency (dependent sparse steps over L2 / LDS-resident data: ~130 levels of triangular solves per pivot)"
    # the reference's own refactorisation cadence (lower_upper/mod.rs:199-202: updates.len() > 10)
    t, res11 = solve(engine.ENGINE_LU, update_block=11)
    res11["refactorisations"] = t.lu_stats()["refactorisations"]
    t.close()
    out["reference_cadence_update_block_11"] = res11
    # the same cadence with the pipelined look-ahead (RELP_LU_PIPELINE_SHORT, relp_engine_lu.cpp: run_ft): the kernel pivots on into a
    # tail twice as long while the host factorises; the factors lag one interval behind, the device does not wait for a factorisation
    os.environ["RELP_LU_PIPELINE_SHORT"] = "1"
    try:
        t, res11p = solve(engine.ENGINE_LU, update_block=11)
        st = t.lu_stats()
        res11p["refactorisations"] = st["refactorisations"]; res11p["lookahead_installs"] = st["lookahead_installs"]
        t.close()
    finally:
        del os.environ["RELP_LU_PIPELINE_SHORT"]
    out["reference_cadence_update_block_11_pipelined"] = res11p
    # the same LP on the explicit-inverse engine (m = 790: B^-1 is 5 MB) and on the dense tableau engine
    t, res = solve(engine.ENGINE_REVISED, update_block=0)
    res["reinversions"] = t.reinversions()
    t.close()
    out["explicit_inverse_engine"] = res
    t, res = solve(engine.ENGINE_TABLEAU, update_block=32)
    res["retabulations"] = t.reinversions()
    t.close()
    out["tableau_engine"] = res
    if with_cpu:
        # the reference's CPU path for this config is `Carry<_, LUDecomposition<_>>` (src/bin/main.rs:52): the f64 port of that
        # back-end (oracle/relp_f64_lu.h: Markowitz LU, eta file, re-inverted whenever more than 10 updates are pending) is the
        # baseline; the sparse-rows back-end (`BasisInverseRows`) is timed beside it
        from oracle import relp_f64
        ref = relp_f64.OracleF64(md, basis_inverse=1, lu_threshold=0.1)
        c0 = time.perf_counter()
        ref.run(max_iters=3000)
        cdt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": 3000 / cdt, "unit": "iterations/s", "cores": 1, "kind": "port",
                               "back_end": "LUDecomposition + eta file, refactorised at the reference's cadence (lower_upper/mod.rs:199-202)",
                               "sample": f"first 3000 pivots of the same LP, oracle/relp_f64_lu.h, {cdt:.1f} s",
                               "refactorisations": ref.lu_stats()["refactorisations"]}
        # the LU engine at the same cadence walks the same pivots (the instrument VERDICT r3 asked for: same back-end, same cadence)
        t = engine.Tableau(md, engine=engine.ENGINE_LU, update_block=11, trace_capacity=3000)
        total = 0
        while total < 3000:
            done, oc = t.run(3000 - total)
            total += done
            if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE) or done == 0:
                break
        tr = t.trace()
        same = next((k for k, (a, b) in enumerate(zip(tr, ref.trace)) if tuple(a) != tuple(b)), min(len(tr), len(ref.trace)))
        out["cpu_baseline"]["lu_engine_at_the_same_cadence_walks_the_same_pivots_for"] = same
        t.close()
        rows = relp_f64.OracleF64(md)
        c0 = time.perf_counter()
        rows.run(max_iters=3000, record=False)
        rdt = time.perf_counter() - c0
        out["cpu_baseline_rows_back_end"] = {"value": 3000 / rdt, "unit": "iterations/s", "cores": 1, "kind": "port",
                                             "back_end": "BasisInverseRows (explicit inverse as sparse rows, never refactorised)",
                                             "sample": f"first 3000 pivots of the same LP, oracle/relp_f64.c, {rdt:.1f} s"}
    return out


def _load_fixture(rel, fixed):
    """tests/golden/mps/<rel> through the build's MPS reader, presolve and standardisation -> (general form, MatrixData)."""
    from rust_lp_amd import general_form, mps
    path = os.path.join(ROOT, "tests", "golden", "mps", rel)
    if not os.path.exists(path):
        return None, None
    gf = general_form.GeneralForm.from_mps(mps.import_file(path, fixed))
    return gf, gf.to_matrix_data(gf.derive_matrix_data_exact())


def _timed_run(md, gf, kind, max_pivots=1 << 40, **cfg):
    """One engine on one LP: whole solve (or the first `max_pivots` pivots), wall clock around relp_run."""
    from rust_lp_amd import engine
    reinv = cfg.pop("reinversion_interval", None)
    t = engine.Tableau(md, engine=kind, **cfg)
    if reinv is not None and kind != engine.ENGINE_LU:
        t.set_reinversion_interval(reinv)
    t0 = time.perf_counter()
    total, oc = 0, engine.RUNNING
    while total < max_pivots:
        done, oc = t.run(max_pivots - total)
        total += done
        if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE) or done == 0:
            break
    dt = time.perf_counter() - t0
    res = {"outcome": engine.OUTCOME_NAMES.get(oc), "pivots": total, "value": total / dt if dt > 0 else None,
           "unit": "iterations/s", "seconds": round(dt, 4), "degenerate_pivots": t.degenerate_pivots(),
           "objective": t.objective_function_value() + float(gf.fixed_cost), "rows": t.nr_rows(), "columns": t.nr_columns()}
    return t, res


ENGINE_LABELS = {"lu": 2, "revised": 0, "tableau": 1}


def config_one(with_cpu):
    """BASELINE.json configs[0]: tests/burkardt adlittle.mps.  The exact (rational) path on the host -- oracle/relp_exact.py,
    the restatement of `Carry<RationalBig, BasisInverseRows>` the reference's own test runs (tests/burkardt/test.rs:34-54) --
    beside the f64 GPU engines, whose pivot sequence must equal the exact one."""
    from rust_lp_amd import engine
    gf, md = _load_fixture("burkardt/adlittle.mps", False)
    if md is None:
        return None
    out = {"workload": f"burkardt adlittle.mps after presolve: {md.nr_rows} rows, {md.nr_columns} columns; two phases to optimality",
           "reference_objective": "24975305659811992079614961229/120651674036153428931840"}
    exact_trace = None
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fractions import Fraction
        from oracle import relp_exact as ox
        from lp_files import exact_solve, load
        gfx, ex, mdx, emd = load("burkardt/adlittle.mps")
        tr = []
        c0 = time.perf_counter()
        status, obj, _ = exact_solve(gfx, emd, ox.BasisInverseRows, trace=tr.append)
        cdt = time.perf_counter() - c0
        exact_trace = [(e["phase"], e["entering"], e["row"], e["leaving"]) for e in tr]
        out["exact_cpu"] = {"value": len(tr) / cdt, "unit": "iterations/s", "cores": 1, "kind": "port", "pivots": len(tr),
                            "seconds": round(cdt, 3), "objective_is_the_reference_pin":
                            obj == Fraction(24975305659811992079614961229, 120651674036153428931840),
                            "sample": "whole solve, oracle/relp_exact.py (fractions.Fraction restatement of the reference's exact path)"}
    for label, kind in ENGINE_LABELS.items():
        t, res = _timed_run(md, gf, kind, trace_capacity=4096, update_block={"lu": 11, "tableau": 32}.get(label, -1))
        if exact_trace is not None:
            res["trace_identical_to_exact"] = t.trace() == exact_trace
        t.close()
        out[label] = res
    return out


def config_five():
    """BASELINE.json configs[4]: MIPLIB LP relaxations.  50v-10 to optimality (the reference's active test, tests/miplib/test.rs:4)
    and the degenerate stress acc-tight4 (`#[ignore]`d there as too expensive): the first 100,000 pivots per engine, with the
    number of degenerate pivots (ratio exactly 0) SURVEY.md 8d asks for."""
    out = {}
    gf, md = _load_fixture("miplib/50v-10.mps", False)
    if md is not None:
        sec = {"workload": f"MIPLIB 50v-10 relaxation after presolve: {md.nr_rows} rows, {md.nr_columns} columns; to optimality",
               "reference_objective": 2879.065687}
        for label, kind in ENGINE_LABELS.items():
            t, res = _timed_run(md, gf, kind)
            t.close()
            sec[label] = res
        out["50v-10"] = sec
    gf, md = _load_fixture("miplib/acc-tight4.mps", False)
    if md is not None:
        sec = {"workload": f"MIPLIB acc-tight4 relaxation after presolve: {md.nr_rows} rows, {md.nr_columns} columns; first 100,000 "
                           "pivots (phase 1 needs more than 400,000)"}
        for label, kind in ENGINE_LABELS.items():
            t, res = _timed_run(md, gf, kind, max_pivots=100000 if label != "revised" else 30000)
            t.close()
            sec[label] = res
        out["acc-tight4"] = sec
    return out


def sparse_large():
    """The LU engine beyond one CU's LDS-resident work vectors (the persistent kernel's second layout: x and -pi in LDS, spike,
    permutations and eta pool in L2): the reference's big Netlib files -- which it `#[ignore]`s as too expensive -- next to
    the dense engines on the same LP.  GREENBEA / GREENBEB / 80BAU3B run with the f64 safeguards their pins need
    (tests/test_gpu_big_pins.py), DFL001 for its first 40,000 pivots."""
    from rust_lp_amd import engine
    cases = [("GREENBEA", -0.72555248129845987457557870574845e8, dict(ratio_rule=1, artificial_removal=1), 1 << 40),
             ("GREENBEB", -0.43022602612065867539213672544432e7, dict(ratio_rule=1, artificial_removal=1), 1 << 40),
             ("80BAU3B", 9.872241924e+05, dict(artificial_removal=1), 1 << 40),
             ("DFL001", None, dict(ratio_rule=1, artificial_removal=1), 40000)]
    out = {}
    for name, pin, cfg, budget in cases:
        gf, md = _load_fixture(f"netlib/{name}.SIF", True)
        if md is None:
            continue
        sec = {"workload": f"Netlib {name} after presolve: {md.nr_rows} rows, {md.nr_columns} columns"
                           + ("" if budget >= 1 << 40 else f"; first {budget} pivots"),
               "reference_pin": pin, "config": cfg}
        for label, kind in ENGINE_LABELS.items():
            extra = dict(cfg)
            if label != "lu":                   # (as tests/test_gpu_big_pins.py: an f64 inverse / tableau that is only ever updated
                extra["reinversion_interval"] = 200 if name == "80BAU3B" else 1000     # does not survive these files)
            if label == "tableau":
                extra["update_block"] = 32
            t, res = _timed_run(md, gf, kind, max_pivots=budget, **extra)
            if label == "lu":
                st = t.lu_stats()
                res["last_factor"] = {k: st[k] for k in ("nnz_l", "nnz_u", "levels_l", "levels_u")}
                res["refactorisations"] = st["refactorisations"]
                try:
                    ph = t.lu_phase_cycles()
                    res["pivot_kernel_clocks_per_pivot"] = round(sum(ph.values()) / max(res["pivots"], 1))
                except engine.RelpError:
                    res["pivot_kernel_clocks_per_pivot"] = None       # product-form fallback: no persistent kernel
            if pin is not None and res["outcome"] == "optimal":
                res["pin_error"] = res["objective"] - pin
            t.close()
            sec[label] = res
        out[name] = sec
    return out


def sparse_scale(with_cpu=True):
    """Where the sparse engine is the engine to take: a multi-commodity flow LP of the KEN / PDS shape
    (synthetic.multicommodity_lp: 12 commodities on 4,000 nodes / 16,000 arcs -> 63,988 rows x 255,988 columns, 3 entries per
    column), far beyond what one CU's LDS holds per row.  The persistent pivot kernel runs in its third layout (x, -pi and the
    slot table in L2; tests/test_gpu_lu_layout2.py); beside it the product-form loop the engine used to fall back to there
    (RELP_FT_BIG=0 leaves no layout), the dense tableau engine (131 GB of tableau + 98 GB of dense A) and the explicit inverse
    (33 GB rewritten at every pivot), each over its first pivots of phase 1 -- the same pivots on every engine."""
    from rust_lp_amd import MatrixData, engine, synthetic
    v, e, k = 4000, 16000, 12
    md = MatrixData.from_sparse_dict(synthetic.multicommodity_lp(v, e, k, 7))

    class NoFixedCost:
        fixed_cost = 0.0
    out = {"workload": f"synthetic.multicommodity_lp({v}, {e}, {k}, seed 7): {md.nr_rows} rows, {md.nr_columns} columns, "
                       f"{len(md.values)} entries; FirstProfitableWithMemory, phase 1, first pivots"}
    legs = [("lu", engine.ENGINE_LU, 20000, None), ("lu_product_form_fallback", engine.ENGINE_LU, 1000, "0"),
            ("tableau", engine.ENGINE_TABLEAU, 5000, None), ("revised", engine.ENGINE_REVISED, 250, None)]
    for label, kind, budget, force in legs:
        if force is not None:
            os.environ["RELP_FT_BIG"] = force
        try:
            t0 = time.perf_counter()
            t, res = _timed_run(md, NoFixedCost, kind, max_pivots=budget, trace_capacity=1024)
            res["create_and_run_seconds"] = round(time.perf_counter() - t0, 3)
        finally:
            if force is not None:
                del os.environ["RELP_FT_BIG"]
        res["first_pivots"] = [list(p) for p in t.trace()[:250]]
        if kind == engine.ENGINE_LU:
            res["kernel_layout"] = t.lu_kernel_layout()
            st = t.lu_stats()
            res["refactorisations"] = st["refactorisations"]
            res["last_factor"] = {q: st[q] for q in ("nnz_l", "nnz_u", "levels_l", "levels_u")}
            if res["kernel_layout"]["persistent_kernel"]:
                ph = t.lu_phase_cycles()
                tot = sum(ph.values())
                res["pivot_kernel_clocks_per_pivot"] = round(tot / max(res["pivots"], 1))
                res["pivot_kernel_phase_share"] = {q: round(c / max(tot, 1), 4) for q, c in ph.items()}
        t.close()
        out[label] = res
    first = out["lu"].pop("first_pivots")
    for label, _, _, _ in legs[1:]:
        out[label]["first_250_pivots_equal_the_lu_engines"] = out[label].pop("first_pivots") == first
    if with_cpu:
        # the f64 CPU port (oracle/relp_f64.c: explicit inverse as sparse rows, one core) over the first 5,000 pivots, and the LU
        # engine once more over the same stretch: same pivots, same objective
        from oracle import relp_f64
        ref = relp_f64.OracleF64(md)
        t0 = time.perf_counter()
        ref.run(max_iters=5000)
        cdt = time.perf_counter() - t0
        t = engine.Tableau(md, engine=engine.ENGINE_LU, trace_capacity=5000)
        total = 0
        t1 = time.perf_counter()
        while total < 5000:
            done, oc = t.run(5000 - total)
            total += done
            if oc not in (engine.RUNNING, engine.PHASE_ONE_DONE) or done == 0:
                break
        gdt = time.perf_counter() - t1
        # the reference's own back-end for sparse LPs, `LUDecomposition` (oracle/relp_f64_lu.h), at this size: its left-hand solves
        # scan every later / earlier column per entry of the work vector and its Markowitz search every remaining entry per step
        # (lower_upper/mod.rs:314-347, pivoting.rs:45-81), so a bounded sample of ~20 s is all there is to time
        lu_ref = relp_f64.OracleF64(md, basis_inverse=1, lu_threshold=0.1)
        l0 = time.perf_counter()
        lu_done = 0
        while time.perf_counter() - l0 < CPU_BUDGET_S and lu_done < 5000:
            lu_ref.run(max_iters=5, record=False)
            lu_done += lu_ref.last_n_done
            if lu_ref.last_n_done == 0:
                break
        ldt = time.perf_counter() - l0
        out["cpu_baseline_lu_back_end"] = {"value": lu_done / ldt if ldt > 0 else None, "unit": "iterations/s", "cores": 1, "kind": "port",
                                           "back_end": "LUDecomposition + eta file at the reference's cadence",
                                           "sample": f"first {lu_done} pivots of the same LP, oracle/relp_f64_lu.h, {ldt:.1f} s"}
        out["cpu_baseline"] = {"value": len(ref.trace) / cdt, "unit": "iterations/s", "cores": 1, "kind": "port",
                               "back_end": "BasisInverseRows (the faster of the reference's two back-ends here; the LU one is cpu_baseline_lu_back_end)",
                               "lu_engine_value_over_sample": total / gdt,
                               "sample": f"first 5000 pivots of the same LP, oracle/relp_f64.c, {cdt:.1f} s",
                               "objective_after_sample": ref.objective,
                               "lu_engine_objective_after_sample": t.objective_function_value(),
                               "lu_engine_takes_the_same_5000_pivots": [list(p) for p in t.trace()] == [list(p) for p in ref.trace]}
        t.close()
    out["lu_over_fallback"] = round(out["lu"]["value"] / out["lu_product_form_fallback"]["value"], 2)
    out["lu_over_tableau"] = round(out["lu"]["value"] / out["tableau"]["value"], 2)
    return out


def sparse_replicas(counts=(1, 8, 32, 64), fixture="netlib/25FV47.SIF", fixed=True, device=None, device_factorisation=False):
    """"LU / eta engine -- replicas only".  R independent LU handles on ONE GPU, each with its own stream, its own
    persistent pivot workgroup (one CU) and its own refactorisations, each solving the same LP to optimality from its own host
    thread (the C calls release the GIL); every replica must walk the solo run's pivot sequence.  Aggregate iterations/s =
    all pivots / wall time of the slowest replica.  This is what an N-GPU run of the sparse path consists of (N x R replicas,
    no collective), and what the 255 CUs the pivot kernel leaves idle are for; it is NOT the headline metric (one LP, one
    solve) and is labelled so."""
    import threading
    from rust_lp_amd import engine
    gf, md = _load_fixture(fixture, fixed)
    out = {"workload": f"{fixture} on R independent LU engines of one GPU (own stream, own persistent kernel, own refactorisations"
                       + (" ON THE DEVICE)" if device_factorisation else " on host threads)"),
           "unit": "iterations/s (aggregate over the replicas)", "replicas": {}}
    solo_trace = None
    for R in counts:
        kw = dict(engine=engine.ENGINE_LU, trace_capacity=1 << 15)
        if device is not None:
            kw["device"] = device
        ts = [engine.Tableau(md, **kw) for _ in range(R)]
        if device_factorisation:                      # no host thread per refactorisation: k_lu_factor + k_lu_schedules on the replica's stream
            for t in ts:
                t.lu_set_device_factorisation(True)
        res = [None] * R
        go = threading.Barrier(R + 1)

        def work(i):
            go.wait()
            t0 = time.perf_counter()
            oc = ts[i].solve_relaxation()
            res[i] = (oc, time.perf_counter() - t0)
        th = [threading.Thread(target=work, args=(i,)) for i in range(R)]
        for x in th:
            x.start()
        go.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        wall = time.perf_counter() - t0
        pivots = [t.iterations() for t in ts]
        traces = [t.trace() for t in ts]
        if solo_trace is None:
            solo_trace = traces[0]
        objective = ts[0].objective_function_value() + float(gf.fixed_cost)
        for t in ts:
            t.close()
        out["replicas"][str(R)] = {"value": sum(pivots) / wall, "seconds": wall, "pivots_per_replica": pivots[0],
                                   "all_optimal": all(r[0] == engine.OPTIMAL for r in res),
                                   "every_replica_walks_the_solo_pivots": all(tr == solo_trace for tr in traces),
                                   "slowest_over_fastest_replica": max(r[1] for r in res) / max(min(r[1] for r in res), 1e-9),
                                   "objective": objective}
    one = out["replicas"].get("1", {}).get("value")
    if one:
        out["scaling_over_one_replica"] = {k: round(v["value"] / one, 2) for k, v in out["replicas"].items()}
    return out


# ------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as a fresh child process (nothing in
    this process has touched the GPU yet) and relay its single JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, stdout=subprocess.PIPE)
    sys.stdout.buffer.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="dense10k", choices=sorted(WORKLOADS))
    ap.add_argument("--engine", default="default", choices=["default", "revised", "tableau", "lu"],
                    help="default = tableau as the measured engine (+ the revised engine's numbers at N = 1); lu = the sparse path as "
                         "independent replicas (SURVEY.md 8e: replicas only), --replicas per GPU, no collective in the data path")
    ap.add_argument("--replicas", type=int, default=8, help="--engine lu: LU engines per GPU")
    ap.add_argument("--no-replicas", action="store_true", help="skip the sparse_engine.replicas section (1 / 8 / 32 / 64 LU engines on one GPU)")
    ap.add_argument("--replicas-section", action="store_true", help=argparse.SUPPRESS)      # child process of the replicas section
    ap.add_argument("--update-block", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sparse", action="store_true", help="skip the sparse-path (LU engine, Netlib 25FV47) section")
    ap.add_argument("--no-c2", action="store_true", help="skip the configs[1] (dense 2,000 x 2,000) section")
    ap.add_argument("--no-c4", action="store_true", help="skip the configs[3] (dense 10,000 x 50,000) section")
    ap.add_argument("--no-c1", action="store_true", help="skip the configs[0] (adlittle, exact CPU path beside the f64 engines) section")
    ap.add_argument("--no-scale", action="store_true", help="skip the 63,988-row multi-commodity section (LU engine, third kernel layout)")
    ap.add_argument("--no-c5", action="store_true", help="skip the configs[4] (MIPLIB relaxations, degenerate pivot counts) section")
    ap.add_argument("--quick", action="store_true",
                    help="the line's schema in a fraction of the time (tests/test_bench_gpu.py): no c4 / c5 / sparse.large / "
                         "sparse.scale / sparse.replicas, CPU samples of 3 s; the headline, roofline, cpu_baseline, c1, c2 and the 25FV47 "
                         "section are measured as always")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--event-stride", type=int, default=64,
                    help="bracket the kernels of every n-th pivot with HIP events (1 = every pivot)")
    ap.add_argument("--shard-loop", default="native", choices=["native", "python"],
                    help="N > 1: pivot loop inside the library calling RCCL itself, or the Python loop over torch.distributed")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU loop (torch.distributed collectives) even at N = 1 (rehearsal)")
    args = ap.parse_args()

    # R engines on R streams run side by side only as far as the HIP runtime has hardware queues for them: its default of 4 capped
    # eight replicas at 2.7 x one (profiles/r04_replicas.md); 16 queues: 7.5 x.  The variable is read when the runtime initialises,
    # so the replicas run in a process of their own (the headline is measured with the runtime's defaults)
    if args.engine == "lu" or args.replicas_section:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    if args.replicas_section:
        sys.stdout.flush()
        fd = os.dup(1)
        os.dup2(2, 1)
        import rust_lp_amd  # noqa: F401
        os.write(fd, (json.dumps(sparse_replicas()) + "\n").encode())
        return
    if args.quick:
        global CPU_BUDGET_S
        CPU_BUDGET_S = 3.0
        args.no_c4 = args.no_c5 = args.no_scale = args.no_replicas = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE JSON line: anything a library prints there on the way (RCCL writes its
    # version banner to stdout at communicator creation) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import rust_lp_amd  # noqa: F401
    from rust_lp_amd import MatrixData, engine, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: RELP_BENCH_REHEARSE=1 puts every rank on device 0 and uses gloo for the
    # exchange (RCCL refuses two ranks on one device); the multi-rank code path is otherwise the same
    rehearse = os.environ.get("RELP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    K, W = max(args.steps, 1), max(args.warmup, 0)
    lib = engine.load_library()
    if args.engine == "lu":
        # the sparse path on N GPUs: replicas only -- every rank solves Netlib 25FV47 on `--replicas` independent LU
        # engines, no data-path collective; barrier + synchronize on both sides, MAX over ranks, value = all pivots / that time
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rep = sparse_replicas(counts=(args.replicas,), device=local_rank)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r = rep["replicas"][str(args.replicas)]
        pivots = torch.tensor([float(r["pivots_per_replica"] * args.replicas)], dtype=torch.float64, device=dev if not rehearse else "cpu")
        secs = torch.tensor([r["seconds"]], dtype=torch.float64, device=dev if not rehearse else "cpu")
        if sharded:
            dist.barrier()
            dist.all_reduce(pivots, op=dist.ReduceOp.SUM)
            dist.all_reduce(secs, op=dist.ReduceOp.MAX)
        if rank == 0:
            line = {"metric": "simplex iterations/sec (sparse path: independent LU-engine replicas, SURVEY.md 8e)", "value": pivots.item() / secs.item(),
                    "unit": "iterations/s", "n_gpus": world, "steps": int(r["pivots_per_replica"]), "warmup": 0,
                    "ms_per_step": 1e3 * secs.item() / max(r["pivots_per_replica"], 1), "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "f64", "data": "Netlib 25FV47 (tests/golden/mps), whole two-phase solves",
                    "config": {"workload": "sparse_replicas", "replicas_per_gpu": args.replicas, "engine": "lu", "parallelism": f"replicas x{world}",
                               "note": "not BASELINE's headline metric (one LP per engine); --steps is the pivot count of a whole solve"},
                    "replicas": r, "seconds_including_create": dt}
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        if sharded:
            dist.destroy_process_group()
        return
    events = not args.no_kernel_events
    primary_kind = "tableau" if args.engine in ("default", "tableau") else "revised"
    loop_kind = {}

    def measure(kind, workload):
        """Build the engine of `kind` on the synthetic LP (generated in HBM), warm up, flush, and time WINDOWS
        windows of whole update blocks."""
        m, n, seed = WORKLOADS[workload]
        nums_b = n * (1000 + (synthetic.splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64))
        nums_c = -(1000 + (synthetic.splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64))
        md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=nums_b / 4000.0, cost=nums_c / 1000.0,
                        upper_bound=np.full(n, np.inf))
        cfg = engine.default_config(device=local_rank, poll_interval=1 << 20, shard_rank=rank, shard_count=world,
                                    engine=engine.ENGINE_TABLEAU if kind == "tableau" else engine.ENGINE_REVISED,
                                    update_block=args.update_block)
        col_lo, col_hi = engine.shard_plan(md, cfg)
        n_local = col_hi - col_lo
        A = torch.empty((max(n_local, 1), m), dtype=torch.float64, device=dev)      # column-major m x n_local
        st = lib.relp_synth_fill_dense(A.data_ptr(), m, m, n_local, seed, col_lo, torch.cuda.current_stream().cuda_stream)
        assert st == 0, "synthetic fill failed"
        t = engine.Tableau(md, config=cfg, device_dense_ptr=A.data_ptr(), device_dense_ld=m)
        torch.cuda.synchronize()
        block = t.update_block()
        steps = window_steps(K, block)
        if not sharded:
            run, flush = t.run, t.flush
            done, oc = t.run(1)                        # phase 1 is empty (slack basis): one PRICE proves it
            assert oc == engine.PHASE_ONE_DONE, engine.OUTCOME_NAMES.get(oc)
        else:
            from rust_lp_amd.sharded import NativeShardedLoop, ShardedPivotLoop
            loop = None
            if not rehearse and args.shard_loop == "native":
                # the loop inside the library, RCCL called from C++ on the engine's stream
                try:
                    loop = NativeShardedLoop(t, dist, dev)
                    loop_kind[kind] = "native (relp_shard_run, RCCL from C++)"
                except engine.RelpError as e:             # agreed on by all ranks (all-reduce MIN inside)
                    print(f"[bench] native sharded loop unavailable, using the Python loop: {e}", file=sys.stderr)
            if loop is None:
                loop = ShardedPivotLoop(t, dist, dev, poll_interval=1 << 20)
                loop_kind[kind] = "python (torch.distributed collectives)"
            run, flush = loop.run, loop.flush
            assert loop.finish_phase_one() == engine.PHASE_ONE_DONE
        if W > 0:
            done, oc = run(W)
            assert done == W and oc == engine.RUNNING, "LP ended inside the warm-up"
        flush()                                        # the first window starts with an empty update block
        if events:
            stride = coprime_stride(args.event_stride, block)
            per_pivot = 12 * (WINDOWS * steps // stride + 1)
            t.profile_enable(True, per_pivot + 4 * WINDOWS * (steps // max(block, 1) + 1) + 64, stride)
        windows = []
        for _ in range(WINDOWS):
            if sharded:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done, oc = run(steps)
            torch.cuda.synchronize()
            if sharded:
                dist.barrier()
            dt = time.perf_counter() - t0
            if sharded:
                tt = torch.tensor([dt], dtype=torch.float64, device=dev if not rehearse else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            assert done == steps and oc == engine.RUNNING, f"only {done} of {steps} pivots were possible ({engine.OUTCOME_NAMES.get(oc)})"
            windows.append(dt)
        prof = t.profile_read() if events else {}
        obj = t.objective_function_value()
        t.close()
        del A
        torch.cuda.empty_cache()
        res = {"steps": steps, "window_s": windows, "prof": prof, "block": block, "objective": obj,
               "event_stride": stride if events else None}
        return section(res, workload, m, n, world, kind)

    primary = measure(primary_kind, args.workload)
    solo = args.engine == "default" and not sharded and rank == 0
    secondary = measure("revised", args.workload) if solo else None
    # BASELINE.json configs[1] (dense 2,000 x 2,000, dense-tableau path) beside the 10k target, same engine
    c2 = None
    if solo and args.workload != "c2" and not args.no_c2:
        m2, n2, _ = WORKLOADS["c2"]
        c2 = dict(measure("tableau", "c2"), workload=f"c2: synthetic dense LP {m2}x{n2} f64, dense tableau")
    # BASELINE.json configs[3] (10,000 x 50,000): at every N, the stored columns sharded like the primary workload
    c4 = None
    if args.engine == "default" and args.workload != "c4" and not args.no_c4:
        m4, n4, _ = WORKLOADS["c4"]
        c4 = dict(measure("tableau", "c4"), workload=f"c4: synthetic dense LP {m4}x{n4} f64, dense tableau, "
                                                      + ("single GPU" if world == 1 else f"stored columns sharded x{world}"))
    sparse = sparse_path(events, not args.no_cpu_baseline) if solo and not args.no_sparse else None
    if sparse is not None:
        if not args.quick:
            sparse["large"] = sparse_large()
        if not args.no_scale:
            sparse["scale"] = sparse_scale(not args.no_cpu_baseline)
        if not args.no_replicas:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--replicas-section"], stdout=subprocess.PIPE, timeout=900,
                                   env=dict(os.environ, GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "16")))
            lines = [ln for ln in child.stdout.decode().splitlines() if ln.strip()]
            sparse["replicas"] = json.loads(lines[-1]) if child.returncode == 0 and lines else {"error": f"child exit {child.returncode}"}
            if isinstance(sparse["replicas"], dict) and "replicas" in sparse["replicas"]:
                sparse["replicas"]["hardware_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "16")
    c1 = config_one(not args.no_cpu_baseline) if solo and not args.no_c1 else None
    c5 = config_five() if solo and not args.no_c5 else None

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            m, n, seed = WORKLOADS[args.workload]
            cpu = cpu_baseline(m, n, seed, W)
        out = assemble(args, world, primary_kind, primary, secondary, c2, c4, sparse, cpu, loop_kind.get(primary_kind))
        if c1 is not None:
            out["c1"] = c1
        if c5 is not None:
            out["c5"] = c5
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
