#!/usr/bin/env python3
"""bench.py -- simplex iterations/s of the MI355X pivot engine on the synthetic dense LP.

A "step" is one pass of the hot path (PRICE -> FTRAN -> RATIO -> UPDATE = one basis change) of
the simplex on the synthetic dense LP of BASELINE.json (rust-lp_amd/synthetic.py), inputs
resident in HBM before the timed region.  One JSON line on stdout (rank 0).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

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
  numbers under "revised_engine" (its own K timed pivots).
N > 1: the same LP (strong scaling); the stored tableau columns are split contiguously over the
ranks, ONE all-gather of [key, j, d_j, alpha(m)] candidates per pivot over RCCL, everything else local.
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {"dense10k": (10000, 10000, 20250002), "c2": (2000, 2000, 20250001), "c4": (10000, 50000, 20250003)}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F64_MFMA_PEAK_TF = 78.6    # FP64 matrix 78.6 TFLOP/s


def cpu_baseline(m, n, seed, warmup, steps, budget_s=20.0):
    """The C (f64) restatement of the reference path (oracle/relp_f64.c, kind "port"), one core,
    timed on the same LP over the same iteration window [warmup, warmup + k) until ~budget_s."""
    from rust_lp_amd import MatrixData, synthetic
    from oracle import relp_f64
    lp = synthetic.dense_lp(m, n, seed)
    md = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"]).ensure_csc()
    md.dense = None
    del lp
    ref = relp_f64.OracleF64(md)
    ref.run(max_iters=1 << 40, through_phases=False, record=False)      # empty phase 1 -> phase 2
    ref.run(max_iters=warmup, record=False)
    done, t0 = 0, time.perf_counter()
    chunk = max(1, min(5, steps))
    while done < steps and time.perf_counter() - t0 < budget_s:
        ref.run(max_iters=min(chunk, steps - done), record=False)
        done += ref.last_n_done
        if ref.last_n_done == 0:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt if dt > 0 else None, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{done} pivots (iterations {warmup}..{warmup + done}) of the same {m}x{n} LP, "
                      f"oracle/relp_f64.c (f64 restatement of Carry<_, BasisInverseRows>), {dt:.1f} s",
            "nproc": os.cpu_count(), "cpu_model": cpu_model(),
            "reference_toolchain": "cargo " + ("present (not used: the crate needs a 2021 nightly and crates.io)"
                                               if shutil.which("cargo") else "absent on this host")}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def sparse_path(events, with_cpu):
    """The sparse path (BASELINE.json configs[2]): Netlib 25FV47 through the build's MPS reader, presolve and
    standardisation, solved to optimality by the LU engine (host Markowitz refactorisation every 128 pivots,
    level-scheduled FTRAN / BTRAN in LDS, CSC PRICE).  The whole solve is timed; it is latency-bound (the
    factors are a few hundred KB), so the HBM figure is reported for what it is."""
    from rust_lp_amd import engine, general_form, mps
    path = os.path.join(ROOT, "tests", "golden", "mps", "netlib", "25FV47.SIF")
    if not os.path.exists(path):
        return None
    gf = general_form.GeneralForm.from_mps(mps.import_file(path, True))
    md = gf.to_matrix_data(gf.derive_matrix_data_exact())
    tol = dict(tol_pivot=1e-5, tol_cost=1e-7)            # see tests/test_gpu_parity.py, config C3
    t = engine.Tableau(md, engine=engine.ENGINE_LU, **tol)
    if events:
        t.profile_enable(True, 40000, 8)
    t0 = time.perf_counter()
    outcome = t.solve_relaxation()
    dt = time.perf_counter() - t0
    its = t.iterations()
    degenerate = t.degenerate_pivots()
    stats = t.lu_stats()
    prof = t.profile_read() if events else {}
    obj = t.objective_function_value() + float(gf.fixed_cost)
    t.close()
    out = {"workload": f"Netlib 25FV47 after presolve: {stats['m']} rows, {md.nr_normal} structural columns, "
                       f"{len(md.values)} nonzeros; FirstProfitableWithMemory / SteepestDescent, whole two-phase solve",
           "engine": "lu", "outcome": engine.OUTCOME_NAMES.get(outcome), "objective": obj, "reference_objective": 5.5018459e+03,
           "pivots": its, "degenerate_pivots": degenerate, "value": its / dt, "unit": "iterations/s", "seconds": dt,
           "tolerances": tol,
           "refactorisations": stats["refactorisations"],
           "last_factor": {k: stats[k] for k in ("nnz_l", "nnz_u", "levels_l", "levels_u")}}
    if prof:
        kt = {name: {"launches": cnt, "avg_us": round(ms * 1e3 / cnt, 3)} for name, (cnt, ms) in prof.items() if cnt}
        out["kernels"] = kt
        if "ftran" in kt:
            factor_bytes = 12.0 * (stats["nnz_l"] + stats["nnz_u"]) + 40.0 * stats["m"]
            out["ftran_GBps"] = round(factor_bytes / (kt["ftran"]["avg_us"] * 1e-6) / 1e9, 3)
            out["ftran_note"] = "12 (nnz L + nnz U) + 40 m bytes of the last factor / average FTRAN time: dependency-bound"
    # the same LP on the explicit-inverse engine (m = 790: B^-1 is 5 MB, the dense kernels are at their latency floor)
    # (re-inverted every 1,000 pivots, the engine's default at this size)
    t = engine.Tableau(md, engine=engine.ENGINE_REVISED, update_block=0, **tol)
    t0 = time.perf_counter()
    outcome2 = t.solve_relaxation()
    dt2 = time.perf_counter() - t0
    out["explicit_inverse_engine"] = {"outcome": engine.OUTCOME_NAMES.get(outcome2), "pivots": t.iterations(),
                                      "value": t.iterations() / dt2, "unit": "iterations/s", "seconds": dt2,
                                      "reinversions": t.reinversions(),
                                      "objective": t.objective_function_value() + float(gf.fixed_cost)}
    t.close()
    # ... and on the dense tableau engine (T0 = 790 x 2,300 doubles; re-tabulated every 1,000 pivots)
    t = engine.Tableau(md, engine=engine.ENGINE_TABLEAU, update_block=32, **tol)
    t0 = time.perf_counter()
    outcome3 = t.solve_relaxation()
    dt3 = time.perf_counter() - t0
    out["tableau_engine"] = {"outcome": engine.OUTCOME_NAMES.get(outcome3), "pivots": t.iterations(),
                             "value": t.iterations() / dt3, "unit": "iterations/s", "seconds": dt3,
                             "retabulations": t.reinversions(),
                             "objective": t.objective_function_value() + float(gf.fixed_cost)}
    t.close()
    if with_cpu:
        from oracle import relp_f64
        ref = relp_f64.OracleF64(md, **tol)
        c0 = time.perf_counter()
        ref.run(max_iters=3000, record=False)
        cdt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": 3000 / cdt, "unit": "iterations/s", "cores": 1, "kind": "port",
                               "sample": f"first 3000 pivots of the same LP, oracle/relp_f64.c, {cdt:.1f} s"}
    return out


def kernel_table(prof, alg_bytes, alg_flops):
    kernels = {}
    for name, (cnt, ms) in prof.items():
        if cnt > 0:
            avg_us = ms * 1e3 / cnt
            entry = {"launches": cnt, "avg_us": round(avg_us, 3)}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="dense10k", choices=sorted(WORKLOADS))
    ap.add_argument("--engine", default="default", choices=["default", "revised", "tableau"],
                    help="default = tableau as the measured engine (+ the revised engine's numbers at N = 1)")
    ap.add_argument("--update-block", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sparse", action="store_true", help="skip the sparse-path (LU engine, Netlib 25FV47) section")
    ap.add_argument("--no-c2", action="store_true", help="skip the configs[1] (dense 2,000 x 2,000) section")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--event-stride", type=int, default=16,
                    help="bracket the kernels of every n-th pivot with HIP events (1 = every pivot)")
    ap.add_argument("--shard-loop", default="native", choices=["native", "python"],
                    help="N > 1: pivot loop inside the library calling RCCL itself, or the Python loop over torch.distributed")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU loop (torch.distributed collectives) even at N = 1 (rehearsal)")
    args = ap.parse_args()

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
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one rank per GPU)")
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

    m, n, seed = WORKLOADS[args.workload]
    K, W = args.steps, args.warmup
    lib = engine.load_library()
    events = not args.no_kernel_events
    primary = "tableau" if args.engine in ("default", "tableau") else "revised"

    loop_kind = {}

    def measure(kind, m=m, n=n, seed=seed):
        """Build the engine of `kind` on the synthetic LP (generated in HBM) and time K pivots."""
        nums_b = n * (1000 + (synthetic.splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64))
        nums_c = -(1000 + (synthetic.splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64))
        md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=nums_b / 4000.0, cost=nums_c / 1000.0,
                        upper_bound=np.full(n, np.inf))
        cfg = engine.default_config(device=local_rank, poll_interval=max(K, W, 1), shard_rank=rank, shard_count=world,
                                    engine=engine.ENGINE_TABLEAU if kind == "tableau" else engine.ENGINE_REVISED,
                                    update_block=args.update_block)
        col_lo, col_hi = engine.shard_plan(md, cfg)
        n_local = col_hi - col_lo
        A = torch.empty((max(n_local, 1), m), dtype=torch.float64, device=dev)      # column-major m x n_local
        st = lib.relp_synth_fill_dense(A.data_ptr(), m, m, n_local, seed, col_lo, torch.cuda.current_stream().cuda_stream)
        assert st == 0, "synthetic fill failed"
        t = engine.Tableau(md, config=cfg, device_dense_ptr=A.data_ptr(), device_dense_ld=m)
        torch.cuda.synchronize()
        if not sharded:
            done, oc = t.run(1)                        # phase 1 is empty (slack basis): one PRICE proves it
            assert oc == engine.PHASE_ONE_DONE, engine.OUTCOME_NAMES.get(oc)
            done, oc = t.run(W)
            assert done == W and oc == engine.RUNNING, "LP ended inside the warm-up"
            if events:
                t.profile_enable(True, 12 * K + 16, args.event_stride)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done, oc = t.run(K)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
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
                loop = ShardedPivotLoop(t, dist, dev)
                loop_kind[kind] = "python (torch.distributed collectives)"
            oc = loop.finish_phase_one()
            assert oc == engine.PHASE_ONE_DONE
            done, oc = loop.run(W)
            assert done == W and oc == engine.RUNNING
            if events:
                t.profile_enable(True, 12 * K + 16, args.event_stride)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done, oc = loop.run(K)
            torch.cuda.synchronize()
            dist.barrier()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert done == K, f"only {done} of {K} pivots were possible"
        prof = t.profile_read() if events else {}
        block = t.update_block()
        obj = t.objective_function_value()
        t.close()
        del A
        torch.cuda.empty_cache()
        # algorithmic bytes / flops per launch, local shard
        if kind == "tableau":
            n_store = n + m
            n_owned = (n_store + world - 1) // world
            alg_bytes = {"flush": 16.0 * m * n_owned}
            alg_flops = {"flush": 2.0 * m * n_owned * block}
        else:
            rows_local = (m + world - 1) // world
            alg_bytes = {"price": 8.0 * m * n_local, "ftran": 8.0 * rows_local * m + 16.0 * m,
                         "update_inverse": 16.0 * rows_local * m, "flush": 16.0 * rows_local * m}
            alg_flops = {"flush": 2.0 * rows_local * m * max(block, 1)}
        return {"dt": dt, "kernels": kernel_table(prof, alg_bytes, alg_flops), "alg_bytes": alg_bytes, "block": block,
                "objective": obj}

    res = measure(primary)
    dt, kernels = res["dt"], res["kernels"]

    roofline = None
    if kernels:
        if primary == "tableau" and "flush" in kernels:
            # the flush T0 += W R0 is the one kernel that streams the m x (n + m) tableau; at K = 64 its HBM time
            # (16 m n bytes) exceeds its MFMA time (2 m n K flops at 78.6 TFLOP/s), so it is priced against HBM
            fl = kernels["flush"]
            roofline = {"kernel": "k_tab_flush_lds", "bound": "hbm", "achieved": fl["GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(fl["GBps"] / HBM_PEAK_GBS, 4),
                        "traffic": load_traffic(args.workload, "tab_flush_lds"),
                        "algorithmic_bytes_per_launch": res["alg_bytes"]["flush"],
                        "launches_per_pivot": 1.0 / max(res["block"], 1),
                        "mfma": {"achieved": fl.get("TFLOPs"), "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s (f64 matrix)",
                                 "frac": round(fl.get("TFLOPs", 0.0) / F64_MFMA_PEAK_TF, 4)}}
        else:
            streaming = [k for k in kernels if k in ("price", "ftran", "update_inverse")]
            dom = max(streaming, key=lambda k: kernels[k]["avg_us"] * kernels[k]["launches"])
            traffic_key = {"price": "price_all", "ftran": "ftran", "update_inverse": "update_inverse_vectors"}[dom]
            roofline = {"kernel": "k_" + traffic_key, "bound": "hbm", "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(kernels[dom]["GBps"] / HBM_PEAK_GBS, 4),
                        "traffic": load_traffic(args.workload, traffic_key),
                        "algorithmic_bytes_per_launch": res["alg_bytes"][dom]}

    secondary = None
    if args.engine == "default" and not sharded and rank == 0:
        r2 = measure("revised")
        k2 = r2["kernels"]
        secondary = {"value": K / r2["dt"], "unit": "iterations/s", "ms_per_step": r2["dt"] * 1e3 / K,
                     "update_block": r2["block"], "kernels": k2,
                     "note": "explicit dense inverse (Carry<_, BasisInverseRows<_>>): PRICE streams A, FTRAN streams B^-1 "
                             "every pivot; GB/s = algorithmic bytes / HIP-event duration",
                     "ftran_hbm_frac": round(k2["ftran"]["GBps"] / HBM_PEAK_GBS, 4) if "ftran" in k2 else None,
                     "price_hbm_frac": round(k2["price"]["GBps"] / HBM_PEAK_GBS, 4) if "price" in k2 else None,
                     "ftran_traffic": load_traffic(args.workload, "ftran"),
                     "objective_after_run": r2["objective"]}

    # BASELINE.json configs[1] (dense 2,000 x 2,000, dense-tableau path) beside the 10k target, same engine
    c2 = None
    if args.engine == "default" and not sharded and rank == 0 and args.workload != "c2" and not args.no_c2:
        m2, n2, seed2 = WORKLOADS["c2"]
        r3 = measure("tableau", m2, n2, seed2)
        c2 = {"workload": f"c2: synthetic dense LP {m2}x{n2} f64, dense tableau", "value": K / r3["dt"], "unit": "iterations/s",
              "ms_per_step": r3["dt"] * 1e3 / K, "update_block": r3["block"], "kernels": r3["kernels"],
              "objective_after_run": r3["objective"]}

    sparse = None
    if args.engine == "default" and not sharded and rank == 0 and not args.no_sparse:
        sparse = sparse_path(events, not args.no_cpu_baseline)

    if rank == 0:
        out = {
            "metric": "simplex iterations/sec", "value": K / dt, "unit": "iterations/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt * 1e3 / K, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic dense LP {m}x{n} f64 (+{m} slack columns), SteepestDescent, "
                                   + ("dense tableau with blocked f64-MFMA updates" if primary == "tableau"
                                      else "explicit dense basis inverse"),
                       "m": m, "n": n, "seed": seed, "engine": primary, "update_block": res["block"],
                       "parallelism": "single GPU" if world == 1 else
                       (f"stored tableau columns sharded x{world}, one all-gather per pivot" if primary == "tableau"
                        else f"columns of A and rows of B^-1 sharded x{world}")},
            "roofline": roofline, "kernels": kernels, "kernel_event_stride": args.event_stride if events else None,
            "objective_after_run": res["objective"],
            "reference_iteration_algorithmic_bytes": 8.0 * m * n + 24.0 * m * m,
        }
        if sharded:
            out["config"]["shard_loop"] = loop_kind.get(primary)
        if secondary is not None:
            out["revised_engine"] = secondary
        if c2 is not None:
            out["c2"] = c2
        if sparse is not None:
            out["sparse_engine"] = sparse
        if world == 1:
            out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(m, n, seed, W, K)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
