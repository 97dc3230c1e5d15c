"""Differential fuzz of the GPU engines against the f64 CPU oracle on many small random LPs (dense
<=-only and mixed sparse), random engine / block length.  Prints every mismatch; exit code 1 if any.

One class of inputs has no defined answer outside the explicit-inverse back end: when phase 1 ends with a
basic artificial that cannot be pivoted out, the reference marks its INDEX as a redundant row
(phase_one.rs:252); for a >= row the index differs from the row, a non-redundant row is deleted from B^-1
by index surgery (carry/mod.rs:650-689), and what follows is no longer the LP (HiGHS disagrees with the
reference's result on such cases).  The revised engine reproduces that literally and is checked against the
oracle; the tableau engine and the LU engine (like the reference's own LU back end, which re-inverts from the
filtered columns, carry/mod.rs:512-547) continue from a different state, so those cases are only counted.
Usage: python tests/tools/fuzz_gpu.py [N_CASES] [SEED0] [SIZE_SCALE]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic
from oracle import relp_f64

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # multiplies the problem sizes (multi-workgroup kernels)
rng = np.random.default_rng(seed0)
KINDS = [(engine.ENGINE_REVISED, (0, 1, 3, 7, 64)), (engine.ENGINE_TABLEAU, (1, 2, 5, 64)), (engine.ENGINE_LU, (1, 2, 6, 24, 64))]     # 24: the smallest interval with the look-ahead refactorisation
bad = 0
undefined = 0
stats = {"optimal": 0, "unbounded": 0, "infeasible": 0, "other": 0}
for case in range(N):
    seed = seed0 + case
    if rng.random() < 0.5:
        m, n = SCALE * int(rng.integers(2, 70)), SCALE * int(rng.integers(2, 90))
        lp = synthetic.dense_lp(m, n, seed)
        md = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])
        what = f"dense {m}x{n}"
    elif rng.random() < 0.5:
        m, n = SCALE * int(rng.integers(4, 80)), SCALE * int(rng.integers(4, 120))
        md = MatrixData.from_sparse_dict(synthetic.sparse_lp(m, n, seed, nnz_per_col=int(rng.integers(2, 7)),
                                                            frac_eq=float(rng.uniform(0, 0.5)), frac_ge=float(rng.uniform(0, 0.4)),
                                                            frac_bounded=float(rng.uniform(0, 0.6))))
        what = f"sparse {m}x{n}"
    else:
        # every row kind incl. ranges, negative costs (unbounded outcomes), contradictory rows (infeasible)
        m, n = SCALE * int(rng.integers(6, 70)), SCALE * int(rng.integers(4, 100))
        md = MatrixData.from_sparse_dict(synthetic.mixed_lp(
            m, n, seed, nnz_per_col=int(rng.integers(2, 6)), frac_eq=float(rng.uniform(0, 0.3)),
            frac_range=float(rng.uniform(0, 0.3)), frac_ge=float(rng.uniform(0, 0.3)), frac_bounded=float(rng.uniform(0, 0.6)),
            frac_negative_cost=float(rng.choice([0.0, 0.0, 0.1, 0.3])), infeasible=bool(rng.random() < 0.15)))
        what = f"mixed {m}x{n}"
    ref = relp_f64.OracleF64(md.ensure_csc() if md.col_ptr is None else md)
    status = ref.run(200000)
    stats[status if status in stats else "other"] += 1
    kind, blocks = KINDS[int(rng.integers(0, 3))]
    block = int(blocks[int(rng.integers(0, len(blocks)))])
    clean_rows = md.nr_eq + md.nr_range                      # artificial index == row only for these
    if kind != engine.ENGINE_REVISED and any(r >= clean_rows for r in ref.filtered_rows()):
        undefined += 1
        continue
    try:
        t = engine.Tableau(md, engine=kind, update_block=block, trace_capacity=1 << 16)
        oc = engine.OUTCOME_NAMES[t.solve_relaxation(max_iters=20 * len(ref.trace) + 1000)]     # bounded: no endless cycling
        tr = t.trace()
        first = next((i for i, (x, y) in enumerate(zip(tr, ref.trace)) if x != y), None)
        detail = f"first trace difference at {first} of {len(tr)} gpu pivots"
        if oc == status == "optimal":
            detail += f", objective gpu {t.objective_function_value():.12g} oracle {ref.objective:.12g}"
        ok = oc == status and tr == ref.trace
        if ok and status == "optimal":
            ok = abs(t.objective_function_value() - ref.objective) <= 1e-9 * max(1.0, abs(ref.objective))
            ok = ok and np.max(np.abs(t.b() - ref.b())) <= 1e-7 * max(1.0, np.max(np.abs(ref.b())))
        t.close()
    except Exception as e:          # noqa: BLE001
        ok, oc, detail = False, f"exception {e}", ""
        if kind == engine.ENGINE_LU and "singular" in str(e) and ref.filtered_rows():
            # the stuck artificial had re-entered in a foreign basis position: the index surgery removes the
            # wrong position; the reference's LU back end would fail in `invert` on the same matrix
            undefined += 1
            continue
    if case % 25 == 24:
        print(f"... {case + 1} cases, {bad} mismatches so far", flush=True)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case} seed {seed} {what} kind {kind} block {block}: gpu {oc} vs oracle {status} "
              f"({len(ref.trace)} pivots) {detail}", flush=True)
print(f"{N} cases, {bad} mismatches, {undefined} skipped (wrong-row removal, undefined outside the revised engine), "
      f"oracle outcomes {stats}")
sys.exit(1 if bad else 0)
