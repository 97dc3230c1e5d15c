"""Sensitivity of the tableau engine's per-pivot time on dense10k to the block length and the tie band
(diagnostic: is a pivot bound by the bytes it moves or by dependent latencies?)."""
import ctypes as C
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import rust_lp_amd  # noqa: F401
from rust_lp_amd import MatrixData, engine, synthetic

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seed = 20250002
lib = engine.load_library()
b = n * (1000 + (synthetic.splitmix64(seed, 1, np.arange(m, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 4000.0
c = -(1000 + (synthetic.splitmix64(seed, 2, np.arange(n, dtype=np.uint64)) % np.uint64(1000)).astype(np.int64)) / 1000.0
ptr = C.c_void_p()
assert lib.relp_device_alloc(C.byref(ptr), m * n * 8) == 0
assert lib.relp_synth_fill_dense(ptr, m, m, n, seed, 0, None) == 0
md = MatrixData(nr_normal=n, nr_eq=0, nr_range=0, nr_le=m, nr_ge=0, b=b, cost=c, upper_bound=np.full(n, np.inf))
for label, kw in (("K=64", {}), ("K=16", dict(update_block=16)), ("K=8", dict(update_block=8)), ("K=32", dict(update_block=32)),
                  ("K=48", dict(update_block=48)), ("K=80", dict(update_block=80)), ("K=96", dict(update_block=96)),
                  ("K=128", dict(update_block=128)), ("K=64 tol_tie=0", dict(tol_tie=0.0))):
    t = engine.Tableau(md, engine=engine.ENGINE_TABLEAU, device_dense_ptr=ptr.value, device_dense_ld=m, poll_interval=1024, **kw)
    t.run(1)
    t.run(64)
    K = 1536
    t0 = time.perf_counter()
    done, oc = t.run(K)
    dt = time.perf_counter() - t0
    print(f"{label:16s} {done} pivots  {dt / done * 1e6:7.2f} us/pivot  {done / dt:8.0f} it/s", flush=True)
    t.close()
