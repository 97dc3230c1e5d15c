"""Debug driver: one small solve per engine, printing progress unbuffered."""
import sys
import numpy as np
sys.path.insert(0, ".")
import rust_lp_amd  # noqa
from rust_lp_amd import MatrixData, engine, synthetic

def p(*a):
    print(*a, flush=True)

lp = synthetic.dense_lp(32, 48, 7)
md = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])
for kind, block in ((engine.ENGINE_REVISED, 0), (engine.ENGINE_REVISED, 3), (engine.ENGINE_TABLEAU, 3), (engine.ENGINE_LU, 3)):
    p("create", kind, block)
    t = engine.Tableau(md, trace_capacity=4096, update_block=block, engine=kind)
    p("created; run phase 1")
    p(t.run(1 << 20))
    p("run 1"); p(t.run(1))
    p("run 5"); p(t.run(5))
    p("solve"); p(t.solve_relaxation(), t.objective_function_value(), len(t.trace()))
