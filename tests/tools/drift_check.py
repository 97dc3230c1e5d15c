import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
import rust_lp_amd
from rust_lp_amd import MatrixData, engine, synthetic
from oracle import relp_f64
m=n=int(sys.argv[1]); seed=20250001
lp = synthetic.dense_lp(m,n,seed)
md = MatrixData.from_dense_le(lp["A"], lp["b"], lp["c"])
for kind in (engine.ENGINE_TABLEAU, engine.ENGINE_REVISED):
    t = engine.Tableau(md, engine=kind, trace_capacity=1<<20)
    t0=time.time(); oc = t.solve_relaxation(); dt=time.time()-t0
    it = t.iterations()
    basis = t.basis_indices(); b=t.b(); d=t.relative_costs()
    # residual of B x_B = rhs and dual feasibility
    a = np.hstack([md.dense, np.eye(m)])
    B = a[:, basis]
    xB = np.linalg.solve(B, lp["b"])
    print("engine",kind,"outcome",oc,"iters",it,"time",round(dt,2),"obj",t.objective_function_value(),
          "max|b - B^-1 rhs|", np.max(np.abs(xB-b)), "min b", b.min(), "min d nonbasic", d[~np.isin(np.arange(len(d)), basis)].min(), "max|d basic|", np.max(np.abs(d[basis])))
    c = np.concatenate([md.cost, np.zeros(m)])
    print("   true obj from basis", c[basis]@xB)
    if kind==engine.ENGINE_TABLEAU: tr_tab = t.trace()
    else: print("   traces equal:", tr_tab == t.trace(), len(tr_tab))
