set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RELP_DEBUG=1 timeout -k 10 200 python scripts/lu_large.py netlib/25FV47.SIF 0 lu 30000 > gpurun_out/fv_diag.log 2>&1 || true
grep "pass_diag" gpurun_out/fv_diag.log | tail -n 2
grep "optimal\|clocks/pivot\|passes walked" gpurun_out/fv_diag.log | tail -n 6 | cut -c1-330
