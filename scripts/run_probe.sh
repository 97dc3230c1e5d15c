set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python scripts/lu_large.py netlib/DFL001.SIF 1 lu 30000 -1 1 1 > gpurun_out/dfl.log 2>&1 || true
echo "DFL001"; grep "30000 pivots\|clocks/pivot" gpurun_out/dfl.log | cut -c1-330
timeout -k 10 200 python scripts/lu_large.py netlib/80BAU3B.SIF 1 lu 30000 -1 0 1 > gpurun_out/bau.log 2>&1 || true
echo "80BAU3B"; grep "optimal\|clocks/pivot" gpurun_out/bau.log | tail -n 2 | cut -c1-330
timeout -k 10 200 python scripts/lu_large.py miplib/acc-tight4.mps 0 lu 60000 > gpurun_out/acc.log 2>&1 || true
echo "acc-tight4"; grep "60000 pivots\|clocks/pivot" gpurun_out/acc.log | tail -n 2 | cut -c1-330
RELP_FT_BIG=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lu_update.py -m gpu -x -q > gpurun_out/t2.log 2>&1 || { tail -n 40 gpurun_out/t2.log; exit 1; }
tail -n 2 gpurun_out/t2.log
