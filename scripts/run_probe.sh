set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_lu_layout2.py -m gpu -x -q > gpurun_out/layout2_tests.log 2>&1 || { tail -n 60 gpurun_out/layout2_tests.log; exit 1; }
tail -n 2 gpurun_out/layout2_tests.log
for big in 1 2; do
  RELP_FT_BIG=$big timeout -k 10 200 python scripts/lu_large.py netlib/DFL001.SIF 1 lu 30000 -1 1 1 > gpurun_out/dfl_big$big.log 2>&1 || true
  tail -n 3 gpurun_out/dfl_big$big.log
done
RELP_DEBUG=1 timeout -k 10 200 python scripts/xl_probe.py mc:4000,16000,12 0 lu 20000 > gpurun_out/mc64k_lu3.log 2>&1 || true
grep -v "schedule\|so far" gpurun_out/mc64k_lu3.log | tail -n 7
