set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RELP_FT_BIG=2 timeout -k 10 900 python -m pytest tests/test_gpu_lu_update.py tests/test_gpu_parity.py tests/test_gpu_big_pins.py -m gpu -x -q > gpurun_out/tier2_tests.log 2>&1 || { tail -n 40 gpurun_out/tier2_tests.log; exit 1; }
tail -n 5 gpurun_out/tier2_tests.log
for sz in "12000 36000 6000" "30000 90000 4000"; do
  set -- $sz
  RELP_DEBUG=1 timeout -k 10 300 python scripts/xl_probe.py $1 $2 lu $3 > gpurun_out/xl2_$1_lu.log 2>&1 || { tail -n 20 gpurun_out/xl2_$1_lu.log; exit 1; }
  grep -v "schedule" gpurun_out/xl2_$1_lu.log | tail -n 8
done
