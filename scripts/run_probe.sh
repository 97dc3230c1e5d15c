set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_lu_layout2.py -m gpu -x -q > gpurun_out/layout2_tests.log 2>&1 || { tail -n 60 gpurun_out/layout2_tests.log; exit 1; }
tail -n 2 gpurun_out/layout2_tests.log
for spec in "mc:4000,16000,12 0" "mc:6000,24000,16 0" "30000 90000" "le:30000,90000 0"; do
  set -- $spec
  RELP_DEBUG=1 timeout -k 10 200 python scripts/xl_probe.py $1 $2 lu 20000 > gpurun_out/hs_probe.log 2>&1 || true
  echo "$spec"; grep "20000 pivots\|clocks/pivot" gpurun_out/hs_probe.log | cut -c1-330
done
RELP_FT_BIG=2 RELP_LU_LOOKAHEAD=8 timeout -k 10 900 python -m pytest tests/test_gpu_lu_update.py tests/test_gpu_parity.py tests/test_gpu_big_pins.py -m gpu -x -q > gpurun_out/tier2_tests.log 2>&1 || { tail -n 40 gpurun_out/tier2_tests.log; exit 1; }
tail -n 2 gpurun_out/tier2_tests.log
