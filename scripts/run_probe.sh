set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_lu_layout2.py -m gpu -x -q > gpurun_out/layout2_tests.log 2>&1 || { tail -n 60 gpurun_out/layout2_tests.log; exit 1; }
tail -n 2 gpurun_out/layout2_tests.log
for spec in "mc:4000,16000,12 0" "mc:6000,24000,16 0"; do
  set -- $spec
  RELP_DEBUG=1 timeout -k 10 200 python scripts/xl_probe.py $1 $2 lu 20000 > gpurun_out/hs_probe.log 2>&1 || true
  echo "$spec"; grep "20000 pivots\|clocks/pivot" gpurun_out/hs_probe.log | cut -c1-400
done
timeout -k 10 200 python scripts/lu_large.py netlib/DFL001.SIF 1 lu 30000 -1 1 1 > gpurun_out/dfl.log 2>&1 || true
tail -n 3 gpurun_out/dfl.log | cut -c1-300
