set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for spec in "mc:4000,16000,12 0" "mc:6000,24000,16 0"; do
  set -- $spec
  RELP_DEBUG=1 timeout -k 10 200 python scripts/xl_probe.py $1 $2 lu 20000 > gpurun_out/hs_probe.log 2>&1 || true
  echo "$spec"; grep "20000 pivots\|so far" gpurun_out/hs_probe.log | tail -n 2 | cut -c1-330
done
timeout -k 10 200 python scripts/lu_large.py netlib/25FV47.SIF 0 lu 30000 > gpurun_out/fv.log 2>&1 || true
echo "25FV47"; grep "optimal" gpurun_out/fv.log | tail -n 1 | cut -c1-200
timeout -k 10 1000 python -m pytest tests/test_gpu_lu_layout2.py tests/test_gpu_lu_update.py tests/test_gpu_parity.py tests/test_gpu_big_pins.py tests/test_gpu_lu_device.py -m gpu -x -q > gpurun_out/t1.log 2>&1 || { tail -n 40 gpurun_out/t1.log; exit 1; }
tail -n 2 gpurun_out/t1.log
