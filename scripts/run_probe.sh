set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_lu_device.py -q -s > gpurun_out/lu_device_tests.log 2>&1 || true
grep -E "device factorisations|bump |passed|failed|Error" gpurun_out/lu_device_tests.log | tail -n 40
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_lu_device.py 2>&1 | tail -n 25
