set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_lu_layout2.py -m gpu -x -q > gpurun_out/layout2_tests.log 2>&1 || { tail -n 60 gpurun_out/layout2_tests.log; exit 1; }
tail -n 2 gpurun_out/layout2_tests.log
RELP_FT_BIG=2 RELP_LU_LOOKAHEAD=8 timeout -k 10 900 python -m pytest tests/test_gpu_lu_update.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/tier2_tests.log 2>&1 || { tail -n 40 gpurun_out/tier2_tests.log; exit 1; }
tail -n 2 gpurun_out/tier2_tests.log
