set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "fuzz, layout 2 forced, grid price forced"
RELP_FT_BIG=2 RELP_FT_GRID_PRICE=1 timeout -k 10 500 python tests/tools/fuzz_gpu.py 400 7000 > gpurun_out/fuzz_l2.log 2>&1 || { tail -n 30 gpurun_out/fuzz_l2.log; exit 1; }
tail -n 4 gpurun_out/fuzz_l2.log
echo "fuzz, layout 2 forced, larger cases"
RELP_FT_BIG=2 timeout -k 10 500 python tests/tools/fuzz_gpu.py 120 9000 4 > gpurun_out/fuzz_l2b.log 2>&1 || { tail -n 30 gpurun_out/fuzz_l2b.log; exit 1; }
tail -n 4 gpurun_out/fuzz_l2b.log
echo "fuzz, layout 1 forced"
RELP_FT_BIG=1 timeout -k 10 500 python tests/tools/fuzz_gpu.py 300 11000 2 > gpurun_out/fuzz_l1.log 2>&1 || { tail -n 30 gpurun_out/fuzz_l1.log; exit 1; }
tail -n 4 gpurun_out/fuzz_l1.log
