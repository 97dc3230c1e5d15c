set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_gpu.py > gpurun_out/gpu_tier.log 2>&1 || { tail -n 60 gpurun_out/gpu_tier.log; exit 1; }
tail -n 3 gpurun_out/gpu_tier.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_scale_le -o le -- python3 scripts/xl_probe.py le:30000,90000 0 lu 6000 > gpurun_out/prof_scale_le.log 2>&1
grep "pivots " gpurun_out/prof_scale_le.log | tail -n 2
find gpurun_out/prof_scale_le -name "*kernel_trace.csv" -delete
head -4 gpurun_out/prof_scale_le/le_kernel_stats.csv | cut -c1-120
