cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 500 python -m pytest tests/test_gpu_lu_device.py tests/test_gpu_lu_vs_lu_oracle.py tests/test_gpu_lu_update.py tests/test_gpu_lu_layout2.py -x -q -m gpu > gpurun_out/r4/lu_device.log 2>&1; tail -n 3 gpurun_out/r4/lu_device.log
cd /tmp && export TMPDIR=/tmp
RELP_DEBUG=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/prof_luf3 -o luf -- python $GRAFT_REPO_ROOT/scripts/r4_luf_profile.py 11 1 > $GRAFT_REPO_ROOT/gpurun_out/r4/prof_luf3.log 2>&1
cd $GRAFT_REPO_ROOT
grep -E "device factorisation,|schedule |block" gpurun_out/r4/prof_luf3.log | tail -n 6 | cut -c1-700
find gpurun_out/r4/prof_luf3 -name "*kernel_stats.csv" | head -n 1 | xargs -r head -n 8
