# scratch script of the GPU box runs (gpurun -- 'bash scripts/run_probe.sh')
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_lu_device.py -m gpu -x -q > gpurun_out/r4/lu_device.log 2>&1; echo "pytest exit $?"; tail -n 15 gpurun_out/r4/lu_device.log
for args in "24 1" "-1 1" "11 1" "11 0"; do timeout -k 10 120 python scripts/r4_luf_profile.py $args 2>&1 | tail -n 1; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/prof_luf2 -o luf -- python3 $GRAFT_REPO_ROOT/scripts/r4_luf_profile.py 11 1 > $GRAFT_REPO_ROOT/gpurun_out/r4/prof_luf2.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4/prof_luf2 -name "*kernel_stats.csv" | head -n 1 | xargs -r head -n 8
