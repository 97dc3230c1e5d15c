cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 1100 python scripts/r4_guard_sweep.py > gpurun_out/r4/guard_sweep.log 2>&1; cat gpurun_out/r4/guard_sweep.log | cut -c1-200
