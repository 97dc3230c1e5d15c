# the committed GPU-box script: smoke, then the whole GPU tier (gpurun -- 'bash scripts/run_probe.sh')
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout -k 10 1150 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/gpu_tier.log 2>&1 || { tail -n 60 gpurun_out/gpu_tier.log; exit 1; }
tail -n 14 gpurun_out/gpu_tier.log
