# scratch script of the GPU box runs (gpurun -- 'bash scripts/run_probe.sh'); as committed: smoke, then the whole GPU tier
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout -k 10 1150 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/gpu_tier.log 2>&1 || { tail -n 60 gpurun_out/gpu_tier.log; exit 1; }
tail -n 10 gpurun_out/gpu_tier.log
