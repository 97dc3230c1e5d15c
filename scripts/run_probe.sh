# scratch script of the GPU box runs (gpurun -- 'bash scripts/run_probe.sh'); as committed: the bench line of record, then the GPU tier
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err || { tail -n 30 gpurun_out/r03_bench_final.err; exit 1; }
python scripts/show_scale.py gpurun_out/r03_bench_final.json
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_gpu.py::test_driver_command_emits_one_complete_json_line > gpurun_out/gpu_tier.log 2>&1 || { tail -n 60 gpurun_out/gpu_tier.log; exit 1; }
tail -n 3 gpurun_out/gpu_tier.log
