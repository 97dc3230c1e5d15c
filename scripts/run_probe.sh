set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err || { tail -n 30 gpurun_out/r03_bench_final.err; exit 1; }
python scripts/show_scale.py gpurun_out/r03_bench_final.json
