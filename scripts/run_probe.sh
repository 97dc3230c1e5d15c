set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SECONDS=0; python bench.py > gpurun_out/bench_scale.json 2> gpurun_out/bench_scale.err || { tail -n 30 gpurun_out/bench_scale.err; exit 1; }
echo "bench seconds $SECONDS"
python scripts/show_scale.py gpurun_out/bench_scale.json
