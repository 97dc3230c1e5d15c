cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 1100 python scripts/corpus_sweep.py --verified --seconds 20 --out gpurun_out/r4/corpus_verified.json > gpurun_out/r4/corpus_verified.log 2>&1
tail -n 12 gpurun_out/r4/corpus_verified.log
