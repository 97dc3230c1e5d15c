#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats   -> per-kernel average durations
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes)
# Summaries are written under gpurun_out/prof_<tag>/ and later copied into profiles/.
set -u
TAG=${1:-r01}
STEPS=${2:-100}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --no-kernel-events --no-sparse --no-c2 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
echo "trace exit=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch exit=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
echo "pmc write exit=$?"
find "$OUT" -name '*.csv' | head -20
