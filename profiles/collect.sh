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
CMD="python3 $REPO/bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --no-kernel-events --no-sparse --no-c2 --no-c4 --no-c1 --no-c5 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
echo "trace exit=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch exit=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
echo "pmc write exit=$?"
# (SKIP_EXTRA=1: only the three passes above -- the c2 / c4 collections that fill roofline.traffic for those workloads;
# summarize.py then takes the calibration factors from the main collection's cal_* directories passed as CAL_FROM)
if [ "${SKIP_EXTRA:-0}" = "1" ]; then find "$OUT" -name '*.csv' -size +8M -delete; exit 0; fi
# 3. calibration of FETCH_SIZE / WRITE_SIZE for the flush kernel's access pattern (8 bytes per lane along a column of a
#    column-major tile): scripts/microbench/tile_rmw.hip walks the same tiles with a known byte count
#    (MI355X_MICROARCH.md section HBM: "calibrate on a known byte count in your own access pattern")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$REPO/scripts/microbench/tile_rmw.hip" -o "$OUT/tile_rmw" > "$OUT/tile_build.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/cal_fetch" -- "$OUT/tile_rmw" > "$OUT/cal_fetch.log" 2>&1
echo "cal fetch exit=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d "$OUT/cal_write" -- "$OUT/tile_rmw" > "$OUT/cal_write.log" 2>&1
echo "cal write exit=$?"
# 4. the sparse path: one whole 25FV47 solve on the LU engine (persistent pivot kernel): launches per pivot
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/lu_trace" -- python3 "$REPO/scripts/lu_profile.py" > "$OUT/lu_trace.log" 2>&1
echo "lu trace exit=$?"
rm -f "$OUT/tile_rmw"
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*.csv' | head -30
