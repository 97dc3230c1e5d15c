set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { RELP_DEBUG=1 timeout -k 10 250 python scripts/xl_probe.py "$@" > gpurun_out/tbl.log 2>&1 || true; echo "== $*"; grep "pivots \|clocks/pivot\|non-zeros per" gpurun_out/tbl.log | tail -n 4 | cut -c1-330; }
run mc:4000,16000,12 0 lu 20000
run mc:6000,24000,16 0 lu 20000
run mc:2000,8000,10 0 lu 30000
run 30000 90000 lu 4000
run le:30000,90000 0 lu 6000
run 12000 36000 lu 6000
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_scale_mc -o mc -- python3 scripts/xl_probe.py mc:4000,16000,12 0 lu 20000 > gpurun_out/prof_scale_mc.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_scale_le -o le -- python3 scripts/xl_probe.py le:30000,90000 0 lu 6000 > gpurun_out/prof_scale_le.log 2>&1
find gpurun_out/prof_scale_mc gpurun_out/prof_scale_le -name "*kernel_trace.csv" -delete
python bench.py > gpurun_out/bench_r03c.json 2> gpurun_out/bench_r03c.err || { tail -n 30 gpurun_out/bench_r03c.err; exit 1; }
python scripts/show_scale.py gpurun_out/bench_r03c.json
