set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_sq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 140 --warmup 20 --no-cpu-baseline --no-kernel-events --no-sparse --engine tableau"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -f csv -d "$OUT/a" -- $CMD > "$OUT/a.log" 2>&1
echo "a exit=$?"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace -f csv -d "$OUT/b" -- $CMD > "$OUT/b.log" 2>&1
echo "b exit=$?"
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_sq"
for tag in ("a", "b"):
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("relp::", "")
            if "flush_lds" not in name: continue
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, cs in agg.items():
            for c, vals in sorted(cs.items()):
                print(tag, name, c, len(vals), max(vals))
PY
find "$OUT" -name "*.csv" -size +2M -delete
