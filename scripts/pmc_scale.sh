#!/bin/bash
# SQ / instruction-cache counters of the persistent pivot kernel in layout 2 on the 63,988-row multi-commodity LP (separate --pmc
# passes, no trace flags).  Usage (GPU box, repo root): bash scripts/pmc_scale.sh; results under gpurun_out/pmc_scale/.
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_scale
mkdir -p $OUT
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 $ROOT/scripts/xl_probe.py mc:4000,16000,12 0 lu 6000 > $OUT/$tag.log 2>&1 || echo "failed: $set"
done
python3 - <<PY
import csv, glob, collections
tot = collections.Counter(); launches = collections.Counter()
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_ft_run" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); launches[row["Counter_Name"]] += 1
for k, v in sorted(tot.items()):
    print(f"{k:28s} {v:18.0f}   ({launches[k]} launches)")
PY
