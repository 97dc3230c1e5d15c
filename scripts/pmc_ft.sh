#!/bin/bash
# SQ / instruction-cache counters of the LU engine's persistent pivot kernel on Netlib 25FV47 (separate --pmc passes, no
# trace flags).  Usage (GPU box, repo root): bash scripts/pmc_ft.sh; results under gpurun_out/pmc_ft/.
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_ft
mkdir -p $OUT
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SALU"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 $ROOT/scripts/lu_profile.py > $OUT/$tag.log 2>&1 || echo "failed: $set"
done
python3 - <<PY
import csv, glob, collections
tot = collections.Counter(); n = 0
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_ft_run" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k:28s} {v:18.0f}")
PY
