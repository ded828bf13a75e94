#!/bin/bash
# SQ / SQC counter passes over k_fast_keypoints alone (8 batches of 2048 frames through tools/orb_ablate.py --one <bits>, default the
# product code), one rocprofv3 --pmc run per counter group, summed over the kernel's dispatches.
#   gpurun --timeout 900 -- 'bash tools/pmc_fast.sh > gpurun_out/pmc_fast.txt'
bits=${1:-0}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
         "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_MFMA_I8"; do
rm -rf /tmp/pmf; timeout 300 rocprofv3 --pmc $C -d /tmp/pmf -o pm --output-format csv -- python $root/tools/orb_ablate.py --one $bits > /dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=0
for fn in glob.glob("/tmp/pmf/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_fast_keypoints" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n+=1
print({c: round(v) for c,v in agg.items()}, "rows", n)
PY
done
