cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
rm -rf /tmp/pm; rocprofv3 --pmc $C -d /tmp/pm -o pm --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_orb.py > /dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("/tmp/pm/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"]
        if "k_fast_keypoints" not in k and "k_select" not in k and "k_brief" not in k: continue
        if int(r["Grid_Size"]) < 256*200*2048: 
            if "k_fast" in k: continue
        agg[k[:48]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k in agg: print(k, {c: round(v) for c,v in agg[k].items()})
PY
done
