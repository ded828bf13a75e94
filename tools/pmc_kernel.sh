#!/bin/bash
# Counter passes for ONE kernel of a command: one rocprofv3 --pmc run per counter group (quoted, space-separated), values summed over
# the dispatches whose name contains the pattern and divided by the number of dispatches.
#   bash tools/pmc_kernel.sh k_schur_block "SQ_WAVES SQ_INSTS_VALU" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" -- python bench.py --no-extras --no-cpu-baseline
pat=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for C in "${groups[@]}"; do
rm -rf /tmp/pmk; (cd $root && timeout 600 rocprofv3 --pmc $C -d /tmp/pmk -o pm --output-format csv -- "$@" > /dev/null 2>&1)
python - "$pat" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(float); disp=set()
for fn in glob.glob("/tmp/pmk/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        if sys.argv[1] not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n=max(len(disp),1)
print({c: round(v/n) for c,v in agg.items()}, "dispatches", len(disp))
PY
done
