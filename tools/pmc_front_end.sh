#!/bin/bash
# Dynamic instruction counts of the front-end kernels for bench.py's `valu_issue` rooflines: one rocprofv3 --pmc pass over
# tools/bench_orb.py (batches of 2 / 128 / 2048 frames, 1 / 64 / 1024 pairs), per kernel averaged over the dispatches of the LARGEST
# batch.  Prints one JSON object; together with tools/valu_mix.py --json it becomes profiles/front_end_valu.json:
#   gpurun -- 'bash tools/pmc_front_end.sh > gpurun_out/front_end_pmc.json'
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmfe
(cd $root && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/pmfe -o pm --output-format csv -- python tools/bench_orb.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob, collections, json
rows = collections.defaultdict(list)
for fn in glob.glob("/tmp/pmfe/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        for k in ("k_fast_keypoints", "k_select", "k_brief", "k_match"):
            if k + "<" in r["Kernel_Name"] or k + "(" in r["Kernel_Name"]:
                rows[k].append(r)
out = {}
for k, rs in rows.items():
    gmax = max(int(r["Grid_Size"]) for r in rs)
    big = [r for r in rs if int(r["Grid_Size"]) == gmax]
    disp = {r["Dispatch_Id"] for r in big}
    agg = collections.defaultdict(float)
    for r in big:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
    n = len(disp)
    out[k] = {"units_per_launch": 1024 if k == "k_match" else 2048, "unit": "frame pairs" if k == "k_match" else "frames", "grid_threads": gmax, "workgroup_size": int(big[0]["Workgroup_Size"]), "dispatches": n,
              "waves_per_launch": agg["SQ_WAVES"] / n, "valu_per_wave": agg["SQ_INSTS_VALU"] / max(agg["SQ_WAVES"], 1),
              "lds_per_wave": agg["SQ_INSTS_LDS"] / max(agg["SQ_WAVES"], 1), "salu_per_wave": agg["SQ_INSTS_SALU"] / max(agg["SQ_WAVES"], 1)}
print(json.dumps(out, indent=1))
PY
