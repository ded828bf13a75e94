#!/bin/bash
# Ablation builds of k_schur_stream (mageslam_amd/csrc/ba_kernels.hip, SCHUR_ABL): 1 = no record / list loads (records made from the lane
# number: the arithmetic, the claims and the reductions alone), 2 = no arithmetic (the loads, the loop and the reductions alone), 3 = as 2 with the same bytes requested cooperatively (three lanes per
# landmark record, two per slot record: a third of the distinct lines per load instruction -- the time does not move: the loop is bound by
# the LATENCY of a trip's requests, ~1.4 us each, not by their number); each as a
# copy of the product library under tools/_bin/, timed by three LM iterations of the 1k-pose map under rocprofv3 --kernel-trace (results
# are garbage by construction: the step may report an indefinite system).     bash tools/schur_ablate.sh        (on the GPU box)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p tools/_bin
for n in 1 2 3; do
python - "$n" <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from mageslam_amd import build as B
n = sys.argv[1]
B.build()
src = os.path.join(os.getcwd(), "mageslam_amd", "csrc", "ba_kernels.hip")
obj = os.path.join(os.getcwd(), "tools", "_bin", f"abl_{n}.o")
subprocess.check_call([B.HIPCC, *B.flags_for(src), f"-DSCHUR_ABL={n}", "-c", src, "-o", obj])
objs = [os.path.join(B.OBJ, os.path.basename(s) + ".o") for s in B.sources() if not s.endswith("ba_kernels.hip")] + [obj]
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(os.getcwd(), "tools", "_bin", f"libabl_{n}.so"), *objs])
PY
done
[ "$1" = "--build-only" ] && exit 0
cd /tmp && export TMPDIR=/tmp
for L in "" tools/_bin/libabl_1.so tools/_bin/libabl_2.so tools/_bin/libabl_3.so; do
    rm -rf /tmp/kt
    (cd "$root" && MAGE_LIB=${L:+$root/$L} rocprofv3 --kernel-trace -d /tmp/kt -o b -- python tools/schur_time.py > /dev/null 2>&1)
    echo "library: ${L:-product}"
    python "$root/tools/rocpd_stats.py" "$(find /tmp/kt -name '*.db' | head -1)" 2>&1 | grep -i "schur_stream\|schur_block" | cut -c1-170
done
