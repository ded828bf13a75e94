#!/bin/bash
# The Schur build's evidence for one round: A/B of the two launches, per-wavefront stamps, the three ablation builds, counters of k_schur_stream.
#   bash tools/schur_profile.sh r06        (on the GPU box; writes gpurun_out/<round>_schur_stream.txt)
R=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p gpurun_out
out=gpurun_out/${R}_schur_stream.txt
{
echo "# tools/schur_profile.sh $R -- 1k poses / 100k points / 1M observations: 15 522 Schur blocks, 5.49 M contributions, 95 730 trips of 64"
echo "## tools/ab_schur.py (stage span = k_schur_prepare + the blocks' kernel, HIP events, 20 LM iterations; state hash after them)"
timeout 300 python tools/ab_schur.py global
echo "## MAGE_BA_SCHUR_TRACE=1 tools/schur_stamps.py"
MAGE_BA_SCHUR_TRACE=1 timeout 300 python tools/schur_stamps.py
echo "## tools/schur_ablate.sh (rocprofv3 --kernel-trace; columns: calls, total ms, mean us, min us, max us, %)"
timeout 900 bash tools/schur_ablate.sh 2>&1 | grep -v "^$\|warning\|^ *[0-9]* |\|^ *|\|generated"
echo "## tools/pmc_kernel.sh k_schur_stream (per dispatch; SQ_* cycle counters in quad-cycles)"
timeout 900 bash tools/pmc_kernel.sh k_schur_stream "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" -- python tools/schur_time.py
} > $out 2>&1
[ -s $out ] || { echo "empty $out"; exit 1; }
tail -5 $out
