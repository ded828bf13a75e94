#!/bin/bash
# Matrix-core, wait-state and memory-side counters of the dense factorisation's kernels, per kernel, from separate rocprofv3 --pmc
# passes over tools/_bin/chol_test (3 factorisations of the padded order 6016).  ON THE GPU BOX:
#   gpurun --timeout 900 -- 'bash tools/pmc_chol.sh r03'      -> gpurun_out/<tag>_chol_pmc.txt (copy into profiles/)
# Counters are collected in their own runs (no --sys-trace / --hip-trace beside --pmc).
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
res=$out/${tag}_chol_pmc.txt
{
echo "# rocprofv3 --pmc <counters> -- tools/_bin/chol_test 6016 3   (one pass per line below; 3 factorisations + solves of the padded order 6016 per run)"
echo "# available memory-side counters on this box:"
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC_EA0_[A-Z_0-9]*(DRAM|MALL|IO)[A-Z_0-9]*|TCC_[A-Z_]*MALL[A-Z_0-9]*|TCC_HIT_sum|TCC_MISS_sum|TCC_EA0_RDREQ_sum|TCC_EA0_RDREQ_32B_sum|TCC_EA0_WRREQ_sum|TCC_EA0_WRREQ_64B_sum)\b" | sort -u | tr '\n' ' '
echo
} > "$res"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE" "MfmaUtil" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"; do
    d=/tmp/pmc_chol_$$; rm -rf $d
    if timeout 300 rocprofv3 --pmc $set -d $d -o p -- "$root/tools/_bin/chol_test" 6016 3 > /dev/null 2> /tmp/pmc_chol.err; then
        db=$(find $d -name '*.db' | head -1)
        if [ -n "$db" ]; then python "$root/tools/rocpd_pmc.py" "$db" | sed "s#^\# source: .*#\# pass: --pmc $set#" >> "$res"; else echo "# pass: --pmc $set -> no database" >> "$res"; fi
    else
        echo "# pass: --pmc $set -> rocprofv3 failed: $(tail -c 200 /tmp/pmc_chol.err | tr '\n' ' ')" >> "$res"
    fi
    echo >> "$res"
    rm -rf $d
done
cat "$res"
