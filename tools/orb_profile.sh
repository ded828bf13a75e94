#!/bin/bash
# ON THE GPU BOX: the front-end's committed measurements of a round.
#   gpurun_out/<tag>_orb_match_bench.jsonl   tools/bench_orb.py at batch 1 / 64 / 1024 (its own lines)
#   gpurun_out/<tag>_orb_kernel_stats.txt    per-kernel table of the same command under rocprofv3 --kernel-trace
#   gpurun_out/<tag>_orb_phases.txt          in-kernel phase clocks (tools/orb_phase_probe.py; needs the probe build: --build here first)
#   gpurun --timeout 600 -- 'bash tools/orb_profile.sh r05'
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python "$root/tools/bench_orb.py" > "$out/${tag}_orb_match_bench.jsonl" 2> "$out/${tag}_orb.err"
rm -rf /tmp/po
timeout 300 rocprofv3 --kernel-trace -d /tmp/po -o b -- python "$root/tools/bench_orb.py" > /dev/null 2>> "$out/${tag}_orb.err"
python "$root/tools/rocpd_stats.py" "$(find /tmp/po -name '*.db' | head -1)" > "$out/${tag}_orb_kernel_stats.txt"
[ -f "$root/mageslam_amd/_probe/libmageslam_hip_clk.so" ] && python "$root/tools/orb_phase_probe.py" > "$out/${tag}_orb_phases.txt" 2>> "$out/${tag}_orb.err"
head -8 "$out/${tag}_orb_kernel_stats.txt"; cat "$out/${tag}_orb_match_bench.jsonl" | cut -c1-600
