#!/bin/bash
# Regenerates, in one command ON THE GPU BOX, the profile artefacts bench.py and DESIGN.md cite:
#   gpurun_out/<tag>_bench.json                 the bench line
#   gpurun_out/<tag>_bench_kernel_stats.txt     per-kernel table of the same command under rocprofv3 --kernel-trace
#   gpurun_out/<tag>_chol_traffic.json          HBM bytes per factorisation (separate --pmc FETCH_SIZE / WRITE_SIZE passes)
#   gpurun_out/<tag>_ba_stage_traffic.json      HBM bytes per LM iteration of the linearise / Schur / back-substitution stages (same passes over bench.py)
# Copy what is to be judged into profiles/ afterwards (gpurun_out/ is scratch).
#   gpurun --timeout 900 -- 'bash tools/regen_profiles.sh r02'
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python "$root/bench.py" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
rm -rf "$out/${tag}_prof"
timeout 600 rocprofv3 --kernel-trace -d "$out/${tag}_prof" -o b -- python "$root/bench.py" --no-cpu-baseline --no-extras > "$out/${tag}_bench_under_rocprof.json" 2>> "$out/${tag}_bench.err"
python "$root/tools/rocpd_stats.py" "$(find "$out/${tag}_prof" -name '*.db' | head -1)" > "$out/${tag}_bench_kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$out/${tag}_pmc_$c"
    timeout 300 rocprofv3 --pmc $c -d "$out/${tag}_pmc_$c" -o p -- "$root/tools/_bin/chol_test" 6016 3 > /dev/null 2>> "$out/${tag}_bench.err"
done
# chol_test runs the factorisation reps + 0 extra times at a single size (the indefinite check only below n = 640)
python "$root/tools/pmc_to_traffic.py" "$(find "$out/${tag}_pmc_FETCH_SIZE" -name '*.db' | head -1)" "$(find "$out/${tag}_pmc_WRITE_SIZE" -name '*.db' | head -1)" 3 > "$out/${tag}_chol_traffic.json"
rm -rf "$out/${tag}_pmc_FETCH_SIZE" "$out/${tag}_pmc_WRITE_SIZE"
# HBM traffic of the HBM-bound stages of an LM iteration: the same two passes over the bench itself
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c -d "$out/${tag}_pmc_$c" -o p -- python "$root/bench.py" --no-cpu-baseline --no-extras > /dev/null 2>> "$out/${tag}_bench.err"
done
python "$root/tools/pmc_stage_traffic.py" "$(find "$out/${tag}_pmc_FETCH_SIZE" -name '*.db' | head -1)" "$(find "$out/${tag}_pmc_WRITE_SIZE" -name '*.db' | head -1)" > "$out/${tag}_ba_stage_traffic.json"
rm -rf "$out/${tag}_prof" "$out/${tag}_pmc_FETCH_SIZE" "$out/${tag}_pmc_WRITE_SIZE"
tail -c 600 "$out/${tag}_bench.json"; echo; head -12 "$out/${tag}_bench_kernel_stats.txt"; cat "$out/${tag}_chol_traffic.json"; head -8 "$out/${tag}_ba_stage_traffic.json"
# every artefact this script promises must exist and be non-empty (round 5 committed a 0-byte profile that DESIGN.md cited)
bad=0
for f in "$out/${tag}_bench.json" "$out/${tag}_bench_kernel_stats.txt" "$out/${tag}_bench_under_rocprof.json" "$out/${tag}_chol_traffic.json" "$out/${tag}_ba_stage_traffic.json"; do
    if [ ! -s "$f" ]; then echo "regen_profiles: $f is missing or empty" >&2; bad=1; fi
done
# ... and so must every profiles/ file that DESIGN.md or profiles/README.md cites
for f in $(grep -oh "profiles/[A-Za-z0-9_./-]*" "$root/DESIGN.md" "$root/profiles/README.md" 2>/dev/null | sed 's/[.,)]*$//' | sort -u); do
    case "$f" in *'*'*|*/) continue;; esac
    if [ -e "$root/$f" ] && [ ! -s "$root/$f" ]; then echo "regen_profiles: cited profile $f is empty" >&2; bad=1; fi
done
exit $bad

