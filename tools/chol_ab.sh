#!/bin/bash
# A/B of the dense solve's schedule switches on the GPU box: tools/chol_ab.sh <out-file> [n] [reps]
out=${1:-gpurun_out/chol_ab.txt}; n=${2:-6016}; reps=${3:-10}
{
for i in 1 2; do
  echo "== merged half-tile panel solve OFF"; MAGE_CHOL_MERGE2=0 tools/_bin/chol_test $n $reps
  for at in 40 48 56 64; do echo "== merged, strips at $at us"; MAGE_CHOL_MERGE2_AT_US=$at tools/_bin/chol_test $n $reps; done
done
for nn in 1408 2944 3712 9088; do echo "== n=$nn off / on"; MAGE_CHOL_MERGE2=0 tools/_bin/chol_test $nn 5; tools/_bin/chol_test $nn 5; done
} > $out 2>&1
