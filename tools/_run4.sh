cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in off on; do
  rm -rf $R/gpurun_out/kt_$mode
  if [ $mode = off ]; then export MAGE_CHOL_MERGE2=0; else export MAGE_CHOL_MERGE2=1; fi
  MAGE_CHOL_MERGE2_AT_US=56 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$mode -o x -- $R/tools/_bin/chol_test 6016 3 > /dev/null 2>&1
  python $R/tools/rocpd_steps.py "$(find $R/gpurun_out/kt_$mode -name '*.db' | head -1)" > $R/gpurun_out/r04_steps_$mode.txt 2>&1
  rm -rf $R/gpurun_out/kt_$mode
done
