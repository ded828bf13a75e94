cd $GRAFT_REPO_ROOT
T=tools/_bin/chol_test
echo "== 4w"; timeout 120 $T 6016 10 | grep "n= 6016"
echo "== 1w"; MAGE_CHOL_STRIP_1W=1 timeout 120 $T 6016 10 | grep "n= 6016"
echo "== 4w"; timeout 120 $T 6016 10 | grep "n= 6016"
echo "== 1w"; MAGE_CHOL_STRIP_1W=1 timeout 120 $T 6016 10 | grep "n= 6016"
echo "== all sizes 4w"; timeout 200 $T | tail -12
for col in 36; do
echo "== stamps 4w col $col"; CHOL_DBG=1 CHOL_DBG_COL=$col timeout 120 $T 6016 3 | grep -A12 "launch $col"
echo "== stamps 1w col $col"; MAGE_CHOL_STRIP_1W=1 CHOL_DBG=1 CHOL_DBG_COL=$col timeout 120 $T 6016 3 | grep -A12 "launch $col"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1; rocprofv3 --kernel-trace -d /tmp/p1 -o b -- $GRAFT_REPO_ROOT/$T 6016 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p1 -name '*.db' | head -1) | head -10
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_chol_gpu.py -q -x 2>&1 | tail -3
