{
for i in 1 2; do
echo "== merge2 off, wt off"; MAGE_CHOL_MERGE2=0 MAGE_CHOL_WT_HANDOFF=0 tools/_bin/chol_test 6016 10
echo "== merge2 off, wt on"; MAGE_CHOL_MERGE2=0 tools/_bin/chol_test 6016 10
echo "== merge2 on(56), wt off"; MAGE_CHOL_MERGE2_AT_US=56 MAGE_CHOL_WT_HANDOFF=0 tools/_bin/chol_test 6016 10
echo "== merge2 on, strips skip compute"; MAGE_CHOL_MERGE2_AT_US=56 MAGE_CHOL_WT_HANDOFF=0 MAGE_CHOL_MERGE2_DBG=4 tools/_bin/chol_test 6016 10
echo "== merge2 on, no release"; MAGE_CHOL_MERGE2_AT_US=56 MAGE_CHOL_WT_HANDOFF=0 MAGE_CHOL_MERGE2_DBG=8 tools/_bin/chol_test 6016 10
echo "== merge2 on, both"; MAGE_CHOL_MERGE2_AT_US=56 MAGE_CHOL_WT_HANDOFF=0 MAGE_CHOL_MERGE2_DBG=12 tools/_bin/chol_test 6016 10
done
echo "== stamps wt off"; MAGE_CHOL_MERGE2=0 MAGE_CHOL_WT_HANDOFF=0 CHOL_DBG=1 CHOL_DBG_COL=30 tools/_bin/chol_test 6016 3 | grep -v "col "
echo "== stamps wt on"; MAGE_CHOL_MERGE2=0 CHOL_DBG=1 CHOL_DBG_COL=30 tools/_bin/chol_test 6016 3 | grep -v "col "
for nn in 256 640 1408 3712; do MAGE_CHOL_MERGE2=0 tools/_bin/chol_test $nn 3;  MAGE_CHOL_MERGE2=0 MAGE_CHOL_WT_HANDOFF=0 tools/_bin/chol_test $nn 3; done
} > gpurun_out/r04_ab3.txt 2>&1
