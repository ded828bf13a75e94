R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
{
echo "== chol: gemm strips on / off / (wt off)"
for i in 1 2; do tools/_bin/chol_test 6016 10; MAGE_CHOL_GEMM_STRIPS=0 tools/_bin/chol_test 6016 10; MAGE_CHOL_WT_HANDOFF=0 tools/_bin/chol_test 6016 10; done
for nn in 128 640 1408 3712 9088; do tools/_bin/chol_test $nn 3; MAGE_CHOL_GEMM_STRIPS=0 tools/_bin/chol_test $nn 3; done
echo "== stamps gemm"; CHOL_DBG=1 CHOL_DBG_COL=30 tools/_bin/chol_test 6016 3 | grep -v "col "
echo "== stamps no gemm"; MAGE_CHOL_GEMM_STRIPS=0 CHOL_DBG=1 CHOL_DBG_COL=30 tools/_bin/chol_test 6016 3 | grep -v "col "
echo "== stamps wt off"; MAGE_CHOL_WT_HANDOFF=0 CHOL_DBG=1 CHOL_DBG_COL=30 tools/_bin/chol_test 6016 3 | grep -v "col "
} > $O/r04_chol_gemm3.txt 2>&1
python tools/small_shapes.py > $O/r04_small3.txt 2>&1
python -m pytest tests/test_ba_gpu.py tests/test_chol_gpu.py tests/test_sharded_gpu.py tests/test_windowed_gpu.py -m gpu -x -q 2>&1 | tail -5 >> $O/r04_small3.txt
