R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
{
echo "== chol (factor_block16 inlined)"
for i in 1 2 3; do tools/_bin/chol_test 6016 10; done
for nn in 128 640 1408 3712 9088; do tools/_bin/chol_test $nn 3; done
echo "== stamps"; CHOL_DBG=1 CHOL_DBG_COL=30 tools/_bin/chol_test 6016 3 | grep -v "col "
} > $O/r04_chol_inl.txt 2>&1
for i in 1 2; do
python bench.py --no-extras --no-cpu-baseline > $O/r04_bench_skyline_$i.json 2>/dev/null
MAGE_BA_ZERO_FULL=1 python bench.py --no-extras --no-cpu-baseline > $O/r04_bench_fullzero_$i.json 2>/dev/null
done
python -m pytest tests/test_ba_gpu.py tests/test_chol_gpu.py tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/r04_tests_a.txt
python tools/small_latency.py >> $O/r04_tests_a.txt 2>&1
