R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
{
for i in 1 2 3; do echo "quarters on"; tools/_bin/chol_test 6016 10; echo "quarters off"; MAGE_CHOL_NO_QUARTERS=1 tools/_bin/chol_test 6016 10; done
for nn in 3712 9088; do tools/_bin/chol_test $nn 3; MAGE_CHOL_NO_QUARTERS=1 tools/_bin/chol_test $nn 3; done
} > $O/r04_chol_quarters.txt 2>&1
python bench.py > $O/r04_bench_try1.json 2> $O/r04_bench_try1.err
MAGE_SOAK_SECONDS=20 python -m pytest tests/test_soak_gpu.py tests/test_bench_gpu.py tests/test_chol_gpu.py -m gpu -x -q -s 2>&1 | tail -15 > $O/r04_tests_c.txt
