R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
( time python -m pytest tests/test_bench_gpu.py -m gpu -x -q -s ) > $O/r04_tests_e.txt 2>&1
