cd $GRAFT_REPO_ROOT
for mode in fold sep; do
  if [ $mode = sep ]; then export MAGE_BA_SEPARATE_REDUCE=1; else unset MAGE_BA_SEPARATE_REDUCE; fi
  python bench.py --no-cpu-baseline --no-extras > gpurun_out/fold_$mode.json 2> gpurun_out/fold_$mode.err
  python - <<PY
import json
b=json.load(open('gpurun_out/fold_$mode.json'))
print('$mode', b['value'], b['ms_per_step'], b['final_reproj_rmse_px'], {k:v['ms'] for k,v in b['roofline_hbm']['stages'].items()} if 'stages' in b['roofline_hbm'] else '')
PY
done
unset MAGE_BA_SEPARATE_REDUCE
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1; rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p1 -name '*.db' | head -1) | sed -n 6,16p
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_build_gpu.py tests/test_sharded_gpu.py tests/test_windowed_gpu.py tests/test_concurrency_gpu.py -q -x 2>&1 | tail -3
