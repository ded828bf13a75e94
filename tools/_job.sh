cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_windowed_gpu.py tests/test_concurrency_gpu.py tests/test_abi.py -q -x -m gpu 2>&1 | tail -3
( timeout 600 python tools/fuzz_ba.py --cases 3000 --seed 51 2>&1 | grep -v amdgpu | tail -4 )
for mode in lpt row; do
  if [ $mode = row ]; then export MAGE_BA_SCHUR_ROW_ORDER=1; else unset MAGE_BA_SCHUR_ROW_ORDER; fi
  python bench.py --no-cpu-baseline --no-extras > gpurun_out/x.json 2> gpurun_out/x.err
  python - <<PY
import json
b=json.load(open('gpurun_out/x.json'))
print('$mode', b['value'], b['ms_per_step'], b['final_reproj_rmse_px'], {k:v['ms'] for k,v in b['roofline_hbm']['stages'].items()})
PY
done
