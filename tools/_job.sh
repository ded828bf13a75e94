cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-extras > gpurun_out/wcm_cm.json 2> gpurun_out/wcm_cm.err
python - <<PY
import json
b=json.load(open('gpurun_out/wcm_cm.json'))
print(b['value'], b['ms_per_step'], b['final_reproj_rmse_px'], {k:v['ms'] for k,v in b['roofline_hbm']['stages'].items()} if 'stages' in b['roofline_hbm'] else '')
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1; rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p1 -name '*.db' | head -1) | sed -n 6,16p
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_build_gpu.py -q -x 2>&1 | tail -3
