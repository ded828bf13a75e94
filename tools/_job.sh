cd $GRAFT_REPO_ROOT
T=tools/_bin/chol_test
echo "== default (phased)"; timeout 120 $T 6016 10 | grep "n= 6016"
echo "== phased off"; MAGE_CHOL_PHASED_TRSM=0 timeout 120 $T 6016 10 | grep "n= 6016"
echo "== all sizes"; timeout 300 $T | tail -20
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-extras > gpurun_out/x.json 2> gpurun_out/x.err
python - <<PY
import json
b=json.load(open('gpurun_out/x.json'))
print(b['value'], b['ms_per_step'], b['final_reproj_rmse_px'], b['roofline']['frac'], b['roofline']['ms_per_launch'], b['stall_counters'])
PY
