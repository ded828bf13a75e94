cd $GRAFT_REPO_ROOT
timeout 300 python tools/small_shapes.py 2>&1 | python -c "
import sys, json
d=json.load(sys.stdin)
print('hip window', d['hip']['reference_window']['window_ms'], d['hip']['reference_window']['window_result'])
print('cpu window', d['cpu_oracle']['reference_window']['window_ms'], d['cpu_oracle']['reference_window']['window_result'])
print('hip pose', d['hip']['pose_only']['pose_only_pass1_ms']['total'], d['hip']['pose_only']['pose_only_pass2_ms']['total'])
"
MAGE_BA_NO_RESULT_RIDE=1 timeout 300 python tools/small_shapes.py 2>&1 | python -c "
import sys, json
d=json.load(sys.stdin)
print('no-ride hip window', d['hip']['reference_window']['window_ms'], d['hip']['reference_window']['window_result'])
"
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_abi.py tests/test_concurrency_gpu.py -q -x -m gpu 2>&1 | tail -3
( timeout 600 python tools/fuzz_ba.py --cases 4000 --seed 61 2>&1 | grep -v amdgpu | tail -4 )
