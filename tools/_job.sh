R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
bash tools/pmc_front_end.sh > $O/front_end_pmc.json 2> $O/front_end_pmc.err
cd /tmp; python -c "
import sys; sys.path.insert(0, '$R')
from mageslam_amd import scene
scene.save_scene(scene.make_scene(n_cams=12, n_pts=400, n_obs=2000, seed=0x5EED0012, fixed=(0,1,2,3), outlier_frac=0.02), '/tmp/window.scene')
"
MAGE_BA_TIMING=1 $R/tools/_bin/shim_small_shapes window /tmp/window.scene 3 2>&1 | tail -30 > $O/r04_window_phases.txt
rm -rf $O/kt_small; timeout 120 rocprofv3 --kernel-trace -d $O/kt_small -o x -- $R/tools/_bin/shim_small_shapes window /tmp/window.scene 50 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $O/kt_small -name "*.db" | head -1) > $O/r04_window_kernels2.txt; rm -rf $O/kt_small
cd $R; python -m pytest tests -m gpu -x -q --deselect tests/test_soak_gpu.py --deselect tests/test_bench_gpu.py 2>&1 | tail -5 > $O/r04_tests_b.txt
