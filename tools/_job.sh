cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, tempfile
sys.path.insert(0, '.')
import bench
from mageslam_amd import scene
cfg = bench.SMALL_SHAPES["reference_window"]
d = tempfile.mkdtemp()
path = os.path.join(d, "w.scene")
scene.save_scene(scene.make_scene(**cfg["scene"]), path)
env = dict(os.environ, MAGE_BA_TIMING="1")
p = subprocess.run(["tools/_bin/shim_small_shapes", "window", path, "30"], capture_output=True, text=True, env=env)
lines = p.stderr.splitlines()
print("\n".join(lines[-45:]))
print(p.stdout[-600:])
PY
cd /tmp && export TMPDIR=/tmp
python - <<'PY'
import os, sys, subprocess, tempfile
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
os.chdir(os.environ["GRAFT_REPO_ROOT"])
import bench
from mageslam_amd import scene
cfg = bench.SMALL_SHAPES["reference_window"]
path = "/tmp/w.scene"
scene.save_scene(scene.make_scene(**cfg["scene"]), path)
PY
rm -rf /tmp/p1; rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/p1 -o b -- $GRAFT_REPO_ROOT/tools/_bin/shim_small_shapes window /tmp/w.scene 200 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p1 -name '*.db' | head -1) | head -14
