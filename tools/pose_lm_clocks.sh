#!/bin/bash
# Development probe: the phases of the one-launch pose-only solve (k_pose_lm) in shader clocks -- a second build of the library with
# -DPOSE_LM_CLOCKS (thread 0 prints the clock at every phase), run under the native caller of the tracker's shape.
#   here:        bash tools/pose_lm_clocks.sh build
#   on the GPU:  gpurun -- 'bash tools/pose_lm_clocks.sh run'
set -e
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "${1:-}" = build ]; then
    mkdir -p "$root/tools/_bin/clk"
    objs=""
    for f in "$root"/mageslam_amd/csrc/*.hip; do
        b=$(basename "$f")
        if [ "$b" = ba_kernels.hip ]; then
            /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I"$root/include" -DPOSE_LM_CLOCKS -c "$f" -o "$root/tools/_bin/clk/$b.o"
            objs="$objs $root/tools/_bin/clk/$b.o"
        else
            objs="$objs $root/mageslam_amd/csrc/_obj/$b.o"
        fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_bin/clk/libmageslam_hip.so" $objs
    exit 0
fi
cd "$root"
python - <<'PY'
import sys; sys.path.insert(0, '.')
import bench
from mageslam_amd import scene
scene.save_scene(scene.make_scene(**bench.SMALL_SHAPES["pose_only"]["scene"]), "/tmp/po.scene")
PY
LD_LIBRARY_PATH="$root/tools/_bin/clk" "$root/tools/_bin/shim_small_shapes" pose-only /tmp/po.scene 3 2>/dev/null | grep -v amdgpu | tail -80
