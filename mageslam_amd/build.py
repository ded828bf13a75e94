"""Builds libmageslam_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  Usage:  python -m mageslam_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmageslam_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(os.path.dirname(HERE), "include")]


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    files += [os.path.join(inc, f) for f in os.listdir(inc)]
    return max(os.path.getmtime(f) for f in files)


# per-file additions: the i8 matrix-core products of the fused blur keep their accumulators in ordinary vector registers (the
# default puts them in accumulation registers and moves all 24 results back one v_accvgpr_read at a time -- in a kernel bound by
# vector-ALU issue)
EXTRA_FLAGS = {"orb_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


_flag_ok: dict[tuple, bool] = {}


def _accepted(extra: list[str]) -> bool:
    """An internal LLVM option (-mllvm ...) is a performance hint only: a compiler build that does not know it rejects the whole
    translation unit, so it is probed once on an empty file and dropped when refused."""
    key = tuple(extra)
    if key not in _flag_ok:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            open(src, "w").write("#include <hip/hip_runtime.h>\n__global__ void k() {}\n")
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", *extra, "-c", src, "-o", os.path.join(d, "probe.o")], capture_output=True)
            _flag_ok[key] = r.returncode == 0
        if not _flag_ok[key]:
            print(f"[build] {' '.join(extra)} not accepted by {HIPCC}: dropped (a performance hint only)", file=sys.stderr)
    return _flag_ok[key]


def flags_for(src: str) -> list[str]:
    extra = EXTRA_FLAGS.get(os.path.basename(src), [])
    return FLAGS + (extra if not extra or _accepted(extra) else [])


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(_deps_mtime(), os.path.getmtime(__file__)):
        return obj
    subprocess.check_call([HIPCC, *flags_for(src), "-c", src, "-o", obj])
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), sources()))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
