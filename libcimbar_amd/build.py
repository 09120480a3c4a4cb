"""Build recipe for the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m libcimbar_amd.build

produces libcimbar_amd/libcimbar_hip.so in-tree (git-ignored, shipped to the GPU box by gpurun).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cimbar_hip.hip")
OUT = os.path.join(HERE, "libcimbar_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "cimbar_hip.h")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the cimbar HIP library cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in (SRC, HEADER))


def build_hip(force=False, verbose=False):
    """Compile csrc/cimbar_hip.hip for gfx950 into libcimbar_hip.so. Returns the .so path."""
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-o", OUT + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
