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
OUT_SPILLTEST = os.path.join(HERE, "libcimbar_hip_spilltest.so")   # same code, tiny LDS heap: the flood kernel's spill path under test
HEADER = os.path.join(os.path.dirname(HERE), "include", "cimbar_hip.h")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the cimbar HIP library cannot be built (there is no CPU fallback)")


def sources():
    """the translation unit, the sections it includes (csrc/*.hip.inc) and the public header"""
    d = os.path.dirname(SRC)
    return [SRC, HEADER] + sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".hip.inc"))


def needs_build(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in sources())


def build_hip(force=False, verbose=False, out=OUT, defines=()):
    """Compile csrc/cimbar_hip.hip for gfx950 into libcimbar_hip.so. Returns the .so path."""
    if not force and not needs_build(out):
        return out
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", *[f"-D{d}" for d in defines], "-o", out + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


INGEST_SRC = os.path.join(HERE, "csrc", "ingest.cpp")
INGEST_OUT = os.path.join(HERE, "libcimbar_ingest.so")
INGEST_HEADER = os.path.join(os.path.dirname(HERE), "include", "cimbar_ingest.h")


def build_ingest(force=False, verbose=False):
    """libcimbar_ingest.so: the host-side PNG pool + pinned ring in front of the device path (plain C++, g++; HIP runtime API only)."""
    deps = [INGEST_SRC, os.path.join(HERE, "csrc", "jpeg.inc"), INGEST_HEADER, HEADER]
    if not force and os.path.exists(INGEST_OUT) and os.path.getmtime(INGEST_OUT) >= max(os.path.getmtime(p) for p in deps) \
            and os.path.getmtime(INGEST_OUT) >= os.path.getmtime(OUT):
        return INGEST_OUT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", "-o", INGEST_OUT + ".tmp",
           INGEST_SRC, f"-L{HERE}", "-lcimbar_hip", f"-L{rocm}/lib", "-lamdhip64", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{rocm}/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(INGEST_OUT + ".tmp", INGEST_OUT)
    return INGEST_OUT


RECV_SRC = os.path.join(HERE, "host", "cimbar_recv_c.cpp")
RECV_OUT = os.path.join(HERE, "libcimbar_recv_hip.so")
RECV_HEADER = os.path.join(os.path.dirname(HERE), "include", "cimbar_recv_hip.h")


def build_recv(force=False, verbose=False):
    """libcimbar_recv_hip.so: the reference's own receive-side C symbols for the decode step (cimbard_configure_decode / _get_bufsize /
    _scan_extract_decode, cimbar_recv_js.h:16-17,36) over libcimbar_hip.so -- plain C++ (g++), include/cimbar_recv_hip.h."""
    deps = [RECV_SRC, RECV_HEADER, HEADER]
    if not force and os.path.exists(RECV_OUT) and os.path.getmtime(RECV_OUT) >= max(os.path.getmtime(p) for p in deps) \
            and os.path.getmtime(RECV_OUT) >= os.path.getmtime(OUT):
        return RECV_OUT
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", RECV_OUT + ".tmp", RECV_SRC,
           f"-L{HERE}", "-lcimbar_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(RECV_OUT + ".tmp", RECV_OUT)
    return RECV_OUT


def build_spilltest(force=False, verbose=False):
    """The test-only variant whose flood heap keeps 1024 slots in LDS (everything deeper goes through the spill path) and whose anchor search keeps
    ONE hit per scan row (so that ordinary captures overflow the fast kernels' lists and take the serial slow path, k_scan_serial)."""
    return build_hip(force, verbose, OUT_SPILLTEST, ("CIMBAR_HEAP_LDS=1024", "CIMBAR_SCAN_TINY_LISTS"))


def build_probes(force=False, verbose=False):
    """libcimbar_amd/variants/libcimbar_hip_probes.so: the same source with -DCIMBAR_PROBES -- the only build in which CIMBAR_HIP_DEBUG_SKIP (chain
    kernels dropped, results wrong on purpose) and the k_rs<.., NOSYND> timing instances exist. Selected with CIMBAR_HIP_LIB=<that file>
    (tools/skip_probe.sh); never loaded by default, and bench.py refuses to print a line while CIMBAR_HIP_DEBUG_SKIP is set."""
    d = os.path.join(HERE, "variants")
    os.makedirs(d, exist_ok=True)
    return build_hip(force, verbose, os.path.join(d, "libcimbar_hip_probes.so"), ("CIMBAR_PROBES",))


if __name__ == "__main__":
    if "--probes" in sys.argv:
        print(build_probes(force="--force" in sys.argv, verbose=True))
        sys.exit(0)
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_spilltest(force="--force" in sys.argv, verbose=True))
    print(build_ingest(force="--force" in sys.argv, verbose=True))
    print(build_recv(force="--force" in sys.argv, verbose=True))
