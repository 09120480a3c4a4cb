"""GPU: one PROCESS driving several GPUs through the C library alone (cimbar_hip_comm_init_all + cimbar_hip_gather_chunks: the library's own
RCCL exchange, what a C++ ./cimbar would use) -- tests/cpp/test_comm_all.cpp. The multi-device run needs >= 2 GPUs and skips on the one-GPU boxes
the builder gets; the same program with one device (a one-rank communicator) runs everywhere."""
import os
import subprocess

import numpy as np
import pytest
import torch

from libcimbar_amd import decoder, framegen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = tmp_path / "test_comm_all"
    libdir = os.path.dirname(decoder.LIB_PATH)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "test_comm_all.cpp"),
                    "-L" + libdir, "-lcimbar_hip", "-Wl,-rpath," + libdir, f"-L{rocm}/lib", "-lamdhip64", f"-Wl,-rpath,{rocm}/lib", "-lpthread"], check=True)
    framegen.FrameSynth("cpu").template.numpy().tofile(tmp_path / "template.bin")
    return exe


def test_comm_init_all_with_one_device(tmp_path):
    exe = build(tmp_path)
    res = subprocess.run([str(exe), "1", "5", str(tmp_path / "template.bin")], capture_output=True, text=True, timeout=300)
    # (RCCL may print its version banner to stdout first, depending on the box's environment)
    assert res.returncode == 0 and any(l.startswith("OK") for l in res.stdout.splitlines()), res.stdout + res.stderr


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process (the builder's boxes have one)")
def test_comm_init_all_gathers_across_every_gpu_of_the_node(tmp_path):
    exe = build(tmp_path)
    ndev = min(torch.cuda.device_count(), 8)
    res = subprocess.run([str(exe), str(ndev), "64", str(tmp_path / "template.bin")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    # (RCCL may print its version banner to stdout first, depending on the box's environment)
    assert res.returncode == 0 and any(l.startswith("OK") for l in res.stdout.splitlines()), res.stdout + res.stderr
