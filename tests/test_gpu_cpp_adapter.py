"""GPU: the C++ host adapter (libcimbar_amd/host/Decoder.h) built with g++ against libcimbar_hip.so and driven like the
reference's own callers drive Decoder."""
import os
import subprocess

import numpy as np
import pytest

from libcimbar_amd import decoder
from oracle import pyref
from tests import frames as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_decoder_adapter(tmp_path, synth, hip_decoder):
    exe = tmp_path / "test_adapter"
    libdir = os.path.dirname(decoder.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp"),
                    "-L" + libdir, "-lcimbar_hip", "-Wl,-rpath," + libdir], check=True)
    n = 3
    payload, frames = F.clean_frames(synth, n, seed=2024)
    frames.tofile(tmp_path / "frames.bin")
    payload.tofile(tmp_path / "payload.bin")
    pyref.oracle_decode(frames[0])
    sym, col, _ = pyref.oracle_stage()
    np.concatenate([sym, col]).astype(np.uint8).tofile(tmp_path / "cells.bin")
    res = subprocess.run([str(exe), str(tmp_path / "frames.bin"), str(tmp_path / "payload.bin"), str(tmp_path / "cells.bin"), str(n)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.startswith("OK"), res.stdout + res.stderr
