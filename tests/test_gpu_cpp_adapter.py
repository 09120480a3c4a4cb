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
    # the same classes in mode 67 (1024x720): frames, payload, and a 1080p capture of the first frame
    from libcimbar_amd import framegen
    p67, f67 = F.clean_frames(framegen.FrameSynth("cpu", 67), n, seed=2025)
    f67.tofile(tmp_path / "frames67.bin")
    p67.tofile(tmp_path / "payload67.bin")
    np.ascontiguousarray(F.camera_frame(f67[0], quad=((300, 150), (1600, 170), (290, 930), (1620, 915)), background=15)).tofile(tmp_path / "cap67.bin")
    res = subprocess.run([str(exe), str(tmp_path / "frames.bin"), str(tmp_path / "payload.bin"), str(tmp_path / "cells.bin"), str(n),
                          str(tmp_path / "frames67.bin"), str(tmp_path / "payload67.bin"), str(tmp_path / "cap67.bin")],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.startswith("OK"), res.stdout + res.stderr


@pytest.mark.parametrize("mode", [68, 67, 66, 4, 8])
def test_cpp_adapter_in_every_mode(tmp_path, mode):
    """tests/cpp/test_adapter_modes.cpp: Decoder / CimbDecoder / CimbReader / load_ccm / a cv::UMat-shaped image in every mode the library
    accepts, chunk space sized by cimbar_hip_ctx_bufsize -- built plain and with AddressSanitizer (mode 8's frame is 8750 bytes, larger than
    mode B's 7500 that the adapter's fixed buffers used to have)."""
    from libcimbar_amd import framegen
    libdir = os.path.dirname(decoder.LIB_PATH)
    src = os.path.join(ROOT, "tests", "cpp", "test_adapter_modes.cpp")
    exe, exe_asan = tmp_path / "test_adapter_modes", tmp_path / "test_adapter_modes_asan"
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), src, "-L" + libdir, "-lcimbar_hip", "-Wl,-rpath," + libdir], check=True)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-o", str(exe_asan), src,
                    "-L" + libdir, "-lcimbar_hip", "-Wl,-rpath," + libdir], check=True)
    n = 2
    payload, frames = F.clean_frames(framegen.FrameSynth("cpu", mode), n, seed=3000 + mode)
    frames.tofile(tmp_path / "frames.bin")
    payload.tofile(tmp_path / "payload.bin")
    pyref.oracle_decode(frames[0], mode=mode)
    sym, col, _ = pyref.oracle_stage(mode=mode)
    np.concatenate([sym, col]).astype(np.uint8).tofile(tmp_path / "cells.bin")
    args = [str(mode), str(tmp_path / "frames.bin"), str(tmp_path / "payload.bin"), str(tmp_path / "cells.bin"), str(n), str(tmp_path)]
    res = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.startswith("OK"), res.stdout + res.stderr
    # the same under AddressSanitizer: the HIP runtime's own allocations are not instrumented and its exit-time leaks are not ours
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0")
    res = subprocess.run([str(exe_asan)] + args, capture_output=True, text=True, timeout=600, env=env)
    assert "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
    assert res.returncode == 0 and res.stdout.startswith("OK"), res.stdout + res.stderr


def test_unknown_mode_is_mode_b_like_the_reference(hip_decoder):
    """Config::temp_conf's `default:` (Config.h:41-43): any value that is no listed mode gives Conf8x8"""
    from libcimbar_amd import HipDecoder
    for mode_val in (0, 5, 69, -1):
        d = HipDecoder(0, mode_val)
        assert d.geo.MODE == 68 and d.bufsize() == 7500
        d.close()


@pytest.mark.parametrize("mode", [68, 67, 66, 4, 8])
def test_dropin_against_the_reference_headers_in_every_mode(tmp_path, ref, mode):
    """tests/cpp/test_dropin.cpp with the reference configured for `mode` (Config::update): the reference's encoder makes the frames, the
    adapter decodes them into the reference's sinks -- cv::Mat and cv::UMat, the decode loop of cimbar.cpp:124-171 as written there"""
    exe = os.path.join(ROOT, "oracle", "_ref", "test_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_dropin not built (needs /root/reference at build time: make -C oracle dropin)")
    from oracle.pyref import P
    from libcimbar_amd import geometry
    geo = geometry.for_mode(mode)
    size, n = 40000, 16
    data = np.random.default_rng(770 + mode).integers(0, 256, size, dtype=np.uint8)
    frames = np.zeros((n, geo.IMG_H, geo.IMG_W, 3), np.uint8)
    with pyref.ref_mode(mode):
        assert ref.ref_encode_fountain(P(data), size, 9, 0, n, P(frames)) == n
    frames.tofile(tmp_path / "frames.bin")
    data.tofile(tmp_path / "file.bin")
    cams = np.ascontiguousarray(np.stack([F.camera_frame(frames[k], quad=((500, 40), (1480, 70), (470, 1030), (1500, 1000)), background=10) for k in range(2)]))
    cams.tofile(tmp_path / "caps.bin")
    res = subprocess.run([exe, str(tmp_path / "frames.bin"), str(n), str(tmp_path / "file.bin"), str(tmp_path / "caps.bin"), "1920", "1080", "2"],
                         capture_output=True, text=True, timeout=300, env=dict(os.environ, CIMBAR_MODE=str(mode)))
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


def test_dropin_against_the_reference_headers(tmp_path, synth, ref):
    """tests/cpp/test_dropin.cpp, compiled in the build container against the reference's own headers (cv::Mat via the shim,
    fountain_decoder_sink, concurrent_fountain_decoder_sink + wirehair): a wirehair stream of a file, rendered by the reference encoder, goes
    through cimbar_amd::Decoder::decode_fountain into the reference's sinks and the file comes back; captures go through cimbar_amd::Extractor"""
    exe = os.path.join(ROOT, "oracle", "_ref", "test_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_dropin not built (needs /root/reference at build time: make -C oracle dropin)")
    from oracle.pyref import P
    size, n = 60000, 14            # 97 wirehair blocks -> complete within 9 frames
    data = np.random.default_rng(77).integers(0, 256, size, dtype=np.uint8)
    frames = np.zeros((n, 1024, 1024, 3), np.uint8)
    assert ref.ref_encode_fountain(P(data), size, 9, 0, n, P(frames)) == n
    frames.tofile(tmp_path / "frames.bin")
    data.tofile(tmp_path / "file.bin")
    cams = np.ascontiguousarray(np.stack([F.camera_frame(frames[k], quad=((500, 40), (1480, 70), (470, 1030), (1500, 1000)), background=10) for k in range(2)]))
    cams.tofile(tmp_path / "caps.bin")
    res = subprocess.run([exe, str(tmp_path / "frames.bin"), str(n), str(tmp_path / "file.bin"), str(tmp_path / "caps.bin"), "1920", "1080", "2"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr
