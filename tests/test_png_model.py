"""tools/png_model.cpp -- the CPU model of the device PNG decoder's two algorithms (lane-parallel canonical Huffman decode over 64-dword
input chunks, 32 KiB ring with 4 KiB flushes and running Adler-32, periodic overlapped copies; the skewed one-row-per-lane un-filter) --
against images whose pixels are known: every filter type, every deflate block type and zlib strategy, small windows, split IDATs, odd
sizes, Pillow's writer. The kernels in csrc/png.hip.inc restate the model; tests/test_gpu_png.py runs the same cases on the device."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import png_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("pngmodel") / "libpng_model.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "png_model.cpp")], check=True)
    L = ctypes.CDLL(so)
    L.png_model_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint),
                                   ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.c_float]
    return L


def decode(L, png, scale=1.0):
    buf = np.frombuffer(png, np.uint8)
    w, h, b = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    rc = L.png_model_decode(buf.ctypes.data, buf.size, None, 0, w, h, b, scale)
    if rc:
        return rc, None
    out = np.zeros((h.value, w.value, b.value), np.uint8)
    rc = L.png_model_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.size, w, h, b, scale)
    return rc, out


# (input chunk model, LDS ring): a plain dword stream | k_png_inflate's 64-dword chunks with the whole window or the 8 KiB ring (far matches come
# back from the flushed output) | k_png_inflate4's 16-dword chunks per row with 2 KiB / 4 KiB rings
KERNELS = [(0, 32768), (64, 32768), (64, 8192), (16, 2048), (16, 4096)]


@pytest.mark.parametrize("chunk,ring", KERNELS)
@pytest.mark.parametrize("scale", [1.0, 1.0 + 2e-7, 1.0 - 2e-7])   # the device's v_rcp_f32 is not exactly rounded: the j mod dist fix-up must absorb it
def test_model_decodes_every_case(model, synth, scale, chunk, ring):
    from tests.frames import clean_frames
    model.png_model_chunk(chunk)
    model.png_model_ring(ring)
    _p, frames = clean_frames(synth, 1, seed=5151)
    for name, png, want in png_cases.cases(frames[0], big=(scale == 1.0)):
        rc, got = decode(model, png, scale)
        assert rc == 0, (name, rc)
        if name == "palette":
            continue                     # (the model stops at the un-filtered indices)
        rgb = got if got.shape[2] == 3 else got[..., :3] if got.shape[2] == 4 else np.repeat(got, 3, axis=2)
        assert (rgb == want).all(), name


def test_model_refuses_damaged_streams(model):
    model.png_model_chunk(0)
    model.png_model_ring(32768)
    for name, png in png_cases.corrupt_cases():
        rc, _ = decode(model, png)
        assert rc != 0, name


@pytest.mark.parametrize("chunk,ring", KERNELS[1:])
def test_model_fuzz(model, chunk, ring):
    model.png_model_chunk(chunk)
    model.png_model_ring(ring)
    for name, png, want in png_cases.fuzz_cases(150, seed=ring + chunk):
        rc, got = decode(model, png)
        assert rc == 0, (name, rc)
        if name.endswith("_pal"):
            continue                     # a palette image: the model stops at the indices
        rgb = got if got.shape[2] == 3 else got[..., :3] if got.shape[2] == 4 else np.repeat(got, 3, axis=2)
        assert (rgb == want).all(), name


@pytest.mark.parametrize("ring", [8192, 32768])
def test_model_all_offsets_turn(model, synth, ring):
    """k_png_inflate<RING, true>'s turn (every lane decodes the token that would start at its bit of the window; literals stored ahead of the
    turn's matches; short runs filled by their own lanes) on every case, a fuzz batch and the streams built around the ring limits -- the model
    returns E_MODEL if a match's source could have been overwritten by something the turn stored ahead of it, or was not flushed yet."""
    from tests.frames import clean_frames
    model.png_model_chunk(64)
    model.png_model_ring(ring)
    model.png_model_par(1)
    try:
        _p, frames = clean_frames(synth, 1, seed=5151)
        batch = png_cases.cases(frames[0], big=True) + png_cases.fuzz_cases(200, seed=ring) + png_cases.ring_stress_cases()
        for name, png, want in batch:
            rc, got = decode(model, png)
            assert rc == 0, (name, rc)
            if name == "palette" or name.endswith("_pal"):
                continue
            rgb = got if got.shape[2] == 3 else got[..., :3] if got.shape[2] == 4 else np.repeat(got, 3, axis=2)
            assert (rgb == want).all(), name
            if name in ("frame_cvdefault", "pillow_frame_l1"):
                st = (ctypes.c_ulonglong * 8)()
                model.png_model_stats(st)
                # (nearly every token goes through the all-offsets turn, several per turn)
                assert st[7] > 0 and (st[0] + st[1]) / st[7] > 2.5, (name, list(st))
        for name, png in png_cases.corrupt_cases():
            rc, _ = decode(model, png)
            assert rc != 0, name
    finally:
        model.png_model_par(0)
