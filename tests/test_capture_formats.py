"""The `format` argument of the reference's C ABI (cimbard_scan_extract_decode, src/lib/cimbar_js/cimbar_recv_js.h:17; get_rgb,
cimbar_recv_js.cpp:94-120): the oracle's restatement of the four capture formats against the cv-shim's cvtColor, and the oracle's whole
capture chain against the reference's own entry point, which oracle/_ref links unmodified. OpenCV's YUV arithmetic is [assumed-OpenCV]
(tests/test_opencv_pin.py compares it with a real cv2 where one exists)."""
import ctypes

import numpy as np
import pytest

from oracle import pyref
from oracle.pyref import P
from tests import capture_formats as CF
from tests import frames as F
from tests.test_oracle_vs_ref import CAMERA_CASES

CV_CODE = {12: 90, 420: 98, 4: 1}       # COLOR_YUV2RGB_NV12, COLOR_YUV420p2RGB (= COLOR_YUV2RGB_YV12), COLOR_RGBA2RGB


@pytest.mark.parametrize("fmt", [4, 12, 420])
@pytest.mark.parametrize("size", [(16, 8), (64, 40), (70, 38), (130, 50), (1920, 1080)])      # heights % 4 == 0 and == 2: the chroma planes' two layouts
def test_oracle_conversion_equals_the_shims_cvtcolor(oracle, ref, fmt, size):
    w, h = size
    rng = np.random.default_rng(w * 1000 + h + fmt)
    img = rng.integers(0, 256, CF.capture_bytes(w, h, fmt), dtype=np.uint8)
    img[: w] = np.arange(w) % 256                     # every Y from 0 up, next to random chroma: both saturation ends
    oracle.co_capture_bytes.restype = ctypes.c_size_t
    assert oracle.co_capture_bytes(w, h, fmt) == img.size
    got = np.zeros((h, w, 3), np.uint8)
    assert oracle.co_capture_to_rgb(P(img), w, h, fmt, P(got)) == 0
    want = np.zeros((h, w, 3), np.uint8)
    rows, ch = (h * 3 // 2, 1) if fmt != 4 else (h, 4)
    assert ref.ref_cvtcolor(P(img), rows, w, ch, CV_CODE[fmt], P(want)) == h
    assert (got == want).all()


def test_formats_the_layout_cannot_hold_and_the_default(oracle):
    oracle.co_capture_bytes.restype = ctypes.c_size_t
    buf = np.zeros(64 * 64 * 4, np.uint8)
    out = np.zeros(64 * 64 * 3, np.uint8)
    for (w, h) in ((63, 64), (64, 63)):              # cvtColor(YUV420) asserts an even width and height: the reference would throw
        assert oracle.co_capture_bytes(w, h, 12) == 0 and oracle.co_capture_to_rgb(P(buf), w, h, 420, P(out)) == -1
    rgb = np.random.default_rng(1).integers(0, 256, 64 * 64 * 3, dtype=np.uint8)
    for fmt in (0, -1, 3, 7):                        # `format <= 0` is 3 (cimbar_recv_js.cpp:150-151); get_rgb's default: RGB8 as it is
        assert oracle.co_capture_to_rgb(P(rgb), 64, 64, fmt, P(out)) == 0 and (out == rgb).all()


def test_known_yuv_triples(oracle):
    """(Y, U, V) -> (R, G, B) worked out from the constants of color_yuv.simd.hpp with plain integers (the BT.601 primaries come back to within
    one count, e.g. red (81, 90, 240): (65 * 1220542 + 2^19 + 112 * 1673527) >> 20 = 254)"""
    cases = {(16, 128, 128): (0, 0, 0), (235, 128, 128): (255, 255, 255), (81, 90, 240): (254, 0, 0), (145, 54, 34): (0, 255, 1),
             (41, 240, 110): (0, 0, 255), (0, 128, 128): (0, 0, 0), (255, 128, 128): (255, 255, 255), (128, 0, 255): (255, 77, 0),
             (200, 255, 0): (10, 255, 255), (17, 129, 127): (0, 2, 3)}
    for (y, u, v), want in cases.items():
        img = np.array([y] * 4 + [u, v], np.uint8)           # a 2x2 NV12 image
        out = np.zeros(12, np.uint8)
        assert oracle.co_capture_to_rgb(P(img), 2, 2, 12, P(out)) == 0
        assert tuple(out[:3]) == want and (out.reshape(4, 3) == out[:3]).all(), ((y, u, v), tuple(out[:3]))
        img = np.array([y] * 4 + [v, u], np.uint8)           # the same as 420: V plane, then U plane
        assert oracle.co_capture_to_rgb(P(img), 2, 2, 420, P(out)) == 0 and tuple(out[:3]) == want


def reference_capture_chain(ref, img, w, h, fmt, mode=68):
    """the reference's cimbard_scan_extract_decode itself: returns (return value, the bytes it packed)"""
    g = pyref.GEOMETRY[mode]
    buf = np.zeros(g[6] * g[3], np.uint8)
    ref.cimbard_configure_decode(mode)
    r = ref.cimbard_scan_extract_decode(P(img), w, h, fmt, P(buf), buf.size)
    return r, buf


def oracle_capture_chain(oracle, img, w, h, fmt, ccm, mode=68):
    g = pyref.GEOMETRY[mode]
    frame = np.zeros((g[1], g[0], 3), np.uint8)
    rc = oracle.co_extract_fmt(P(img), w, h, fmt, P(frame), None)
    if rc == 0:
        return -3, np.zeros(0, np.uint8), ccm
    r, chunks, mask, ccm = pyref.oracle_decode(frame, 1, 2, ccm, mode=mode)          # cimbar_recv_js.cpp:166,185: shouldPreprocess = true
    packed = np.concatenate([chunks[j] for j in range(g[6]) if mask >> j & 1] + [np.zeros(0, np.uint8)])
    return packed.size, packed, ccm


@pytest.mark.parametrize("fmt", CF.FORMATS)
def test_oracle_capture_chain_equals_cimbard_scan_extract_decode(oracle, ref, synth, fmt):
    payload, frames = F.clean_frames(synth, len(CAMERA_CASES), seed=90)
    ref.ref_reset_ccm()
    ccm = pyref.CoCcm()
    decoded = 0
    for k, (bg, quad, blur) in enumerate(CAMERA_CASES):
        cam = F.camera_frame(frames[k], quad=quad, background=bg, blur=blur)
        h, w = cam.shape[:2]
        img = CF.rgb_to_format(cam, fmt)
        r, buf = reference_capture_chain(ref, img, w, h, fmt)
        got, packed, ccm = oracle_capture_chain(oracle, img, w, h, fmt, ccm)
        assert got == r, (k, got, r)
        if r > 0:
            assert (buf[:r] == packed).all()
            decoded += r
    assert decoded >= 7500          # (how much of a capture survives is the reference's business; agreeing with it is ours)
    # a capture with nothing in it: -3 (cimbar_recv_js.cpp:175-176)
    blank = np.full(CF.capture_bytes(640, 480, fmt), 16 if fmt in (12, 420) else 0, np.uint8)
    assert reference_capture_chain(ref, blank, 640, 480, fmt)[0] == -3
    assert oracle_capture_chain(oracle, blank, 640, 480, fmt, pyref.CoCcm())[0] == -3


def test_python_wrapper_refuses_a_format_its_array_cannot_describe():
    """HipDecoder._captures without size=: RGB from an (n,h,w,3) array, RGBA from (n,h,w,4); NV12 / 4:2:0 planes need size=(w,h) -- never a
    silent decode of another format's bytes as RGB."""
    from libcimbar_amd import decoder as D

    class Stub:
        pass
    stub = Stub()
    rgb = np.zeros((2, 8, 16, 3), np.uint8)
    arr, n, w, h, fmt = D.HipDecoder._captures(stub, rgb, None, 3)
    assert (n, w, h, fmt) == (2, 16, 8, 3)
    assert D.HipDecoder._captures(stub, rgb, None, 0)[4] == 3          # <= 0 is RGB, as in the C ABI
    rgba = np.zeros((2, 8, 16, 4), np.uint8)
    assert D.HipDecoder._captures(stub, rgba, None, 4)[1:] == (2, 16, 8, 4)
    for bad_fmt, bad in ((12, rgb), (420, rgb), (4, rgb), (3, rgba), (12, np.zeros((2, 8 * 16 * 3 // 2), np.uint8))):
        with pytest.raises(D.CimbarHipError):
            D.HipDecoder._captures(stub, bad, None, bad_fmt)
