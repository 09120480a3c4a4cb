"""One attempt at a REAL OpenCV pin (DESIGN "Unpinned part"): wherever `import cv2` works -- it does not in the build image nor on the GPU
boxes of the pool, and there is no network to install it from, so this module normally skips -- the OpenCV calls the reference makes on
the path (cvtColor RGB2GRAY, adaptiveThreshold 5 / 7, filter2D with CimbReader's sharpen kernel, GaussianBlur 3 / 5 / 9, threshold OTSU,
getPerspectiveTransform + warpPerspective INTER_LINEAR) are compared bit for bit with the oracle's restatement of them on synthetic frames
and captures. A maintainer with OpenCV >= 4.5 runs `pytest tests/test_opencv_pin.py` and turns every [assumed-OpenCV] of the sources into a
checked statement (or finds the first counter-example).

The portable form of the same pin -- a JSON written once on any machine with cv2 and checked here without it -- is tools/opencv_pin_vectors.py +
tests/test_opencv_pin_vectors.py."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed here (and cannot be: no network) -- the OpenCV boundary stays [assumed-OpenCV]")

from oracle import pyref            # noqa: E402
from oracle.pyref import P          # noqa: E402
from tests import frames as F      # noqa: E402


def plane_from_mask(mask):
    """bitmatrix::mat_to_bitbuffer (bit_file/bitmatrix.h:14-46): 8 mask bytes -> one byte, MSB = leftmost pixel"""
    return np.packbits((mask & 1).astype(np.uint8), axis=1).reshape(-1)


@pytest.fixture(scope="module")
def inputs(synth):
    _, fr = F.clean_frames(synth, 3, seed=321)
    return [fr[0], F.add_noise(F.shift(fr[1], 2, -1), 40, 1), F.rescale(fr[2], 6)]


@pytest.mark.parametrize("pre", [0, 1])
def test_gray_sharpen_adaptive_threshold(inputs, pre):
    """CimbReader.cpp:17-46: cvtColor -> (filter2D with [0 -1 0; -1 4.5 -1; 0 -1 0]) -> adaptiveThreshold(MEAN_C, BINARY, 5 | 7, 0)"""
    L = pyref.oracle_lib()
    for img in inputs:
        gray = cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)
        block = 5
        if pre:
            k = np.array([[0, -1, 0], [-1, 4.5, -1], [0, -1, 0]], np.float32)
            gray = cv2.filter2D(gray, -1, k)
            block = 7
        mask = cv2.adaptiveThreshold(gray, 255, cv2.ADAPTIVE_THRESH_MEAN_C, cv2.THRESH_BINARY, block, 0)
        want = np.zeros(1024 * 1024 // 8, np.uint8)
        L.co_threshold_bitplane(P(np.ascontiguousarray(img)), 1024, 1024, pre, P(want))
        assert (plane_from_mask(mask) == want).all(), f"preprocess {pre}: {int((plane_from_mask(mask) != want).sum())} plane bytes differ from OpenCV {cv2.__version__}"


@pytest.mark.parametrize("size", [(1280, 720), (1920, 1080), (3840, 2160), (4000, 2600)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_scan_preprocess_gray_blur_otsu(inputs, size):
    """Scanner::preprocess_image (Scanner.h:148-165): cvtColor -> GaussianBlur(unit 3 | 5 | 9, sigma 0) -> threshold(BINARY | OTSU)"""
    L = pyref.oracle_lib()
    w, h = size
    s = min(w, h) / 1080.0
    quad = tuple((int(x * s) + (w - int(1920 * s)) // 2, int(y * s)) for x, y in ((500, 40), (1480, 70), (470, 1030), (1500, 1000)))
    cam = np.ascontiguousarray(F.camera_frame(inputs[0], width=w, height=h, quad=quad, background=40, blur=0.5))
    unit = 3 if min(w, h) < 1500 else (5 if min(w, h) < 2500 else 9)
    gray = cv2.cvtColor(cam, cv2.COLOR_RGB2GRAY)
    blur = cv2.GaussianBlur(gray, (unit, unit), 0)
    thr, binary = cv2.threshold(blur, 0, 255, cv2.THRESH_BINARY | cv2.THRESH_OTSU)
    out = np.zeros((h, w), np.uint8)
    t = L.co_scan_preprocess(P(cam), w, h, P(out))
    assert t == int(thr) and (out == binary).all(), f"{w}x{h}: Otsu {t} vs {thr}, {int((out != binary).sum())} pixels differ"


def test_perspective_transform_and_warp(inputs):
    """Deskewer::deskew (Deskewer.h:26-40): getPerspectiveTransform + warpPerspective(INTER_LINEAR) to the frame size"""
    L = pyref.oracle_lib()
    cam = np.ascontiguousarray(F.camera_frame(inputs[0], quad=((500, 40), (1480, 70), (470, 1030), (1500, 1000)), background=30))
    corners = np.array([530, 70, 1452, 98, 501, 1001, 1470, 972], np.float32)
    dst = np.zeros(8, np.float32)
    L.co_deskew_points(P(dst))
    m = cv2.getPerspectiveTransform(corners.reshape(4, 2), dst.reshape(4, 2))
    want = cv2.warpPerspective(cam, m, (1024, 1024), flags=cv2.INTER_LINEAR)
    out = np.zeros((1024, 1024, 3), np.uint8)
    assert L.co_deskew(P(cam), 1920, 1080, P(corners), P(out)) == 0
    assert (out == want).all(), f"{int((out != want).any(axis=2).sum())} deskewed pixels differ from OpenCV {cv2.__version__}"
