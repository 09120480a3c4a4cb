"""The portable OpenCV pin. tools/opencv_pin_vectors.py, run anywhere with cv2 >= 4.5, writes
tests/golden/opencv_pin.json; here -- no cv2 needed -- the oracle's restatement of every cv:: call on the path AND the cv-shim that oracle/_ref is
compiled against are run on the same seeded inputs (tests/opencv_pin_cases.py) and compared with that file when it exists and says it came from
OpenCV. Whether or not it exists, the two must agree with each other: one committed JSON then closes every [assumed-OpenCV] for both at once."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyref
from oracle.pyref import P
from tests import opencv_pin_cases as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tests", "golden", "opencv_pin.json")
CV_CODE = {12: 90, 420: 98, 4: 1}       # COLOR_YUV2RGB_NV12, COLOR_YUV420p2RGB, COLOR_RGBA2RGB
FAST = [n for n in C.CASES if "4700x4600" not in n and "3200x2600" not in n]      # the two large blurs take the oracle a few seconds each


class OracleBackend:
    """oracle/cimbar_oracle*.c (mode B build): the restatement the GPU path is checked against"""
    name = "oracle (C restatement)"

    def __init__(self):
        self.L = pyref.oracle_lib(68)

    def threshold(self, img, pre):
        h, w = img.shape[:2]
        out = np.zeros(w * h // 8, np.uint8)
        self.L.co_threshold_bitplane(P(img), w, h, int(pre), P(out))
        return out

    def gray_blur(self, img, unit):
        h, w = img.shape[:2]
        out = np.zeros((h, w), np.uint8)
        assert self.L.co_gray_blur(P(img), w, h, P(out), None) == unit, "Scanner's rule picks another kernel for this size"
        return out

    def otsu(self, img, unit):
        h, w = img.shape[:2]
        out = np.zeros((h, w), np.uint8)
        t = self.L.co_scan_preprocess(P(img), w, h, P(out))
        return int(t), out

    def deskew(self, img, corners8):
        h, w = img.shape[:2]
        out = np.zeros((1024, 1024, 3), np.uint8)
        assert self.L.co_deskew(P(img), w, h, P(np.ascontiguousarray(corners8, np.float32)), P(out)) == 0
        return out

    def cvtcolor(self, buf, w, h, fmt):
        out = np.zeros((h, w, 3), np.uint8)
        assert self.L.co_capture_to_rgb(P(buf), w, h, fmt, P(out)) == 0
        return out

    def lsm(self, actual, desired):
        out = np.zeros(9, np.float32)
        self.L.co_moore_penrose_lsm(P(np.ascontiguousarray(actual)), P(np.ascontiguousarray(desired)), int(actual.shape[0]), P(out))
        return out


class ShimBackend:
    """oracle/cvshim behind the reference's own code (oracle/_ref): the reference's call sequences, the shim's arithmetic"""
    name = "cv-shim behind oracle/_ref"

    def __init__(self):
        self.R = pyref.ref_lib()
        if self.R is None:
            raise RuntimeError("oracle/_ref is not built")

    def threshold(self, img, pre):
        h, w = img.shape[:2]
        out = np.zeros(w * h // 8, np.uint8)
        self.R.ref_symbol_pass(P(img), w, h, int(pre), P(out), None)          # CimbReader's constructor: cvtColor, (filter2D,) adaptiveThreshold, bit-pack
        return out

    def gray_blur(self, img, unit):
        h, w = img.shape[:2]
        out = np.zeros((h, w), np.uint8)
        self.R.ref_gray_blur(P(img), w, h, int(unit), P(out))
        return out

    def otsu(self, img, unit):
        h, w = img.shape[:2]
        gray = self.gray_blur(img, unit)
        out = np.zeros((h, w), np.uint8)
        t = self.R.ref_otsu_threshold(P(gray), w, h, P(out))
        whole = np.zeros((h, w), np.uint8)
        self.R.ref_scan_preprocess(P(img), w, h, P(whole))                      # Scanner::preprocess_image itself picks the same kernel and gets the same image
        assert (whole == out).all()
        return int(t), out

    def deskew(self, img, corners8):
        h, w = img.shape[:2]
        out = np.zeros((1024, 1024, 3), np.uint8)
        assert self.R.ref_deskew(P(img), w, h, P(np.ascontiguousarray(corners8, np.float32)), P(out)) == 1024
        return out

    def cvtcolor(self, buf, w, h, fmt):
        out = np.zeros((h, w, 3), np.uint8)
        rows, ch = (h * 3 // 2, 1) if fmt != 4 else (h, 4)
        assert self.R.ref_cvtcolor(P(buf), rows, w, ch, CV_CODE[fmt], P(out)) == h
        return out

    def lsm(self, actual, desired):
        out = np.zeros(9, np.float32)
        self.R.ref_moore_penrose_lsm(P(np.ascontiguousarray(actual)), P(np.ascontiguousarray(desired)), int(actual.shape[0]), P(out))
        return out


@pytest.fixture(scope="module")
def oracle_results():
    return C.run_all(OracleBackend())


@pytest.fixture(scope="module")
def shim_results(ref):
    return C.run_all(ShimBackend())


def test_the_generator_is_the_same_everywhere():
    """the inputs must not depend on the numpy at hand: known digests of the generator's first bytes and of one image of each kind"""
    assert C.rand_u8(1, 16).tobytes().hex() == C.rand_u8(1, 24)[:16].tobytes().hex()
    assert C.rand_u8(7, 8).tobytes().hex() == "be5f4bceafecd101"
    for spec, want in KNOWN_INPUTS.items():
        kind, w, h, seed = spec
        assert C.digest(C.make_image(spec))["sha256"] == want, spec
    assert C.digest(C.capture(12, 70, 38, 42))["sha256"] == KNOWN_CAPTURE
    assert C.digest(C.lsm_inputs(51)[0])["sha256"] == KNOWN_LSM_ACTUAL


def test_oracle_and_shim_agree_on_every_case(oracle_results, shim_results):
    bad = C.compare(oracle_results, shim_results)
    assert not bad, bad
    assert set(oracle_results) == set(C.CASES) == set(shim_results)


def test_against_the_opencv_pin_file(oracle_results, shim_results):
    if not os.path.exists(PIN):
        pytest.skip("tests/golden/opencv_pin.json is not there: run tools/opencv_pin_vectors.py on a machine with cv2 >= 4.5 and commit the file "
                    "(no cv2 and no network here) -- until then the OpenCV boundary stays [assumed-OpenCV]")
    doc = json.load(open(PIN))
    assert doc.get("format") == C.FORMAT_VERSION
    if not doc.get("is_opencv"):
        pytest.fail(f"{PIN} was written by '{doc.get('produced_by')}', not by OpenCV: it pins nothing and must not be committed")
    want = doc["cases"]
    assert set(want) == set(C.CASES), "the pin file was made from another case list: regenerate it"
    bad = [("oracle",) + b for b in C.compare(oracle_results, want)] + [("cv-shim",) + b for b in C.compare(shim_results, want)]
    assert not bad, f"differs from {doc['produced_by']}: {bad}"


def test_the_vector_tool_end_to_end_with_the_shim_standing_in(tmp_path, ref, oracle_results):
    """tools/opencv_pin_vectors.py --backend shim writes a file of the same shape a cv2 run writes; the comparison code accepts it for the oracle and
    the pin test refuses to treat it as OpenCV's"""
    out = tmp_path / "pin.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "opencv_pin_vectors.py"), "--backend", "shim", "--out", str(out), "--only", *FAST],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    doc = json.load(open(out))
    assert doc["is_opencv"] is False and doc["format"] == C.FORMAT_VERSION and set(doc["cases"]) == set(FAST)
    assert not C.compare(oracle_results, doc["cases"])
    one = doc["cases"]["threshold5_tiles"]["bitplane"]
    assert one["shape"] == [1024 * 1024 // 8] and len(one["sha256"]) == 64
    assert "crop" in doc["cases"]["deskew_1920x1080"]["frame"] and "values_hex" in doc["cases"]["lsm_a"]["ccm_f32"]


def test_a_wrong_restatement_is_caught(oracle_results):
    """the comparison is not vacuous: one flipped bit in one output is named"""
    import copy
    other = copy.deepcopy(oracle_results)
    other["nv12_70x38"]["rgb"]["sha256"] = "0" * 64
    bad = C.compare(oracle_results, other)
    assert len(bad) == 1 and bad[0][0] == "nv12_70x38" and bad[0][1] == "rgb"


KNOWN_INPUTS = {
    ('tiles', 1024, 1024, 11): "e7ebf05a8d25f733c0b199b618a2917488251bcf12ca521d1f318c06cdecbb43",
    ('noise', 1000, 701, 25): "c08a21244c1ac302c8545c74cd024771fab03771479ecb69153d5d201459a9b2",
    ('camera', 1280, 720, 21): "aa8fb431935288164be3c1d9d03348a845c3522039511236056a80dadd991216",
}
KNOWN_CAPTURE = "e57d773407895ce3fe31f7d0f53e5eece1c0d3907d37e43c1a420ed535acb977"
KNOWN_LSM_ACTUAL = "c452c33ad9acabaae445f3f3501ddba72b1b57e41f409a0796700e29f8199266"
