#!/usr/bin/env python3
"""Writes tests/golden/opencv_pin.json: what a REAL OpenCV (cv2 >= 4.5) computes for every cv:: call on the reference's decode path, on seeded
inputs that need nothing but numpy (tests/opencv_pin_cases.py lists the calls with their reference file:line). Run it anywhere:

    pip install opencv-python-headless numpy
    python tools/opencv_pin_vectors.py            # -> tests/golden/opencv_pin.json (SHA-256 + a small raw crop per output, the OpenCV version)

Commit the file. tests/test_opencv_pin_vectors.py then checks the oracle's restatement AND the cv-shim behind oracle/_ref against it on every
run -- no cv2 needed on the test box -- which turns every [assumed-OpenCV] of the sources into a checked statement (or names the first case that
differs). This container and the GPU boxes have no cv2 and no network, so the file can only come from outside.

    python tools/opencv_pin_vectors.py --backend shim --out /tmp/x.json    # the same file from the cv-shim (needs oracle/_ref): harness self-test only
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import opencv_pin_cases as C          # noqa: E402  (numpy only)


class Cv2Backend:
    """each method is the reference's call sequence, spelled with cv2"""

    def __init__(self):
        import cv2
        self.cv2 = cv2
        major, minor = (int(x) for x in cv2.__version__.split(".")[:2])
        if (major, minor) < (4, 5):
            raise SystemExit(f"OpenCV {cv2.__version__}: the reference builds against 4.5 or newer")
        self.name = f"opencv {cv2.__version__}"

    def threshold(self, img, pre):          # CimbReader.cpp:30-46 (+ :17-27)
        cv2 = self.cv2
        gray = cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)
        block = 5
        if pre:
            k = np.array([[0, -1, 0], [-1, 4.5, -1], [0, -1, 0]], np.float32)
            gray = cv2.filter2D(gray, -1, k)
            block = 7
        mask = cv2.adaptiveThreshold(gray, 255, cv2.ADAPTIVE_THRESH_MEAN_C, cv2.THRESH_BINARY, block, 0)
        return np.packbits((mask & 1).astype(np.uint8), axis=1).reshape(-1)          # bitmatrix::mat_to_bitbuffer (bit_file/bitmatrix.h:14-46)

    def gray_blur(self, img, unit):          # Scanner.h:151-160
        cv2 = self.cv2
        return cv2.GaussianBlur(cv2.cvtColor(img, cv2.COLOR_RGB2GRAY), (unit, unit), 0)

    def otsu(self, img, unit):               # + Scanner.h:126-130
        cv2 = self.cv2
        t, binary = cv2.threshold(self.gray_blur(img, unit), 0, 255, cv2.THRESH_BINARY | cv2.THRESH_OTSU)
        return int(t), binary

    def deskew(self, img, corners8, size=1024, anchor=30):          # Deskewer.h:26-40 (mode B: 1024, anchor 30)
        cv2 = self.cv2
        dst = np.array([[anchor, anchor], [size - anchor, anchor], [anchor, size - anchor], [size - anchor, size - anchor]], np.float32)
        m = cv2.getPerspectiveTransform(corners8.reshape(4, 2), dst)
        return cv2.warpPerspective(img, m, (size, size), flags=cv2.INTER_LINEAR)

    def cvtcolor(self, buf, w, h, fmt):      # cimbar_recv_js.cpp:94-120
        cv2 = self.cv2
        if fmt == 4:
            return cv2.cvtColor(buf.reshape(h, w, 4), cv2.COLOR_RGBA2RGB)
        code = cv2.COLOR_YUV2RGB_NV12 if fmt == 12 else cv2.COLOR_YUV420p2RGB
        return cv2.cvtColor(buf.reshape(h * 3 // 2, w), code)

    def lsm(self, actual, desired):          # color_correction.h:26-39
        cv2 = self.cv2
        x, y = cv2.transpose(desired), cv2.transpose(actual)
        _, z = cv2.invert(y, flags=cv2.DECOMP_SVD)
        return cv2.gemm(x, z, 1.0, None, 0.0)          # `y = x * z` on cv::Mat is a gemm


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "opencv_pin.json"))
    ap.add_argument("--backend", default="cv2", choices=("cv2", "shim", "oracle"))
    ap.add_argument("--only", nargs="*", help="case names (default: all)")
    args = ap.parse_args()
    if args.backend == "cv2":
        backend = Cv2Backend()
    else:
        from tests.test_opencv_pin_vectors import OracleBackend, ShimBackend
        backend = ShimBackend() if args.backend == "shim" else OracleBackend()
    res = C.run_all(backend, set(args.only) if args.only else None)
    doc = {"format": C.FORMAT_VERSION, "produced_by": backend.name, "tool": "tools/opencv_pin_vectors.py",
           "note": "is_opencv is true only for files written by the cv2 backend; anything else is a harness self-test and pins nothing",
           "is_opencv": args.backend == "cv2", "cases": res}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(f"{len(res)} cases from {backend.name} -> {args.out}")


if __name__ == "__main__":
    main()
