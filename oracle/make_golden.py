"""Generates tests/golden/modeb_golden.json by running the REFERENCE build (oracle/_ref/libcimbar_ref.so: the reference's
own sources + cv-shim) on the deterministic input set of tests/frames.py. Commit the output; the tests that read it need
neither /root/reference nor oracle/_ref.

For every case: sha256 of the input frame (guards input regeneration), the reference's return value and chunk mask, sha256
of the 12x625 chunk slots, of the packed bitplane, of the flood-ordered visit list (cell, x, y, symbol), the colour-
correction matrix left in the thread_local after the frame (raw float32 bits), and the return value + sha256 of the 7500
bytes Decoder::decode (the --no-fountain path) writes for the same frame from a fresh thread state.
"""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyref import P, ref_lib  # noqa: E402
from libcimbar_amd import framegen  # noqa: E402
from tests import frames as F  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(L, frame, pre, cc):
    frame = np.ascontiguousarray(frame)
    chunks = np.zeros((12, 625), np.uint8)
    mask = ctypes.c_uint32(0)
    r = L.ref_decode_fountain(P(frame), 1024, 1024, pre, cc, 1, P(chunks), ctypes.byref(mask))
    m = (ctypes.c_float * 9)()
    active = L.ref_get_ccm(m)
    plane = np.zeros(131072, np.uint8)
    visit = np.zeros(4 * 12400, np.int32)
    L.ref_symbol_pass(P(frame), 1024, 1024, pre, P(plane), P(visit))
    plain = np.zeros(7500, np.uint8)
    pr = L.ref_decode_plain(P(frame), 1024, 1024, pre, cc, 1, P(plain))      # Decoder::decode, fresh thread state
    return {"input_sha256": sha(frame), "ret": int(r), "mask": int(mask.value), "chunks_sha256": sha(chunks),
            "plain_ret": int(pr), "plain_sha256": sha(plain),
            "bitplane_sha256": sha(plane), "visit_sha256": sha(visit), "ccm_active": int(active),
            "ccm_bits": [int(np.float32(x).view(np.uint32)) for x in m]}


def main():
    L = ref_lib()
    if L is None:
        raise SystemExit("oracle/_ref/libcimbar_ref.so missing: run `make -C oracle ref` first")
    synth = framegen.FrameSynth("cpu")
    out = {"generator": "oracle/make_golden.py", "cases": []}
    for name, frame in F.distorted_set(synth, seed=77):
        for pre in (0, 1):
            c = run_case(L, frame, pre, 2)
            c.update(name=name, preprocess=pre, color_correction=2)
            out["cases"].append(c)
    _, tf = F.tile_error_frames(synth, 2, seed=4321)
    for k in range(2):
        c = run_case(L, tf[k], 0, 2)
        c.update(name=f"tile_errors_{k}", preprocess=0, color_correction=2)
        out["cases"].append(c)
    _, cf = F.clean_frames(synth, 1, seed=31)
    for cc in (0, 1):
        c = run_case(L, cf[0], 0, cc)
        c.update(name=f"clean_cc{cc}", preprocess=0, color_correction=cc)
        out["cases"].append(c)
    # the stage in front of the decoder (Scanner / Deskewer / Extractor of the reference build) on synthetic camera captures
    from tests.test_oracle_vs_ref import CAMERA_CASES
    ex = []
    for k, (bg, quad, blur) in enumerate(CAMERA_CASES):
        _, fr = F.clean_frames(synth, 1, seed=50 + k)
        cam = np.ascontiguousarray(F.camera_frame(fr[0], quad=quad, background=bg, blur=blur))
        h, w = cam.shape[:2]
        binimg = np.zeros((h, w), np.uint8)
        L.ref_scan_preprocess(P(cam), w, h, P(binimg))
        corners = (ctypes.c_float * 8)()
        nanch = L.ref_scan_corners(P(cam), w, h, corners)
        desk = np.zeros((1024, 1024, 3), np.uint8)
        rc = L.ref_extract(P(cam), w, h, P(desk))
        ex.append({"case": k, "input_sha256": sha(cam), "width": w, "height": h, "binary_sha256": sha(binimg), "anchors": int(nanch),
                   "corners": [float(c) for c in corners], "extract_rc": int(rc), "deskewed_sha256": sha(desk)})
    out["extract"] = ex
    # Reed-Solomon known answers straight from libcorrect: 40 random blocks with 0..22 byte errors
    g = np.random.default_rng(2024)
    rs = []
    for t in range(40):
        msg = g.integers(0, 256, 125, dtype=np.uint8)
        enc = np.zeros(155, np.uint8)
        L.ref_rs_encode(P(msg), 125, 30, P(enc))
        ne = int(t * 22 // 39)
        pos = g.choice(155, ne, replace=False)
        bad = enc.copy()
        bad[pos] ^= g.integers(1, 256, ne, dtype=np.uint8)
        outb = np.zeros(125, np.uint8)
        r = L.ref_rs_decode(P(bad), 155, 30, P(outb))
        rs.append({"received": bad.tobytes().hex(), "ret": int(r), "decoded": outb.tobytes().hex() if r > 0 else "", "errors": ne})
    out["rs_vectors"] = rs
    path = os.path.join(ROOT, "tests", "golden", "modeb_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
