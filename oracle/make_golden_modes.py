"""Generates tests/golden/mode67.json and mode66.json by running the REFERENCE build (oracle/_ref) in mode 67 ("Bm", Conf8x8_mini) and 66 ("Bu",
Conf8x8_micro; Config.h:36-39):
for a few seeded frames -- clean and distorted, rendered by libcimbar_amd.framegen from a seeded payload (the frame's SHA-256 is recorded,
and the clean frame is checked here against Encoder::encode_next byte for byte) -- what Decoder::decode_fountain returns: good bytes, the
chunk mask, SHA-256 of the chunk slots, with the colour-correction state carried from frame to frame like one decode thread does.
tests/test_modes.py replays the lists against the oracle (no GPU) and tests/test_gpu_modes.py against the HIP path."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libcimbar_amd import framegen  # noqa: E402
from oracle import pyref  # noqa: E402
from tests import frames as F  # noqa: E402

MODES = (67, 66, 4, 8)


def cases(synth):
    """(name, preprocess, frame) in decode order -- shared with the tests"""
    items = F.distorted_set(synth, seed=synth.geo.MODE)
    out = [(nm, 0, fr) for nm, fr in items]
    out += [("sharpen:" + nm, 1, fr) for nm, fr in items[:5]]
    return out


def main():
    for mode in MODES:
        synth = framegen.FrameSynth("cpu", mode)
        rows = []
        with pyref.ref_mode(mode):
            payload, frames = F.clean_frames(synth, 1, seed=mode)
            assert (pyref.ref_encode_raw(payload[0], mode) == frames[0]).all()
            for k, (nm, pre, fr) in enumerate(cases(synth)):
                r, chunks, mask = pyref.ref_decode(fr, pre, 2, reset_ccm=(k == 0), mode=mode)
                rows.append({"name": nm, "preprocess": pre, "frame_sha256": hashlib.sha256(np.ascontiguousarray(fr).tobytes()).hexdigest(),
                             "good_bytes": int(r), "mask": int(mask), "chunks_sha256": hashlib.sha256(chunks.tobytes()).hexdigest()})
                print(mode, nm, pre, r, hex(mask))
        path = os.path.join(ROOT, "tests", "golden", "mode%d.json" % mode)
        json.dump({"generator": "oracle/make_golden_modes.py", "mode": mode, "color_correction": 2, "frames": rows}, open(path, "w"), indent=1)
        print("wrote", path)


if __name__ == "__main__":
    main()
