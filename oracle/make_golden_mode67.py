"""Generates tests/golden/mode67.json by running the REFERENCE build (oracle/_ref) in mode 67 ("Bm", Conf8x8_mini, Config.h:38-39):
for a few seeded frames -- clean and distorted, rendered by libcimbar_amd.framegen from a seeded payload (the frame's SHA-256 is recorded,
and the clean frame is checked here against Encoder::encode_next byte for byte) -- what Decoder::decode_fountain returns: good bytes, the
chunk mask, SHA-256 of the 12 chunk slots, with the colour-correction state carried from frame to frame like one decode thread does.
tests/test_mode67.py replays the list against the oracle (no GPU) and tests/test_gpu_mode67.py against the HIP path."""
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

MODE = 67


def cases(synth):
    """(name, preprocess, frame) in decode order -- shared with the tests"""
    items = F.distorted_set(synth, seed=67)
    out = [(nm, 0, fr) for nm, fr in items]
    out += [("sharpen:" + nm, 1, fr) for nm, fr in items[:5]]
    return out


def main():
    synth = framegen.FrameSynth("cpu", MODE)
    rows = []
    with pyref.ref_mode(MODE):
        payload, frames = F.clean_frames(synth, 1, seed=67)
        assert (pyref.ref_encode_raw(payload[0], MODE) == frames[0]).all()
        for k, (nm, pre, fr) in enumerate(cases(synth)):
            r, chunks, mask = pyref.ref_decode(fr, pre, 2, reset_ccm=(k == 0), mode=MODE)
            rows.append({"name": nm, "preprocess": pre, "frame_sha256": hashlib.sha256(np.ascontiguousarray(fr).tobytes()).hexdigest(),
                         "good_bytes": int(r), "mask": int(mask), "chunks_sha256": hashlib.sha256(chunks.tobytes()).hexdigest()})
            print(nm, pre, r, hex(mask))
    path = os.path.join(ROOT, "tests", "golden", "mode67.json")
    json.dump({"generator": "oracle/make_golden_mode67.py", "mode": MODE, "color_correction": 2, "frames": rows}, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
