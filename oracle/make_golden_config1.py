"""Generates tests/golden/config1.json: BASELINE configs[0] -- ONE mode-B frame of /root/reference/LICENSE made the way
`./cimbar --encode` makes it (zstd level 16, file name in a skippable header, encode id 109: cimbar.cpp:106-121) -- by running
the REFERENCE build (oracle/_ref). The fixture holds the frame's 7500 payload bytes (its 12 fountain chunks) as the reference
decodes them, SHA-256 of the frame itself (the library's encoder re-renders it from the payload, byte for byte), and length +
SHA-256 of the file that `./cimbar` writes after feeding the chunks to its sink and decompressing."""
import base64
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyref import P, ref_lib, ref_decode  # noqa: E402


def main():
    L = ref_lib()
    data = np.frombuffer(open("/root/reference/LICENSE", "rb").read(), np.uint8)
    rgb = np.zeros((1024, 1024, 3), np.uint8)
    assert L.ref_encode_fountain_z(P(data), len(data), 109, 16, b"LICENSE", 0, 1, P(rgb)) == 1
    r, chunks, mask = ref_decode(rgb)
    assert r == 7500 and mask == 0xFFF
    L.ref_sink_reset(625)
    fid, used = 0, 0
    for j in range(12):
        used += 1
        fid = L.ref_sink_decode_frame(P(chunks[j]), 625)
        if fid > 0:
            break
    assert fid > 0
    size = (int(chunks[0, 1]) << 16) | (int(chunks[0, 2]) << 8) | int(chunks[0, 3])
    comp = np.zeros(size, np.uint8)
    assert L.ref_sink_recover(ctypes.c_uint32(fid), P(comp), size) == 1
    out = np.zeros(1 << 16, np.uint8)
    n = L.ref_zstd_decompress(P(comp), size, P(out), out.size)
    assert n == len(data) and (out[:n] == data).all()
    fix = {"generator": "oracle/make_golden_config1.py", "encode_id": 109, "compression": 16, "file": "LICENSE",
           "file_size": int(len(data)), "file_sha256": hashlib.sha256(data.tobytes()).hexdigest(), "compressed_size": size,
           "chunks_used_by_sink": used, "file_id": int(fid), "frame_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
           "payload_b64": base64.b64encode(chunks.tobytes()).decode()}
    path = os.path.join(ROOT, "tests", "golden", "config1.json")
    json.dump(fix, open(path, "w"), indent=1)
    print("wrote", path, size, used, fid)


if __name__ == "__main__":
    main()
