"""BASELINE configs[0]: one mode-B frame of the reference's LICENSE as `./cimbar --encode` makes it (zstd 16, encode id 109,
cimbar.cpp:106-121), decoded to bytes. The fixture (tests/golden/config1.json, made by oracle/make_golden_config1.py with the
reference build) carries the frame's payload; the frame itself is re-rendered from it."""
import base64
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from libcimbar_amd import framegen
from oracle import pyref
from oracle.pyref import P

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1.json")))
PAYLOAD = np.frombuffer(base64.b64decode(FIX["payload_b64"]), np.uint8).reshape(12, 625)


def sink_to_file(L, chunks, mask):
    """what ./cimbar does with the decoder's output: chunks -> fountain_decoder_sink -> recovered file -> zstd decompress"""
    L.ref_sink_reset(625)
    fid = 0
    for j in range(12):
        if mask & (1 << j):
            fid = L.ref_sink_decode_frame(P(np.ascontiguousarray(chunks[j])), 625)
            if fid > 0:
                break
    assert fid == FIX["file_id"]
    comp = np.zeros(FIX["compressed_size"], np.uint8)
    assert L.ref_sink_recover(ctypes.c_uint32(fid), P(comp), comp.size) == 1
    out = np.zeros(1 << 16, np.uint8)
    n = L.ref_zstd_decompress(P(comp), comp.size, P(out), out.size)
    return out[:n]


def test_fixture_frame_is_the_reference_encoders_frame(ref, synth):
    """pins the fixture: the frame rendered from its payload is byte for byte what Encoder::encode_next made of LICENSE"""
    frame = synth.frames_from_payload(torch.from_numpy(PAYLOAD.reshape(1, 7500).copy())).numpy()[0]
    assert hashlib.sha256(frame.tobytes()).hexdigest() == FIX["frame_sha256"]
    if os.path.exists("/root/reference/LICENSE"):
        data = np.frombuffer(open("/root/reference/LICENSE", "rb").read(), np.uint8)
        rgb = np.zeros((1024, 1024, 3), np.uint8)
        assert ref.ref_encode_fountain_z(P(data), len(data), 109, 16, b"LICENSE", 0, 1, P(rgb)) == 1
        assert (rgb == frame).all()


def test_oracle_decodes_config1_to_the_file(ref, synth):
    frame = synth.frames_from_payload(torch.from_numpy(PAYLOAD.reshape(1, 7500).copy())).numpy()[0]
    r, chunks, mask, _ = pyref.oracle_decode(frame)
    assert r == 7500 and mask == 0xFFF and (chunks == PAYLOAD).all()
    out = sink_to_file(ref, chunks, mask)
    assert out.size == FIX["file_size"] and hashlib.sha256(out.tobytes()).hexdigest() == FIX["file_sha256"]


@pytest.mark.gpu
def test_gpu_decodes_config1_to_the_file(hip_decoder, ref):
    """payload -> cimbar_hip_encode_batch (the frame) -> cimbar_hip_decode_frame -> the reference's sink + zstd -> LICENSE"""
    frame = hip_decoder.encode_batch(PAYLOAD.reshape(1, 7500))[0]
    assert hashlib.sha256(frame.tobytes()).hexdigest() == FIX["frame_sha256"]
    hip_decoder.reset_ccm()
    good, chunks, mask = hip_decoder.decode_frame(frame)
    assert good == 7500 and mask == 0xFFF and (chunks == PAYLOAD).all()
    out = sink_to_file(ref, chunks, mask)
    assert out.size == FIX["file_size"] and hashlib.sha256(out.tobytes()).hexdigest() == FIX["file_sha256"]
