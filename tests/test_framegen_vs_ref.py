"""libcimbar_amd/framegen.py (input manufacture) against Encoder::encode_next of the reference build."""
import numpy as np
import torch

from libcimbar_amd import framegen
from oracle import pyref
from oracle.pyref import P


def test_frames_byte_identical_to_reference_encoder(ref, synth):
    payload = framegen.synth_payload(3, seed=5)
    frames = synth.frames_from_payload(payload).numpy()
    for k in range(3):
        assert (frames[k] == pyref.ref_encode_raw(payload[k].numpy())).all()


def test_fountain_stream_frames(ref, synth):
    # a real wirehair stream rendered by the reference == our renderer fed the same chunk bytes
    data = np.random.default_rng(4321).integers(0, 256, 40000, dtype=np.uint8)
    chunks = np.zeros((24, 625), np.uint8)
    assert ref.ref_fountain_chunks(P(data), data.size, 7, 24, P(chunks)) == 24
    want = np.zeros((2, 1024, 1024, 3), np.uint8)
    assert ref.ref_encode_fountain(P(data), data.size, 7, 0, 2, P(want)) == 2
    got = synth.frames_from_payload(torch.from_numpy(chunks.reshape(2, 7500))).numpy()
    assert (got == want).all()


def test_rs_encode_matches_libcorrect(ref):
    msg = np.random.default_rng(0).integers(0, 256, (8, 125), dtype=np.uint8)
    enc = framegen.rs_encode(torch.from_numpy(msg)).numpy()
    for k in range(8):
        o = np.zeros(155, np.uint8)
        ref.ref_rs_encode(P(msg[k]), 125, 30, P(o))
        assert (o == enc[k]).all()
