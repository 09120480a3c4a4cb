"""CPU: the thread-pooled reference batch the BASELINE-sized GPU tests compare against (tests/refbatch.py) equals the reference build decoding the
same frames on one thread in order, and both return the encoder's payload."""
import numpy as np

from libcimbar_amd import framegen
from tests.refbatch import reference_batch


def test_pooled_reference_batch_equals_one_thread_in_order():
    synth = framegen.FrameSynth("cpu")
    payload = framegen.synth_payload(24, seed=7)
    frames = synth.frames_from_payload(payload).numpy()
    c1, m1 = reference_batch(frames, threads=1)
    c5, m5 = reference_batch(frames, threads=5)
    assert (c1 == c5).all() and (m1 == m5).all()
    assert (m1 == 0xFFF).all() and (c1 == payload.numpy()).all()
